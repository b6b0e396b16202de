# PMC passes of the kernels round 5 changed (separate --pmc passes, kernel-trace only): the U table's reverse stage and H-VJP, the
# Y table's fused forward / reverse steps, the network theta-VJP.  -> gpurun_out/pmc_r05.txt (medians per kernel and counter)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc5; rm -rf $O; mkdir -p $O
run() { # tag counters kernel G law
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py $3 $4 1024 6 $5 > $O/$1.log 2>&1 || tail -2 $O/$1.log
}
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
BUSY="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for k in "adj_stage2 nnU_tab" "vjp_H nnU_tab" "rk_stage2 nnU_tab" "adj_fused_step nnY_tab" "fused_step nnY_tab" "vjp_theta nnU"; do
  set -- $k; t=$1_$2
  run ${t}_f64 "$F64" $1 8 $2
  run ${t}_busy "$BUSY" $1 8 $2
  run ${t}_fetch FETCH_SIZE $1 8 $2
  run ${t}_write WRITE_SIZE $1 8 $2
done
cd $R && { echo "# tools/pmc_r05.sh: rocprofv3 --pmc medians per kernel (8 x 1024^2 = 8 388 608 cells; FETCH_SIZE / WRITE_SIZE in KiB, FETCH x 2 per the gfx950 note)"; for k in "adj_stage2 nnU_tab" "vjp_H nnU_tab" "rk_stage2 nnU_tab" "adj_fused_step nnY_tab" "fused_step nnY_tab" "vjp_theta nnU"; do set -- $k; echo "== $1 $2: $(grep us/launch $O/$1_$2_f64.log | tail -1)"; python tools/pmc_summary.py "$O/$1_$2_*/**/*counter_collection.csv" 2>&1 | grep -v "k_utab_build\|k_ytab_build\|k_begin\|k_controller\|k_set\|k_init" | cut -c1-420; done; } > gpurun_out/pmc_r05.txt; cat gpurun_out/pmc_r05.txt | head -60

# PMC passes of round 4 (profiles/r04/pmc_roofline.json): the bench's workload is the 64 x 1024^2 job now, so the fused step kernel
# (gridded-NN and constant A) and the HBM-bound kernels are counted on 64 glaciers; fp64 instruction counts of the fused reverse
# step and of the reverse / forward stage kernels with an inlined network (Y and U law, 8 x 512^2 like tools/workflow_probe.py).
# Separate --pmc passes, kernel-trace only, as MI355X_MICROARCH.md prescribes.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmcr; rm -rf $O; mkdir -p $O
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
BUSY="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run() { # tag counters kernel G n law
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py $3 $4 $5 6 $6 > $O/$1.log 2>&1
}
for law in nnA const; do
  run fused_${law}_f64 "$F64" fused_step 64 1024 $law
  run fused_${law}_busy "$BUSY" fused_step 64 1024 $law
  run fused_${law}_fetch FETCH_SIZE fused_step 64 1024 $law
  run fused_${law}_write WRITE_SIZE fused_step 64 1024 $law
done
run dhdt_nnA_fetch FETCH_SIZE dhdt 64 1024 nnA
run dhdt_nnA_write WRITE_SIZE dhdt 64 1024 nnA
run stage_fetch FETCH_SIZE rk_stage2 64 1024 const
run stage_write WRITE_SIZE rk_stage2 64 1024 const
run adjf_f64 "$F64" adj_fused_step 64 1024 const
run adjf_busy "$BUSY" adj_fused_step 64 1024 const
run adjf_fetch FETCH_SIZE adj_fused_step 64 1024 const
run adjf_write WRITE_SIZE adj_fused_step 64 1024 const
for law in nnY nnU; do
  run adjs_${law}_f64 "$F64" adj_stage2 8 512 $law
  run adjs_${law}_busy "$BUSY" adj_stage2 8 512 $law
  run fwds_${law}_f64 "$F64" rk_stage2 8 512 $law
  run fwds_${law}_busy "$BUSY" rk_stage2 8 512 $law
done
run fwds_nnY16_f64 "$F64" rk_stage2 8 1024 nnY16
run fwds_nnY16_busy "$BUSY" rk_stage2 8 1024 nnY16
grep -h "us/launch" $O/*_f64.log
cd $R && python tools/pmc_r04.py $O > $R/gpurun_out/r04/pmc_roofline.json && head -c 1500 $R/gpurun_out/r04/pmc_roofline.json

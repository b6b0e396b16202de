#!/bin/bash
# the U law's reverse step, 64 x 1024^2 (or $1 glaciers): the fused LDS-tile step (k_adj_fused_lds) against one staged stage
# (k_adj_stage<2, LM_UTAB>): time, fp64 instructions, VALU busy / waits, LDS, HBM traffic -> gpurun_out/pmc_adju.txt
R=${GRAFT_REPO_ROOT:-$PWD}; G=${1:-64}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_adju; rm -rf $O; mkdir -p $O
run() { # tag counters kernel skip
  if [ "$4" = 1 ]; then export ODINN_TIMED_ADJ_SKIP=1; else unset ODINN_TIMED_ADJ_SKIP; fi
  timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py $3 $G 1024 6 nnU_tab > $O/$1.log 2>&1 || tail -2 $O/$1.log
}
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
BUSY="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
MEM="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
for k in "adj_fused_step 0" "adj_fused_step 1" "adj_stage2 0"; do
  set -- $k; t=$1_skip$2
  run ${t}_f64 "$F64" $1 $2
  run ${t}_busy "$BUSY" $1 $2
  run ${t}_mem "$MEM" $1 $2
  run ${t}_fetch FETCH_SIZE $1 $2
  run ${t}_write WRITE_SIZE $1 $2
done
cd $R && { for k in "adj_fused_step 0" "adj_fused_step 1" "adj_stage2 0"; do set -- $k; t=$1_skip$2; echo "== $t: $(grep us/launch $O/${t}_f64.log | tail -1)"; python tools/pmc_summary.py "$O/${t}_*/**/*counter_collection.csv" 2>&1 | grep "k_adj_fused_lds\|k_adj_stage" | cut -c1-460; done; } > gpurun_out/pmc_adju.txt; cat gpurun_out/pmc_adju.txt

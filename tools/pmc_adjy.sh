#!/bin/bash
# the fused reverse step, constant A against the Y table, dense and with the ice-free shortcut, 64 x 1024^2: time, fp64 instructions,
# VALU busy / waits, LDS, HBM traffic (separate --pmc passes, kernel-trace only) -> gpurun_out/pmc_adjy.txt
R=${GRAFT_REPO_ROOT:-$PWD}; G=${1:-64}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_adjy; rm -rf $O; mkdir -p $O
run() { # tag counters law skip
  if [ "$4" = 1 ]; then export ODINN_TIMED_ADJ_SKIP=1; else unset ODINN_TIMED_ADJ_SKIP; fi
  timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py adj_fused_step $G 1024 6 $3 > $O/$1.log 2>&1 || tail -2 $O/$1.log
}
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
BUSY="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
MEM="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
for law in const nnY_tab; do for sk in 0 1; do
  t=${law}_skip$sk
  run ${t}_f64 "$F64" $law $sk
  run ${t}_busy "$BUSY" $law $sk
  run ${t}_mem "$MEM" $law $sk
  run ${t}_fetch FETCH_SIZE $law $sk
  run ${t}_write WRITE_SIZE $law $sk
done; done
cd $R && { for law in const nnY_tab; do for sk in 0 1; do t=${law}_skip$sk; echo "== $t: $(grep us/launch $O/${t}_f64.log | tail -1)"; python tools/pmc_summary.py "$O/${t}_*/**/*counter_collection.csv" 2>&1 | grep "k_adj_fused_strip" | cut -c1-460; done; done; } > gpurun_out/pmc_adjy.txt; cat gpurun_out/pmc_adjy.txt

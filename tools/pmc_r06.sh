#!/bin/bash
# PMC passes behind bench.py's roofline blocks, collected EVERY round on the bench workload itself (64 x 1024^2, one MI355X):
# fp64 instruction counts, VALU-busy and HBM traffic of the headline fused forward step (gridded A = NN(T), constant A), of the fused
# reverse step (constant A, Y table) and of the Y-table forward step; HBM traffic of the two HBM-bound kernels.  Separate --pmc passes,
# kernel-trace only (MI355X_MICROARCH.md).  -> gpurun_out/pmc_r06/ + gpurun_out/pmc_roofline.json (tools/pmc_r06.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
G=${1:-64}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_r06; rm -rf $O; mkdir -p $O
run() { # tag counters kernel law
  timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py $3 $G 1024 6 $4 > $O/$1.log 2>&1 || tail -2 $O/$1.log
}
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
BUSY="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for k in "fused_step nnA" "fused_step const" "fused_step nnY_tab" "adj_fused_step const" "adj_fused_step nnY_tab"; do
  set -- $k; t=$1_$2
  run ${t}_f64 "$F64" $1 $2
  run ${t}_busy "$BUSY" $1 $2
  run ${t}_fetch FETCH_SIZE $1 $2
  run ${t}_write WRITE_SIZE $1 $2
done
for k in "dhdt nnA" "rk_stage2 const"; do
  set -- $k; t=$1_$2
  run ${t}_fetch FETCH_SIZE $1 $2
  run ${t}_write WRITE_SIZE $1 $2
done
cd $R && python tools/pmc_r06.py $O $G > gpurun_out/pmc_roofline.json && cat gpurun_out/pmc_roofline.json | head -150

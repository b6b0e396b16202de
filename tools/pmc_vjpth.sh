#!/bin/bash
# the per-node theta-VJP of the U law (k_vjp_theta_nn<ArchDef>), 64 x 1024^2: time, fp64 instructions, VALU busy / waits, LDS -> gpurun_out/pmc_vjpth.txt
R=${GRAFT_REPO_ROOT:-$PWD}; G=${1:-64}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_vjpth; rm -rf $O; mkdir -p $O
run() { timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py vjp_theta $G 1024 6 nnU_tab > $O/$1.log 2>&1 || tail -2 $O/$1.log; }
run f64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
run busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run mem "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
cd $R && { grep us/launch $O/f64.log | tail -1; python tools/pmc_summary.py "$O/*/**/*counter_collection.csv" 2>&1 | grep "k_vjp_theta" | cut -c1-460; } > gpurun_out/pmc_vjpth.txt; cat gpurun_out/pmc_vjpth.txt

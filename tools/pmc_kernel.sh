# usage: pmc_kernel.sh <timed-kernel-name> <grep-pattern> [G] [n]   -- SQ / cache counters of one kernel
set -x
R=$GRAFT_REPO_ROOT
K=$1; PAT=$2; G=${3:-8}; N=${4:-1024}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmck; mkdir -p $R/gpurun_out/pmck
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmck/g$i -- python $R/tools/run_kernel.py $K $G $N 6 > $R/gpurun_out/pmck/g$i.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmck/**/*counter_collection.csv" 2>&1 | grep -i "$PAT"

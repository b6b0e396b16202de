# SQ / L1 counters of the reverse stage kernel of the tabulated U law: tools/pmc_utab.sh [kernel=adj_stage2] [G=8] [n=1024] [law=nnU_tab]
R=$GRAFT_REPO_ROOT
K=${1:-adj_stage2}; G=${2:-8}; N=${3:-1024}; LAW=${4:-nnU_tab}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmcu_$K_$LAW; rm -rf $O; mkdir -p $O
python $R/tools/run_kernel.py $K $G $N 20 $LAW
python $R/tools/run_kernel.py $K $G $N 20 const
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python $R/tools/run_kernel.py $K $G $N 6 $LAW > $O/g$i.log 2>&1 || tail -3 $O/g$i.log
done
cd $R && python tools/pmc_summary.py "$O/**/*counter_collection.csv" 2>&1 | grep -i "adj_stage\|rk_stage\|fused" | cut -c1-400

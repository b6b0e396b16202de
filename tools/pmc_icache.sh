# usage: pmc_icache.sh <timed-kernel-name> <grep-pattern> [lib]  -- instruction-fetch / I-cache counters of one kernel
R=$GRAFT_REPO_ROOT
K=$1; PAT=$2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmci; mkdir -p $R/gpurun_out/pmci
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmci/g$i -- python $R/tools/run_kernel.py $K 8 1024 6 > $R/gpurun_out/pmci/g$i.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmci/**/*counter_collection.csv" 2>&1 | grep -i "$PAT"

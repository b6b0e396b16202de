# HBM traffic per launch of ONE timed kernel: pmc_traffic_one.sh <timed-name> [G] [n]  (separate --pmc passes)
R=$GRAFT_REPO_ROOT
K=$1; G=${2:-8}; N=${3:-1024}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmct1; mkdir -p $R/gpurun_out/pmct1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmct1/${K}_$c -- python $R/tools/run_kernel.py $K $G $N 6 > $R/gpurun_out/pmct1/${K}_$c.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmct1/**/*counter_collection.csv" 2>&1 | grep -v "k_begin\|k_adj_begin\|k_controller\|k_poststep"

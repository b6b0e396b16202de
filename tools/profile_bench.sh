# one-stop profile refresh on the GPU box: bench.py un-profiled, then under rocprofv3 kernel-trace
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R && python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-grad-eval > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof.err
find $R/gpurun_out/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/bench_kernel_stats.csv \;
head -12 $R/gpurun_out/bench_kernel_stats.csv
cat $R/gpurun_out/bench.json

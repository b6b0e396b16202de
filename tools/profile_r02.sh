# round-2 profile refresh on the GPU box: bench.py un-profiled, bench.py under rocprofv3 --kernel-trace --stats,
# PMC roofline passes.  Outputs under gpurun_out/r02/ (copy into profiles/r02/ afterwards).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02; mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-grad-eval --no-hbm-sweep > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
head -8 $O/bench_kernel_stats.csv
cd $R && bash tools/pmc_roofline.sh > $O/pmc_roofline.log 2>&1
cp $R/gpurun_out/pmc_roofline.json $O/pmc_roofline.json
rm -rf $R/gpurun_out/pmcr

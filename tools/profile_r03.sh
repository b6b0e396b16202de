# round-3 profile refresh on the GPU box: bench.py un-profiled, bench.py under rocprofv3 --kernel-trace --stats WITH the HBM sweep
# (so that the 32 x 1024^2 launches behind roofline_hbm / roofline_per_stage are in the committed stats), PMC roofline passes.
# Outputs under gpurun_out/r03/ (copy into profiles/r03/ afterwards).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03; mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-grad-eval --no-full-config > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
head -12 $O/bench_kernel_stats.csv
# the same with the gradient evaluations of the bench line (8 x 1024^2: discrete and continuous adjoint, scalar and gridded law,
# 4 / 512 alpine glaciers) and without the HBM sweep: where a gradient evaluation spends its time
rm -rf $O/profg; rocprofv3 --kernel-trace --stats --output-format csv -d $O/profg -- python $R/bench.py --steps 40 --no-cpu-baseline --no-hbm-sweep --no-full-config > $O/grad_under_rocprof.json 2> $O/profg.err
find $O/profg -name "*kernel_stats.csv" -exec cp {} $O/grad_kernel_stats.csv \;
rm -rf $O/profg
cd $R && bash tools/pmc_roofline.sh > $O/pmc_roofline.log 2>&1
cp $R/gpurun_out/pmc_roofline.json $O/pmc_roofline.json
rm -rf $R/gpurun_out/pmcr

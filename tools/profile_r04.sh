# round-4 profile refresh on the GPU box: full GPU suite, bench.py un-profiled, bench.py under rocprofv3 --kernel-trace --stats (with the
# HBM sweep; then with the gradient evaluations), the other workflows (Y / U law), PMC passes.  Outputs under gpurun_out/r04/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
(time timeout 2400 python -m pytest tests -m gpu -q -n 6 --timeout 900) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-grad-eval > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
head -8 $O/bench_kernel_stats.csv
rm -rf $O/profg; rocprofv3 --kernel-trace --stats --output-format csv -d $O/profg -- python $R/bench.py --steps 40 --no-cpu-baseline --no-hbm-sweep --no-weak > $O/grad_under_rocprof.json 2> $O/profg.err
find $O/profg -name "*kernel_stats.csv" -exec cp {} $O/grad_kernel_stats.csv \;
rm -rf $O/profg
( for w in "Y 512 8" "NET Y 512 8" "U 512 8 scaled" "NET U 512 8 scaled" "U 512 8"; do
    rm -rf $O/profw
    e=""; if [ "${w%% *}" = NET ]; then e="ODINN_LAW_TABLE=0 ODINN_INTERP_ASYNC=0"; w=${w#NET }; fi  # (the Y law without its table and without the overlap)
    env $e rocprofv3 --kernel-trace --stats --output-format csv -d $O/profw -- python $R/tools/workflow_probe.py $w > $O/probe.txt 2>&1
    echo "=== $e python tools/workflow_probe.py $w"; grep -E "solve ms|LossH" $O/probe.txt
    python $R/tools/kstats.py $O/profw 14
    rm -rf $O/profw
  done ) > $O/workflows_kernel_stats.txt 2>&1
cat $O/workflows_kernel_stats.txt
cd $R && bash tools/pmc_r04.sh > $O/pmc_roofline.log 2>&1
tail -5 $O/pmc_roofline.log
rm -rf $R/gpurun_out/pmcr

R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_grad; mkdir -p $R/gpurun_out/prof_grad
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_grad -- python $R/tools/grad_eval_bench.py ${1:-512} > $R/gpurun_out/prof_grad/out.txt 2>&1
f=$(find $R/gpurun_out/prof_grad -name "*kernel_stats.csv" | head -1)
head -16 $f | cut -c1-150
tail -2 $R/gpurun_out/prof_grad/out.txt | cut -c1-300

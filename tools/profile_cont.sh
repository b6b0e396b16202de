# rocprofv3 kernel stats of one continuous-adjoint gradient evaluation batch (bench workload)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/profc; mkdir -p $R/gpurun_out/profc
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profc -- python $R/tools/grad_probe.py > $R/gpurun_out/profc.log 2>&1
find $R/gpurun_out/profc -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/cont_kernel_stats.csv \;
head -14 $R/gpurun_out/cont_kernel_stats.csv | cut -c1-200

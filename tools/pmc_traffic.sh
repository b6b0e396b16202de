# HBM traffic per launch of the main kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmct; mkdir -p $R/gpurun_out/pmct
for k in fused_step rk_stage2 vjp_h adj_stage2 euler_cfl dhdt vjp_theta; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmct/${k}_$c -- python $R/tools/run_kernel.py $k 8 1024 6 > $R/gpurun_out/pmct/${k}_$c.log 2>&1
  done
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmct/**/*counter_collection.csv" 2>&1 | grep -v "k_begin\|k_adj_begin\|k_controller\|k_poststep"

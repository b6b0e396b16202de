cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for sel in 1 0; do
  rm -rf $O/profY
  ODINN_INTERP_SELECT=$sel rocprofv3 --kernel-trace --stats --output-format csv -d $O/profY -- python $GRAFT_REPO_ROOT/tools/workflow_probe.py Y 512 8 > $O/probeY_sel$sel.txt 2>&1
  f=$(find $O/profY -name "*kernel_stats.csv" | head -1)
  echo "=== ODINN_INTERP_SELECT=$sel"; tail -3 $O/probeY_sel$sel.txt; python $GRAFT_REPO_ROOT/tools/kstats.py $f 22
  rm -rf $O/profY
done > $O/selY_kernel_stats.txt 2>&1
cat $O/selY_kernel_stats.txt

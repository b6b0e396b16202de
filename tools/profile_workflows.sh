# The other workflows at bench scale under rocprofv3 --kernel-trace --stats (run on the GPU box via gpurun):
# wall-clock lines of the probes + top kernels -> gpurun_out/r02w/workflows_kernel_stats.txt (copy into profiles/r03/).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03w; mkdir -p $O
OUT=$O/workflows_kernel_stats.txt
{
  echo "# Other workflows under rocprofv3 --kernel-trace --stats at bench scale (tools/profile_workflows.sh: workflow_probe.py,"
  echo "# velocity_probe.py, agg_probe.py; top kernels by total time via tools/kstats.py).  Each block: the probe's own wall-clock"
  echo "# lines (ms per gradient evaluation of the whole batch; reverse accept / reject counts), then kernel totals over the whole"
  echo "# probe run (warm-up calls included)."
} > $OUT
cd /tmp && export TMPDIR=/tmp
for w in "workflow_probe.py gridded" "velocity_probe.py" "agg_probe.py" "workflow_probe.py mb" "workflow_probe.py Y 512 8" "workflow_probe.py U 512 8"; do
  n=$(echo $w | tr " ./" "___")
  rm -rf /tmp/p_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -- python $R/tools/$w > /tmp/p_$n.log 2>/dev/null
  { echo; echo "=== python tools/$w"; grep -v "^W20\|^E20\|rocprof" /tmp/p_$n.log | cut -c1-150; python $R/tools/kstats.py /tmp/p_$n 14; } >> $OUT
done
tail -5 $OUT

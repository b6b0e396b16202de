# A/B of the tabulated Y law (odinn_schedule.law_table): its tests, the Y-law workflow probe with the network and with the table,
# and the table run under rocprofv3 --kernel-trace --stats  ->  gpurun_out/ytab_*.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
N=${1:-512}; G=${2:-8}
timeout 900 python -m pytest tests/test_gpu_law_table.py -x -q 2>&1 | tail -30 > gpurun_out/ytab_test.txt
(echo "== network"; ODINN_SCHEDULE=law_table=0,interp_async=0 timeout 900 python tools/workflow_probe.py Y $N $G; echo "== table"; timeout 900 python tools/workflow_probe.py Y $N $G) > gpurun_out/ytab_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_ytab
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ytab -- python $R/tools/workflow_probe.py Y $N $G > /tmp/p_ytab.log 2>/dev/null
{ cat /tmp/p_ytab.log | grep -v "^W20\|^E20\|rocprof" | cut -c1-150; python $R/tools/kstats.py /tmp/p_ytab 24; } > $R/gpurun_out/ytab_kstats.txt
cd $R; cat gpurun_out/ytab_test.txt gpurun_out/ytab_probe.txt gpurun_out/ytab_kstats.txt

#!/bin/bash
# kernel shares of the U law's gradients: rocprofv3 --kernel-trace --stats of tools/workflow_probe.py U n G scaled under ODINN_ADJ_UT_FUSED=$3
n=${1:-1024}; G=${2:-16}; m=${3:-2}
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/ut
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/utp
ODINN_SCHEDULE=adj_ut_fused=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/utp -- python $R/tools/workflow_probe.py U $n $G scaled > $R/gpurun_out/ut/probe_${n}_${G}_m$m.txt 2>&1
f=$(find /tmp/utp -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/ut/kstats_${n}_${G}_m$m.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("mode $m  $G x $n^2: total kernel ms %.1f" % (tot/1e6))
for r in rows[:12]:
    print('  %-80s calls %6s avg %9.1f us  tot %8.1f ms %5s%%' % (r['Name'][:80].replace('void odinn::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
tail -3 $R/gpurun_out/ut/probe_${n}_${G}_m$m.txt

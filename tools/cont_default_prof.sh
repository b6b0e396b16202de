#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/cdp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cdp -- python $R/tools/cont_default_probe.py ${1:-1024} ${2:-64} ${3:-A} > /tmp/cdp.log 2>&1
grep -v "^[EWI]2026" /tmp/cdp.log | tail -12
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/cdp/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms (2 evaluations) %.1f" % (tot/1e6))
for r in rows[:14]:
    print('  %-84s calls %6s avg %9.1f us  tot %8.1f ms %5s%%' % (r['Name'][:84].replace('void odinn::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY

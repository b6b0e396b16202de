set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_small; mkdir -p $R/gpurun_out/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -- python $R/tools/grad_eval_bench.py 4 > $R/gpurun_out/prof_small/out.txt 2>&1
f=$(find $R/gpurun_out/prof_small -name "*kernel_stats.csv" | head -1)
head -20 $f
t=$(find $R/gpurun_out/prof_small -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# look at a window of forward steps in the middle: print 40 consecutive kernels with durations and gaps
k0=len(rows)//3
prev=None
for r in rows[k0:k0+40]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev) if prev else 0
    print(f'{r["Kernel_Name"][:60]:60s} dur={(e-s)/1e3:7.1f}us gap={gap/1e3:6.1f}us')
    prev=e
PY

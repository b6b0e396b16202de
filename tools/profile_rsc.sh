# kernel durations and gaps inside the reverse loop of a small continuous-adjoint gradient: bash tools/profile_rsc.sh "<rsc_probe args>" <tag>
set -x
R=$GRAFT_REPO_ROOT
ARGS=${1:-alpine:4}; TAG=${2:-rsc}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG; mkdir -p $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/tools/rsc_probe.py $ARGS > $R/gpurun_out/prof_$TAG/out.txt 2>&1
cat $R/gpurun_out/prof_$TAG/out.txt | tail -3
python $R/tools/kstats.py $R/gpurun_out/prof_$TAG 12
t=$(find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last quarter of the trace is the self-controlled run's last gradient: 30 consecutive kernels with durations and gaps
for k0 in (len(rows)//3, len(rows)-400):
    prev=None
    for r in rows[k0:k0+24]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        gap=(s-prev) if prev else 0
        print(f'{r["Kernel_Name"][:70]:70s} dur={(e-s)/1e3:7.1f}us gap={gap/1e3:6.1f}us')
        prev=e
    print("----")
PY
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete

# stand-in experiments for tests/test_gpu_reference_matrix.py: bash tools/refm_probe.sh "<pytest -k expression>" "VAR=.. VAR=.." ...
cd $GRAFT_REPO_ROOT
K="$1"; shift
for v in "$@"; do
  rm -f gpurun_out/reference_matrix.jsonl
  (time env $v timeout 900 python -m pytest tests/test_gpu_reference_matrix.py -q --timeout 600 -k "$K") 2>&1 | grep -E "passed|failed|real"
  echo "== $v"
  python - <<P
import json
for l in open("gpurun_out/reference_matrix.jsonl"):
    r=json.loads(l); t=r["thres"]
    ok = abs(r["ratio"])<t[0] and abs(r["angle"])<t[1] and r["relerr"]<t[2]
    print("%-48s %9.2e %9.2e %9.2e | %s %s" % (r["row"], r["ratio"], r["angle"], r["relerr"], t, "ok" if ok else "FAIL"))
P
done

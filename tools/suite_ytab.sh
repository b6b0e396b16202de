cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for env in "ODINN_DUMMY=0" "ODINN_LAW_TABLE=0"; do
  echo "== $env"
  env $env python -m pytest tests -m gpu -q -n 6 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -40
done > gpurun_out/suite_ytab.txt 2>&1
cat gpurun_out/suite_ytab.txt
bash tools/fuzz_big.sh 7200:8700

cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 6 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -10
bash tools/fuzz_big.sh 8700:10200 2>&1 | tail -14

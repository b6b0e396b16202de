cd $GRAFT_REPO_ROOT
python tools/workflow_probe.py Y 512 8 2>&1 | grep -E "solve ms|LossH"
python -m pytest tests -m gpu -q -n 6 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -10
ODINN_LAW_TABLE=0 python -m pytest tests -m gpu -q -n 6 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -10

cd $GRAFT_REPO_ROOT
ODINN_FUZZ_AUDIT=$PWD/gpurun_out/fz19681.jsonl ODINN_FUZZ_SEEDS=19681:19682 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "time_aggregated" -p no:cacheprovider --timeout 1500 2>&1 | grep -E "^E  |passed|failed|skipped|Error" | cut -c1-2500 | head -30
cat gpurun_out/fz19681.jsonl 2>/dev/null

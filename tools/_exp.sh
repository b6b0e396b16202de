cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
  echo "== ODINN_INTERP_BATCH=0 run $i"
  ODINN_INTERP_BATCH=0 python -m pytest tests -m gpu -q -n 6 --deselect tests/test_gpu_schedule.py 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|^E  " | cut -c1-900 | head -8
done

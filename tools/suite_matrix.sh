# the whole GPU suite under the non-default kernel schedules (environment overrides): tools/suite_matrix.sh > gpurun_out/suite_matrix.log
cd $GRAFT_REPO_ROOT
for env in "ODINN_SCHEME=1" "ODINN_STEP_SC=0" "ODINN_STEP_SC=1" "ODINN_ADJ_FUSED=0" "ODINN_SNAP_ON_LOAD=0" "ODINN_VJPH_STRIP=0 ODINN_VJPTH_STRIP=0 ODINN_DHDT_STRIP=0" "ODINN_FUSED_TILES=l" "ODINN_ADJ_ROWS=7" "ODINN_INTERP_BATCH=0" "ODINN_LAW_TABLE=0" "ODINN_INTERP_ASYNC=0" "ODINN_ADJ_SC=0" "ODINN_ADJ_SC=1" "ODINN_ADJ_ROWS=2" "ODINN_INTERP_SELECT=0" "ODINN_UTAB_LEVEL=3" "ODINN_UT_LDS=0"; do
  echo "== $env"
  env $env python -m pytest tests -m gpu -q -x -n 6 --deselect tests/test_gpu_schedule.py 2>&1 | grep -E "passed|failed|error" | tail -2
done

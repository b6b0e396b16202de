# the whole GPU suite under the non-default kernel schedules (ODINN_SCHEDULE / the other override variables): tools/suite_matrix.sh > gpurun_out/suite_matrix.log
cd ${GRAFT_REPO_ROOT:-.}
for env in "ODINN_SCHEDULE=scheme=1" "ODINN_SCHEDULE=step_sc=0" "ODINN_SCHEDULE=step_sc=1" "ODINN_SCHEDULE=adj_fused=0" "ODINN_SCHEDULE=snap_on_load=0" \
           "ODINN_SCHEDULE=vjph_strip=0,vjpth_strip=0,dhdt_strip=0" "ODINN_SCHEDULE=fused_tiles=l" "ODINN_SCHEDULE=adj_rows=7" "ODINN_SCHEDULE=interp_batch=0" \
           "ODINN_SCHEDULE=law_table=0" "ODINN_SCHEDULE=interp_async=0" "ODINN_SCHEDULE=adj_sc=0" "ODINN_SCHEDULE=adj_sc=1" "ODINN_SCHEDULE=adj_rows=2" \
           "ODINN_SCHEDULE=adj_ut_fused=0" "ODINN_INTERP_SELECT=0" "ODINN_UTAB_LEVEL=3" "ODINN_UT_LDS=0"; do
  echo "== $env"
  env $env python -m pytest tests -m gpu -q -x -n 6 --deselect tests/test_gpu_schedule.py --deselect tests/test_gpu_determinism.py 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | cut -c1-300 | tail -6
done

#!/bin/bash
# U law (target :D) through its table, continuous gradient: staged (0) / strip UT (1) / LDS tiles (2): parity test + timings
mkdir -p gpurun_out/ut
python -m pytest tests/test_gpu_law_table_U.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15
for m in 0 2; do
  echo "== ODINN_ADJ_UT_FUSED=$m: 8 x 512^2"; ODINN_SCHEDULE=adj_ut_fused=$m python tools/workflow_probe.py U 512 8 scaled 2>&1 | tail -4
done
for m in 0 2; do
  echo "== ODINN_ADJ_UT_FUSED=$m: 16 x 1024^2"; ODINN_SCHEDULE=adj_ut_fused=$m python tools/workflow_probe.py U 1024 16 scaled 2>&1 | tail -4
done

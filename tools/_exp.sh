cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_law_table_U.py -x -q 2>&1 | tail -3
python -m pytest tests/ -m gpu -x -q -k "U_law or lawU or law_U or D_target or _U_ or pure" 2>&1 | tail -3
for k in adj_stage2 vjp_H rk_stage2; do python tools/run_kernel.py $k 8 1024 20 nnU_tab 2>&1 | tail -1; done
python tools/workflow_probe.py U 512 8 scaled

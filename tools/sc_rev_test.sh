cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_continuous_adjoint.py -x -q -k "self_controlled_reverse or fused_reverse" 2>&1 | tail -30 > gpurun_out/sc_test.txt
cat gpurun_out/sc_test.txt

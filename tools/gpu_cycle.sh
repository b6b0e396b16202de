# one GPU cycle: full GPU test suite + default bench (+ optional extra command); everything into gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.txt
(time timeout 900 python bench.py ${BENCH_ARGS:-}) > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -4 gpurun_out/pytest_gpu.txt
python - <<'P'
import json
d = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
a = d['aux']; r = d['roofline']
print('value %.3e  ms/step %.4f  fused_nn %.4f ms frac %.3f | constA step %.4f kernel %.4f | law_field %.3f ms' % (
    d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'], a['constA_ms_per_step'], a['constA_fused_step_kernel_ms'], a['law_field_ms']))
print('adj_fused %.4f  nn_inlined %.3f  grad' % (a['adj_fused_step_ms'], a.get('nn_inlined_2x16_ms_per_step', -1)), json.dumps(d['grad_evals_per_s'].get('bench_workload', {}))[:200])
P

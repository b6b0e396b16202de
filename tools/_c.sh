cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 6 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -8
ODINN_FUZZ_SEEDS=13200:14200 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 --timeout 300 -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
python bench.py --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1]); g=d.get('grad_evals_per_s') or d['aux'].get('grad_evals_per_s'); y=g.get('bench_workload_Y_law'); print(d['value'], y['discrete_adjoint'], y['continuous_adjoint'])"

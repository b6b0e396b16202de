cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -3 gpurun_out/bench_now.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); g=d.get('grad_evals_per_s') or d['aux'].get('grad_evals_per_s'); print(json.dumps(g.get('bench_workload_Y_law'), indent=1))"
python -m pytest tests/test_gpu_law_table.py -q 2>&1 | tail -2

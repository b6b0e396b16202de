cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1]); g=d.get('grad_evals_per_s') or d['aux'].get('grad_evals_per_s'); y=g.get('bench_workload_Y_law'); print(y['discrete_adjoint'], y['continuous_adjoint'])"

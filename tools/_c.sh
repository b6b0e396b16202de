cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -3 gpurun_out/bench_now.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps({k:v for k,v in d['aux']['roofline_adjoint_nn']['laws'].items() if 'table' in k}, indent=1))"

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=gpurun_out/r04
(time timeout 1800 python -m pytest tests -m gpu -q -n 6 --timeout 900 -k "baseline or velocity or abi or two_rank or rccl") > $O/pytest_f.txt 2>&1; tail -5 $O/pytest_f.txt
(time timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r04/bench.json').read().strip().splitlines()[0])
print('value %.4e ms %.4f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print('adj', {k: d['roofline_adjoint'][k] for k in ('ms_per_launch', 'achieved', 'frac', 'traffic_over_algorithmic')})
print('adj_nn', json.dumps(d['aux'].get('roofline_adjoint_nn', {}).get('laws'))[:700])
print('nn_inl', {k: d['aux']['roofline_nn_inlined'][k] for k in ('flop_per_cell_stage', 'achieved', 'frac', 'ms_per_step')})
print('hbm traffic', d['roofline_hbm']['traffic'], d['roofline_per_stage']['traffic'])
P
(time ODINN_BENCH_BACKEND=gloo ODINN_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --no-cpu-baseline --no-hbm-sweep) > $O/bench_2rank_dry_run.json 2> $O/bench2.err; echo "bench2 rc=$?"; tail -3 $O/bench2.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r04/bench_2rank_dry_run.json').read().strip().splitlines()[0])
print('2-rank dry run: value %.4e n_gpus %d scaling %s' % (d['value'], d['n_gpus'], d['scaling']), d['config']['parallelism'][:200])
print(json.dumps(d['grad_evals_per_s']['bench_workload'])[:400])
P

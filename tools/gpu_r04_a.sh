# round 4, first GPU cycle: full GPU suite (xdist), the new bench line, the fuzz-skip audit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=gpurun_out/r04
(time timeout 2400 python -m pytest tests -m gpu -q -n 6 --timeout 900 -x) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
(time timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'P'
import json
try:
    d = json.loads(open('gpurun_out/r04/bench.json').read().strip().splitlines()[0])
    a = d['aux']; r = d['roofline']
    print('value %.3e  ms/step %.4f  fused_nn %.4f ms frac %.3f | constA step %.4f | weak' % (d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'], a['constA_ms_per_step']), json.dumps(a.get('weak_8_per_gpu'))[:300])
    print('grad', json.dumps(d['grad_evals_per_s'])[:1200])
    print('cpu', json.dumps(d['cpu_baseline'])[:400])
    print('hbm', json.dumps({k: round(v.get('frac_of_hbm_peak', 0), 3) for k, v in a['hbm_past_infinity_cache'].get('kernels', {}).items()}))
except Exception as e:
    print('bench parse failed', e)
P
rm -f $O/fuzz_skips.jsonl
(time ODINN_FUZZ_AUDIT=$PWD/$O/fuzz_skips.jsonl ODINN_FUZZ_SEEDS=0:1200 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 --timeout 240 \
   -k "gradient_matches or velocity_loss_gradient or time_aggregated") > $O/fuzz_audit_pytest.txt 2>&1
tail -4 $O/fuzz_audit_pytest.txt
python tools/fuzz_audit.py $O/fuzz_skips.jsonl > $O/fuzz_skips.txt 2>&1; cat $O/fuzz_skips.txt | head -60

# round 4, cycle b: parity tests of the per-node-network laws with the perturbed evaluation, A/B of the Y / U workflows against
# the library of the previous commit (ab/libodinn_base.so), fuzz-skip audit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=gpurun_out/r04
(time timeout 1500 python -m pytest tests -m gpu -q -n 6 --timeout 900 -k "nn or law or Y_ or U_ or velocity or continuous or fuzz or golden or inlined") > $O/pytest_nn.txt 2>&1; tail -5 $O/pytest_nn.txt
for w in "Y 512 8" "U 512 8"; do
  for lib in base new; do
    if [ $lib = base ]; then export ODINN_LIB=$PWD/ab/libodinn_base.so; else unset ODINN_LIB; fi
    echo "== $w $lib"; timeout 600 python tools/workflow_probe.py $w 2>&1 | tail -3
  done
done > $O/ab_nn_workflows.txt 2>&1
unset ODINN_LIB
cat $O/ab_nn_workflows.txt
rm -f $O/fuzz_skips.jsonl
(time ODINN_FUZZ_AUDIT=$PWD/$O/fuzz_skips.jsonl ODINN_FUZZ_SEEDS=0:1200 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 --timeout 240 \
   -k "gradient_matches or velocity_loss_gradient or time_aggregated") > $O/fuzz_audit_pytest.txt 2>&1
tail -4 $O/fuzz_audit_pytest.txt
python tools/fuzz_audit.py $O/fuzz_skips.jsonl > $O/fuzz_skips.txt 2>&1; head -60 $O/fuzz_skips.txt

# round 4, cycle d: full GPU suite (split reverse kernels; one composite-key sort in the Y law's `:Linear` interpolation); A/B of the
# Y / U workflows against the previous round's library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=gpurun_out/r04
(time timeout 2400 python -m pytest tests -m gpu -q -n 6 --timeout 900 -k "interp or Y_law or golden or fuzz or schedule or hybrid") > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
for w in "Y 512 8" "U 512 8"; do
  for lib in base new; do
    unset ODINN_LIB
    if [ $lib = base ]; then export ODINN_LIB=$PWD/ab/libodinn_base.so; fi
    echo "== $w $lib"; timeout 600 python tools/workflow_probe.py $w 2>&1 | tail -3
  done
done > $O/ab_nn_workflows.txt 2>&1
unset ODINN_LIB
cat $O/ab_nn_workflows.txt

# round 4, cycle c: A/B of the variants of the perturbed evaluation on the Y / U workflows (same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=gpurun_out/r04
for w in "Y 512 8"; do
  for lib in base ty0 ty1g5 ty1g2 ty1g1; do
    export ODINN_LIB=$PWD/ab/libodinn_$lib.so
    echo "== $w $lib"; timeout 600 python tools/workflow_probe.py $w 2>&1 | tail -3
  done
done > $O/ab_nn_variants2.txt 2>&1
unset ODINN_LIB
cat $O/ab_nn_variants2.txt

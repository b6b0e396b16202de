#!/bin/bash
# BIG fuzz draws whose xdist worker aborted inside loss_grad_continuous (seeds 32320, 32407 of the aggregated-terms test): each seed
# in $2 concurrent processes, $3 rounds, stderr kept
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/abort; rm -rf $O; mkdir -p $O
SEEDS=${1:-"32320 32407"}; NP=${2:-16}; NR=${3:-2}
for r in $(seq 1 $NR); do
  for s in $SEEDS; do
    for p in $(seq 1 $NP); do
      ( ODINN_FUZZ_BIG=1 ODINN_FUZZ_SEEDS=$s:$((s+1)) timeout 300 python -X faulthandler -m pytest "tests/test_gpu_fuzz.py::test_random_batch_time_aggregated_terms_match_the_oracle" -m gpu -q -x --timeout 250 -p no:cacheprovider > $O/s${s}_r${r}_p${p}.log 2>&1; echo "seed $s round $r proc $p rc=$?" >> $O/rc.txt ) &
    done
  done
  wait
done
sort $O/rc.txt | awk '{print $NF}' | sort | uniq -c
grep -l -i "fault\|abort\|terminate" $O/*.log | head -5
f=$(grep -l -i "fault\|abort\|terminate" $O/*.log | head -1); [ -n "$f" ] && grep -i -B2 -A12 "fault\|terminate\|Aborted" $f | head -60

#!/bin/bash
# cross-process determinism of the self-controlled forward loop on a fuzz draw: N fresh processes per mode, then per-glacier traces
SEED=${1:-24379}; N=${2:-6}
mkdir -p gpurun_out/sc
export SC_REPRO_VERBOSE=1
python tools/sc_repro.py $SEED step_sc=0 2>&1 | tail -3 > gpurun_out/sc/base_$SEED.txt
unset SC_REPRO_VERBOSE
for i in $(seq 1 $N); do python tools/sc_repro.py $SEED step_sc=1 2>&1 | tail -1; done > gpurun_out/sc/sc1_$SEED.txt
for i in $(seq 1 3); do python tools/sc_repro.py $SEED step_sc=0 2>&1 | tail -1; done > gpurun_out/sc/sc0_$SEED.txt
for g in 0 1 2; do
  for m in 0 1; do
    ODINN_TRACE_STEPS=64 ODINN_TRACE_GLACIER=$g python tools/sc_repro.py $SEED step_sc=$m > gpurun_out/sc/trace_${SEED}_g${g}_sc${m}.txt 2>&1
  done
done
cat gpurun_out/sc/base_$SEED.txt gpurun_out/sc/sc1_$SEED.txt gpurun_out/sc/sc0_$SEED.txt

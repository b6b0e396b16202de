#!/bin/bash
# build_nn_variant.sh <name> <extra -D flags...>: A/B library ab/libodinn_<name>.so that differs from the in-tree library only by
# the given macros in the reverse kernels with an inlined network (k_adj.hip, law modes 3..5)
set -e
cd "$(dirname "$0")/../odinn.jl_amd/csrc"
name=$1; shift
out=../../ab/obj_$name; mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
for lm in 3 4 5; do hipcc $F -DODINN_LM=$lm -c k_adj.hip -o $out/k_adj$lm.o & done
wait
objs=""
for o in odinn_hip k_misc k_vel k_interp k_fwd0 k_fwd1 k_fwd2 k_fwd3 k_fwd4 k_fwd5 k_adj0 k_adj1 k_adj2 k_fused0 k_fused1 k_fused2 k_fused3 k_fused4 k_fused5 k_adjf; do objs="$objs $o.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libodinn_$name.so $objs $out/k_adj3.o $out/k_adj4.o $out/k_adj5.o -ldl
rm -rf $out
echo built ab/libodinn_$name.so

#!/bin/bash
# build_variant.sh <name> <extra -D flags...> : A/B library variants/libodinn_<name>.so (git-ignored) that
# differs from the in-tree library only by the given macros in the fused-step kernels and the host TU
set -e
cd "$(dirname "$0")/../odinn.jl_amd/csrc"
name=$1; shift
out=../../variants/obj_$name; mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
hipcc $F -c odinn_hip.hip -o $out/odinn_hip.o &
hipcc $F -DODINN_LM=0 -c k_fused.hip -o $out/k_fused0.o &
hipcc $F -DODINN_LM=1 -c k_fused.hip -o $out/k_fused1.o &
hipcc $F -c k_adjf.hip -o $out/k_adjf.o &
hipcc $F -c k_adjfs.hip -o $out/k_adjfs.o &
wait
objs=""
for o in k_misc k_vel k_interp k_fwd0 k_fwd1 k_fwd2 k_fwd3 k_fwd4 k_fwd5 k_fwd6 k_fwd7 k_fwd8 k_adj0 k_adj1 k_adj2 k_adj3 k_adj4 k_adj5 k_adj6 k_adj7 k_adj8 k_fused2 k_fused3 k_fused4 k_fused5 k_fused6 k_fused7 k_fused8; do objs="$objs $o.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libodinn_$name.so $out/odinn_hip.o $out/k_fused0.o $out/k_fused1.o $out/k_adjf.o $out/k_adjfs.o $objs -ldl
echo built variants/libodinn_$name.so

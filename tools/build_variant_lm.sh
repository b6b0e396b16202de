#!/bin/bash
# build_variant_lm.sh <name> <lm> <extra -D flags...> : variants/libodinn_<name>.so (git-ignored) that differs from the in-tree library
# only by the given macros in the law-mode <lm> translation units (k_fwd<lm>, k_adj<lm>, k_fused<lm>)
set -e
cd "$(dirname "$0")/../odinn.jl_amd/csrc"
name=$1; lm=$2; shift; shift
out=../../abvar/obj_$name; mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
hipcc $F -DODINN_LM=$lm -c k_fwd.hip -o $out/k_fwd$lm.o 2>/dev/null &
hipcc $F -DODINN_LM=$lm -c k_adj.hip -o $out/k_adj$lm.o 2>/dev/null &
hipcc $F -DODINN_LM=$lm -c k_fused.hip -o $out/k_fused$lm.o 2>/dev/null &
wait
objs=""
for o in odinn_hip k_misc k_vel k_interp k_adjf k_adjfs; do objs="$objs $o.o"; done
for k in k_fwd k_adj k_fused; do for i in 0 1 2 3 4 5 6 7 8; do if [ $i = $lm ]; then objs="$objs $out/$k$i.o"; else objs="$objs $k$i.o"; fi; done; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abvar/libodinn_$name.so $objs -ldl
echo built variants/libodinn_$name.so

#!/bin/bash
# asm_variant.sh <device.s> <out.o> [host -D flags...]: build the k_fused.hip (ODINN_LM=0) object from a (hand-edited) gfx950 assembly
# listing of its device side (hipcc -S --cuda-device-only): assemble, link the code object, bundle, compile the host side around it.
set -e
S=$1; OUT=$2; shift 2
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$S" -o $T/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
cd "$(dirname "$0")/../odinn.jl_amd/csrc"
hipcc --offload-arch=gfx950 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -O3 -std=c++17 -fPIC -w -DODINN_LM=0 "$@" -c k_fused.hip -o "$OUT"
rm -rf $T

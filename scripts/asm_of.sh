#!/bin/bash
# Device assembly of one translation unit of rodeo_amd/csrc (for reading what hipcc made of a kernel):
#   scripts/asm_of.sh solve_dense.hip /tmp/out.s [extra hipcc flags]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$1; out=$2; shift 2
cd "$root/rodeo_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form \
    -falign-loops=64 -I../../include --cuda-device-only -S "$src" -o "$out" "$@" 2>&1 | grep -v "argument unused" || true

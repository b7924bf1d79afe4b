#!/bin/bash
# Experiment builds of the chkrebtii forward step (solve_tile3_kernels.hpp): what parts of the step cost.  Builds
# rodeo_amd/librodeo_kalman_abl_<name>.so with -D<name> (results are WRONG by construction); run
#   RK_LIB_PATH=rodeo_amd/librodeo_kalman_abl_<name>.so python scripts/bench_configs.py c4
#   names: RK_T3_ABLATE=1 (one-wave kernel without generator), RK_T3_ABLATE=2 (... and without square root),
#          RK_T3_ABLATE_COV (split kernel: no hand-off writes), RK_T3_ABLATE_MEAN (split kernel: mean wave idle),
#          RK_T3_ABLATE_COVIDLE (covariance wave idle); several macros joined by commas: RK_T3_ABLATE=1,RK_T3_ABLATE_MEAN
set -e
cd "$(dirname "$0")/../rodeo_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form -falign-loops=64 -I../../include"
for k in "$@"; do
  n=$(echo $k | tr '=,' '__')
  /opt/rocm/bin/hipcc $FLAGS $(echo $k | sed 's/,/ -D/g; s/^/-D/') -c solve_tile3.hip -o build/solve_tile3_abl_$n.o &
done
wait
for k in "$@"; do
  n=$(echo $k | tr '=,' '__')
  /opt/rocm/bin/hipcc $(for f in *.hip; do [ "$f" != solve_tile3.hip ] && echo build/${f%.hip}.o; done) build/solve_tile3_abl_$n.o -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -lhiprtc -Wl,-rpath,/opt/rocm/lib -o ../librodeo_kalman_abl_$n.so
done

#!/bin/bash
# GPU box: time C5 (N = 50) with variant builds of the dense kernels (make -C rodeo_amd/csrc variant NAME=.. DEFS=..).
#   scripts/c5_variant.sh <timing-lib-name|-> [<stamps-lib-name>] [extra bench_configs args]
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
t=$1; s=$2; shift; shift
if [ "$t" != "-" ]; then RK_LIB_PATH=$root/rodeo_amd/librodeo_kalman_$t.so python3 scripts/bench_configs.py c5 --c5-steps 50 "$@" 2>&1 | cut -c1-330 | tee $out/var_$t.json; fi
if [ -n "$s" ] && [ "$s" != "-" ]; then RK_DENSE_STAMPS=1 RK_LIB_PATH=$root/rodeo_amd/librodeo_kalman_$s.so python3 scripts/bench_configs.py c5 --c5-steps 50 "$@" 2>&1 | grep phase | tee $out/var_${s}_stamps.txt; fi

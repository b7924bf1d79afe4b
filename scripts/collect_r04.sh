#!/bin/bash
# Runs on the GPU box (via gpurun): the round-4 evidence set, in two parts so that each fits one gpurun call.
#   scripts/collect_r04.sh <tag> a     headline (bench.py, kernel stats, PMC traffic), C3 / C4 full size + kernel stats, n_deriv sweep
#   scripts/collect_r04.sh <tag> b     C5 both forms: full size, kernel stats, MFMA counters, PMC traffic (panel-loop LU and
#                                      the register-resident LU), phase stamps
# Everything lands under gpurun_out/<tag>_*; scripts/r04_to_profiles.py turns it into profiles/<tag>_*.
set -e
tag=${1:-r04}
part=${2:-a}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
prof() { # prof <name> <rocprofv3 args...> -- <script args...>: rocprofv3 from /tmp, program directly behind "--"
  local name=$1; shift
  ( cd /tmp && rocprofv3 "$@" > $out/${tag}_${name}.log 2>&1 )
}
if [ "$part" = a ]; then
  scripts/collect_profiles.sh $tag > $out/${tag}_collect_profiles.log 2>&1
  python3 scripts/bench_configs.py c3 c4 > $out/${tag}_configs_c3_c4.jsonl 2> $out/${tag}_configs_c3_c4.err
  python3 scripts/bench_configs.py c4 --c4-unfused > $out/${tag}_c4_unfused.jsonl 2> $out/${tag}_c4_unfused.err
  for c in c3 c4; do
    prof ${c}_stats --kernel-trace --stats --output-format csv -d $out/${tag}_${c}_stats -o stats -- python3 $root/scripts/bench_configs.py $c
  done
  prof c4_fetch --pmc FETCH_SIZE --output-format csv -d $out/${tag}_c4_fetch -o fetch -- python3 $root/scripts/bench_configs.py c4
  prof c4_write --pmc WRITE_SIZE --output-format csv -d $out/${tag}_c4_write -o write -- python3 $root/scripts/bench_configs.py c4
  prof c3_fetch --pmc FETCH_SIZE --output-format csv -d $out/${tag}_c3_fetch -o fetch -- python3 $root/scripts/bench_configs.py c3
  prof c3_write --pmc WRITE_SIZE --output-format csv -d $out/${tag}_c3_write -o write -- python3 $root/scripts/bench_configs.py c3
  # C4 forward step: what the generator and the square root cost (experiment builds of scripts/c4_ablation.sh, results wrong by construction)
  for abl in RK_T3_ABLATE_1 RK_T3_ABLATE_2 RK_T3_ABLATE_SIMZ; do
    if [ -f rodeo_amd/librodeo_kalman_abl_$abl.so ]; then
      RK_LIB_PATH=$root/rodeo_amd/librodeo_kalman_abl_$abl.so python3 scripts/bench_configs.py c4 > $out/${tag}_c4_$abl.jsonl 2> $out/${tag}_c4_$abl.err || true
    fi
  done
  RK_T4_BWD=quad python3 scripts/bench_configs.py c3 > $out/${tag}_c3_quad.jsonl 2> $out/${tag}_c3_quad.err
  python3 scripts/nderiv_times.py > $out/${tag}_nderiv_times.jsonl 2> $out/${tag}_nderiv_times.err
  python3 scripts/block_vs_dense_times.py > $out/${tag}_block_vs_dense.jsonl 2> $out/${tag}_block_vs_dense.err || true
else
  python3 scripts/bench_configs.py c5 --c5-check > $out/${tag}_c5_standard_full.json 2> $out/${tag}_c5_standard_full.err
  python3 scripts/bench_configs.py c5 --c5-kalman square-root --c5-check > $out/${tag}_c5_sqrt_full.json 2> $out/${tag}_c5_sqrt_full.err
  prof c5_stats --kernel-trace --stats --output-format csv -d $out/${tag}_c5_stats -o stats -- python3 $root/scripts/bench_configs.py c5 --c5-steps 200
  prof c5sq_stats --kernel-trace --stats --output-format csv -d $out/${tag}_c5sq_stats -o stats -- python3 $root/scripts/bench_configs.py c5 --c5-steps 200 --c5-kalman square-root
  CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES"
  prof c5_mfma --pmc $CNT --output-format csv -d $out/${tag}_c5_mfma -o mfma -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50
  prof c5sq_mfma --pmc $CNT --output-format csv -d $out/${tag}_c5sq_mfma -o mfma -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50 --c5-kalman square-root
  for cnt in FETCH_SIZE WRITE_SIZE; do
    prof c5_$cnt --pmc $cnt --output-format csv -d $out/${tag}_c5_$cnt -o pmc -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50
    prof c5sq_$cnt --pmc $cnt --output-format csv -d $out/${tag}_c5sq_$cnt -o pmc -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50 --c5-kalman square-root
    RK_DENSE_LU=regs prof c5regs_$cnt --pmc $cnt --output-format csv -d $out/${tag}_c5regs_$cnt -o pmc -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50
  done
  RK_DENSE_LU=regs python3 scripts/bench_configs.py c5 --c5-steps 50 > $out/${tag}_c5_regs_lu_N50.json 2> $out/${tag}_c5_regs_lu_N50.err
  python3 scripts/bench_configs.py c5 --c5-steps 50 > $out/${tag}_c5_N50.json 2> $out/${tag}_c5_N50.err
  if [ -f rodeo_amd/librodeo_kalman_stamps.so ]; then
    RK_DENSE_STAMPS=1 RK_LIB_PATH=$root/rodeo_amd/librodeo_kalman_stamps.so python3 scripts/bench_configs.py c5 --c5-steps 50 > $out/${tag}_c5_stamps.txt 2>&1 || true
    RK_DENSE_STAMPS=1 RK_LIB_PATH=$root/rodeo_amd/librodeo_kalman_stamps.so python3 scripts/bench_configs.py c5 --c5-steps 50 --c5-kalman square-root > $out/${tag}_c5sq_stamps.txt 2>&1 || true
  fi
fi
ls $out | grep "^${tag}_" | head -80

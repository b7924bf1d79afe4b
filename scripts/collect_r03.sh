#!/bin/bash
# Runs on the GPU box (via gpurun): the round-3 evidence set.  Everything lands under gpurun_out/<tag>_*.
#   scripts/collect_r03.sh <tag>
set -e
tag=${1:-r03}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
# 1. headline: bench.py + rocprofv3 kernel stats + the two PMC traffic passes (scripts/collect_profiles.sh)
scripts/collect_profiles.sh $tag > $out/${tag}_collect_profiles.log 2>&1
# 2. the other configurations: HIP-event timings at full size, then rocprofv3 kernel stats of the same script
python3 scripts/bench_configs.py c3 c4 > $out/${tag}_configs_c3_c4.jsonl 2> $out/${tag}_configs_c3_c4.err
python3 scripts/bench_configs.py c5 --c5-check > $out/${tag}_c5_standard_full.json 2> $out/${tag}_c5_standard_full.err
python3 scripts/bench_configs.py c5 --c5-kalman square-root --c5-check > $out/${tag}_c5_sqrt_full.json 2> $out/${tag}_c5_sqrt_full.err
cd /tmp
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${c}_stats -o stats -- python3 $root/scripts/bench_configs.py $c > $out/${tag}_${c}_stats.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c5_stats -o stats -- python3 $root/scripts/bench_configs.py c5 --c5-steps 200 > $out/${tag}_c5_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c5sq_stats -o stats -- python3 $root/scripts/bench_configs.py c5 --c5-steps 200 --c5-kalman square-root > $out/${tag}_c5sq_stats.log 2>&1
# 3. MFMA counters of the dense kernels, both forms, N = 50 (own pass: counters and traces are never combined)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_c5_mfma -o mfma -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50 > $out/${tag}_c5_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_c5sq_mfma -o mfma -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50 --c5-kalman square-root > $out/${tag}_c5sq_mfma.log 2>&1
# 4. n_deriv = 5 on the headline shape: HBM traffic of the blocked-tile path (two PMC passes)
rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_nd5_fetch -o fetch --output-format csv -- python3 $root/scripts/nderiv_times.py 5 > $out/${tag}_nd5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_nd5_write -o write --output-format csv -- python3 $root/scripts/nderiv_times.py 5 > $out/${tag}_nd5_write.log 2>&1
cd $root
python3 scripts/nderiv_times.py > $out/${tag}_nderiv_times.jsonl 2> $out/${tag}_nderiv_times.err
ls $out | grep "^${tag}_" | head -60

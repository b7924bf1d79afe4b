#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel statistics of scripts/bench_configs.py for C3 / C4 / C5, and one PMC
# pass with the MFMA counters on the dense kernels of C5 at N = 50.
# Usage: scripts/collect_config_profiles.sh <tag>   -> gpurun_out/<tag>_{c3,c4,c5}_stats/, gpurun_out/<tag>_c5_mfma/
set -e
tag=${1:-r03}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
cd /tmp
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${c}_stats -o stats -- python3 $root/scripts/bench_configs.py $c > $out/${tag}_${c}_stats.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c5_stats -o stats -- python3 $root/scripts/bench_configs.py c5 --c5-steps 200 > $out/${tag}_c5_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_c5_mfma -o mfma -- python3 $root/scripts/bench_configs.py c5 --c5-steps 50 > $out/${tag}_c5_mfma.log 2>&1
cd $root
find $out/${tag}_c3_stats $out/${tag}_c4_stats $out/${tag}_c5_stats $out/${tag}_c5_mfma -name "*.csv" | head -40

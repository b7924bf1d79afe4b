#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel statistics and the two PMC passes for bench.py's C2 workload.
# Usage: scripts/collect_profiles.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write}/..., gpurun_out/<tag>_bench.json
set -e
tag=${1:-r01}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o stats -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_fetch -o fetch --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_write -o write --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_write.log 2>&1
cd $root
find $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write -name "*.csv" | head -20

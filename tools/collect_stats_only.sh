#!/bin/bash
# GPU box: only the rocprofv3 kernel stats of the default bench command + the default bench line (tools/collect_profiles.sh
# without its PMC passes and other workloads).  usage: bash tools/collect_stats_only.sh <tag>
set -u
TAG=${1:-roundX}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --lanes 1"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- $BENCH > /tmp/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
grep '^{"metric"' /tmp/prof_stats.log | tail -1 > $OUT/${TAG}_bench.json
cd $R && python bench.py 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench_default.json
head -5 $OUT/${TAG}_kernel_stats.csv | cut -c1-160

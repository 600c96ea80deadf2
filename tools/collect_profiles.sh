#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh <tag>'): rocprofv3 kernel stats of the
# default bench.py command, then two separate PMC passes (FETCH_SIZE, WRITE_SIZE -- never combined
# with trace domains other than --kernel-trace) and one SQ pass; summaries land in gpurun_out/<tag>_*.
set -u
TAG=${1:-roundX}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --lanes 1"
$BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- $BENCH > /tmp/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
# four lanes, no chaining (the higher-throughput mode, bench.py's four_lanes leg): the same kernels sharing the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats4 -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --lanes 4 --stagger 0 > /tmp/prof_stats4.log 2>&1
cp $(find /tmp/prof_stats4 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats_4lanes.csv
grep '^{"metric"' /tmp/prof_stats4.log | tail -1 > $OUT/${TAG}_bench_4lanes.json
python $R/bench.py --lanes 4 --stagger 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench_4lanes_unprofiled.json
# the other BASELINE shapes (informational bench lines, CPU baseline + parity leg included)
for W in tumvi hilti mono640 map tumvi512 d455 d435i; do
  python $R/bench.py --workload $W 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench_$W.json
done
python $R/bench.py --workload hilti --split cameras 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench_hilti_split_cameras.json
# counter calibration on K1's own access shapes (4 B/lane reads, 16 B/lane writes): known byte counts
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/cal_$C -o p -- $R/tools/ubench/fetch_calib > /tmp/cal_$C.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/cal_$C -name '*counter_collection.csv' | head -1) $OUT/${TAG}_calib_$C.json > /dev/null
done
export OKVFE_PMC_CALIB=1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -o p -- $BENCH --steps 3 > /tmp/prof_$C.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/prof_$C -name '*counter_collection.csv' | head -1) $OUT/${TAG}_pmc_$C.json > /dev/null
done
# the same two passes on the other image shapes (their own traffic figures instead of EuRoC's)
for W in tumvi mono640 hilti; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_${W}_$C -o p -- $BENCH --steps 3 --workload $W > /tmp/prof_${W}_$C.log 2>&1
    python $R/tools/pmc_summary.py $(find /tmp/prof_${W}_$C -name '*counter_collection.csv' | head -1) $OUT/${TAG}_${W}_pmc_$C.json > /dev/null
  done
done
# (still under OKVFE_PMC_CALIB: bench.py then skips its score-map leg, so every K1 dispatch is the product form)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq -o p -- $BENCH --steps 3 > /tmp/prof_sq.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/prof_sq -name '*counter_collection.csv' | head -1) $OUT/${TAG}_pmc_sq.json > /dev/null
# the same two byte passes with the score map kept (okvfe_set_keep_score_map through the lab knob): the
# 5 P form of the kernel, for the with_score_map block of the bench line
if [ -f $R/okvis2_amd/libokvfe_lab.so ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    OKVFE_LIB=$R/okvis2_amd/libokvfe_lab.so OKVFE_KEEP_SCORE_MAP=1 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_map_$C -o p -- $BENCH --steps 3 > /tmp/prof_map_$C.log 2>&1
    python $R/tools/pmc_summary.py $(find /tmp/prof_map_$C -name '*counter_collection.csv' | head -1) $OUT/${TAG}_withmap_pmc_$C.json > /dev/null
  done
fi
unset OKVFE_PMC_CALIB
ls -la $OUT | tail -12

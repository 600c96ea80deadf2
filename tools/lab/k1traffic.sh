#!/bin/bash
# GPU box: HBM traffic of the fused score kernel (FETCH_SIZE / WRITE_SIZE in separate passes) next to
# the calibration kernels of tools/ubench/fetch_calib (known byte counts, K1's access shapes).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C /tmp/k1_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/cal_$C -o p -- $R/tools/ubench/fetch_calib > /tmp/cal_$C.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/cal_$C -name '*counter_collection.csv' | head -1) $R/gpurun_out/${1:-r2}_calib_$C.json > /dev/null
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/k1_$C -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/k1_$C.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/k1_$C -name '*counter_collection.csv' | head -1) $R/gpurun_out/${1:-r2}_pmc_$C.json > /dev/null
done
python - <<PY
import json
for C in ("FETCH_SIZE","WRITE_SIZE"):
    cal=json.load(open("$R/gpurun_out/${1:-r2}_calib_%s.json"%C)); k=json.load(open("$R/gpurun_out/${1:-r2}_pmc_%s.json"%C))
    for n,v in cal.items(): print(C, n, v["mean_per_dispatch"])
    for n,v in k.items():
        if "harris" in n or "describe_kernel" in n: print(C, n, v["dispatches"], v["mean_per_dispatch"])
PY

#!/bin/bash
# GPU box: SQ / TCC counters of describe_kernel for the workloads given (euroc mono640 ...)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  rm -rf /tmp/prof_d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/prof_d -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --workload $wl > /tmp/prof_d.log 2>&1
  echo "== $wl: $set"
  python $R/tools/pmc_summary.py $(find /tmp/prof_d -name '*counter_collection.csv' | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('describe_kernel'):
        print(k, 'vgpr',v['vgpr'],'lds',v['lds'], {a:round(b) for a,b in v['mean_per_dispatch'].items()})
"
  done
done

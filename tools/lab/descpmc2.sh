#!/bin/bash
# GPU box: where describe_kernel's cycles go (SQ busy / wait / per-unit active counters), EuRoC workload
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/prof_d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/prof_d -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --workload ${WL:-euroc} > /tmp/prof_d.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/prof_d -name '*counter_collection.csv' | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('${KERNEL:-describe_kernel}'):
        print({a:round(b) for a,b in v['mean_per_dispatch'].items()})
" || tail -3 /tmp/prof_d.log
done

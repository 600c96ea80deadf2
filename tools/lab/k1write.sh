#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --exact-steps > /tmp/prof_w.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/prof_w -name '*counter_collection.csv' | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'harris' in k: print(k, v['lds'], v['vgpr'], v['mean_per_dispatch'])
"

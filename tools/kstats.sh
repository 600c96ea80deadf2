#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench command -> gpurun_out/kstats.csv (top rows printed)
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/kstats.csv && python - $R/gpurun_out/kstats.csv <<'PY'
import csv,sys
for x in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-46s calls %4s avg %9.1f us'%(x['Name'].replace('okvfe::(anonymous namespace)::','')[:46], x['Calls'], float(x['AverageNs'])/1e3))
PY
grep '^{"metric"' /tmp/ks.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('fps',round(r['value']),'ms',round(r['ms_per_step'],3))"

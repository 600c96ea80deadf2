#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench command, top kernels only.  usage: bash tools/lab/kstats_once.sh [repeats]
R=$PWD; cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 ${1:-1}); do
  rm -rf /tmp/ks$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$i -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ks$i.log 2>&1
  python - <<EOF2
import csv,glob
f=glob.glob("/tmp/ks$i/**/*kernel_stats.csv", recursive=True)[0]
print(" ".join("%s=%.1f" % (r["Name"].split("::")[-1][:22], float(r["AverageNs"])/1e3) for r in list(csv.DictReader(open(f)))[:4]))
EOF2
done

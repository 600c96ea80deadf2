#!/bin/bash
# GPU box: same-box A/B of K1 variants built with tools/lab/variant.sh (libokvfe_<name>.so); three interleaved repeats
R=$PWD; cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for n in "$@"; do
    rm -rf /tmp/ks_$n
    OKVFE_LIB=$R/okvis2_amd/libokvfe_$n.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ks_$n.log 2>&1
    f=$(find /tmp/ks_$n -name '*kernel_stats.csv' | head -1)
    python - $n $f <<'PY'
import csv,sys
for x in csv.DictReader(open(sys.argv[2])):
    if 'harris_kernel<61, true, false, false, true>' in x['Name']: print(sys.argv[1], 'K1 avg %.1f us'%(float(x['AverageNs'])/1e3)); break
PY
  done
done

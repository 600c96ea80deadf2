#!/bin/bash
# GPU box: same-box A/B of describe_kernel variants (boxes of the pool differ by several per cent, so every
# comparison below 5 % has to run on ONE box, interleaved).  usage: bash tools/lab/descab5.sh name1 name2 ...
R=$PWD
# build the variants first (here or in the build container: the .so files travel with the snapshot):
#   bash tools/lab/variant.sh <name> k_describe.hip "<-D flags>"
names=("$@")
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for n in "${names[@]}"; do
    rm -rf /tmp/ks_$n
    OKVFE_LIB=$R/okvis2_amd/libokvfe_$n.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ks_$n.log 2>&1
    f=$(find /tmp/ks_$n -name '*kernel_stats.csv' | head -1)
    python - $n $f <<'PY'
import csv,sys
for x in csv.DictReader(open(sys.argv[2])):
    if 'describe_kernel' in x['Name']: print(sys.argv[1], 'describe avg %.1f us'%(float(x['AverageNs'])/1e3)); break
PY
  done
done

#!/bin/bash
# GPU box: same-box A/B of variant libraries (tools/lab/variant.sh): per variant the top kernels' average times under
# rocprofv3 and the bench value.  usage: bash tools/lab/libab.sh <kernel substring> libokvfe.so libokvfe_x.so ...   env REPS
R=$PWD; cd /tmp && export TMPDIR=/tmp
K=$1; shift
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    rm -rf /tmp/ks_lib
    OKVFE_LIB=$R/okvis2_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_lib -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > /tmp/ks_lib.log 2>&1
    f=$(find /tmp/ks_lib -name '*kernel_stats.csv' | head -1)
    python - "$lib" "$K" $f /tmp/ks_lib.log <<'PY'
import csv,sys,json
fps=""
for l in open(sys.argv[4]):
    if l.startswith('{"metric"'):
        r=json.loads(l); fps="fps %d"%r["value"]
for x in csv.DictReader(open(sys.argv[3])):
    if sys.argv[2] in x['Name']: print(sys.argv[1], sys.argv[2], 'avg %.1f us'%(float(x['AverageNs'])/1e3), fps); break
PY
  done
done

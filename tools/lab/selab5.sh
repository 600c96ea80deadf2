#!/bin/bash
# GPU box: same-box A/B of select_lazy_kernel under lab-build environment knobs; args: "ENV=val ..." strings ("-" = defaults)
R=$PWD; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e="OKVFE_DUMMY=1"
    rm -rf /tmp/ks_sel
    env OKVFE_LIB=$R/okvis2_amd/libokvfe_lab.so $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_sel -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ks_sel.log 2>&1
    f=$(find /tmp/ks_sel -name '*kernel_stats.csv' | head -1)
    python - "$v" $f <<'PY'
import csv,sys
for x in csv.DictReader(open(sys.argv[2])):
    if 'select_lazy_kernel' in x['Name']: print(sys.argv[1], 'select avg %.1f us'%(float(x['AverageNs'])/1e3)); break
PY
  done
done

#!/bin/bash
# GPU box: one lane vs three lanes for variant libraries ("-" = default)
for n in "$@"; do
  lib=okvis2_amd/libokvfe_$n.so; [ "$n" = "-" ] && lib=okvis2_amd/libokvfe.so
  for l in 1 3; do
    OKVFE_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --lanes $l 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('$n lanes $l fps %.0f ms/step %.3f k1 %.3f frac %.3f'%(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"
  done
done

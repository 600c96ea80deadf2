#!/bin/bash
# GPU box: per-stage times of the default bench step, product library vs lab library (built with whatever -D the A/B
# needs), interleaved.  usage: bash tools/lab/stage_ab.sh [bench args]   env REPS (default 2)
for i in $(seq 1 ${REPS:-2}); do
  for lib in libokvfe.so libokvfe_lab.so; do
    OKVFE_LIB=$PWD/okvis2_amd/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; r=json.loads(sys.stdin.read()); s=r.get('stage_ms_per_launch',{}); print('$lib', round(r['value']), 'ms', round(r['ms_per_step'],3), {k:round(v,3) for k,v in s.items() if v and v>0.02})"
  done
done

#!/bin/bash
# GPU box: lanes inside one call, plain vs priority form (ONE low-priority stream for all score kernels, high-priority
# lane streams for the tails), interleaved.  Lab library.  usage: bash tools/lab/lanes_prio_ab.sh [bench args]
export OKVFE_LIB=$PWD/okvis2_amd/libokvfe_lab.so
run() {
  env "$@" python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('$*', round(r['value']), 'ms', round(r['ms_per_step'],3))"
}
for i in $(seq 1 ${REPS:-2}); do
  run A=0
  run OKVFE_INTERNAL_LANES=4
  run OKVFE_INTERNAL_LANES=2 OKVFE_LANES_PRIO=1
  run OKVFE_INTERNAL_LANES=4 OKVFE_LANES_PRIO=1
  run OKVFE_INTERNAL_LANES=8 OKVFE_LANES_PRIO=1
done

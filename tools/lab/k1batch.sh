#!/bin/bash
# GPU box: K1 launch time vs rows-per-wave (OKVFE_K1_TH) for several batch sizes
for b in "$@"; do
  for th in 61 43 31 25; do
    OKVFE_K1_TH=$th python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --batch $b 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('batch $b TH $th fps %.0f k1_ms %.4f frac %.3f'%(r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"
  done
done

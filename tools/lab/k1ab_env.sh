#!/bin/bash
# GPU box: interleaved A/B of K1 variants.  Each argument = name[:ENV=VAL[,ENV=VAL...]]; name as for
# k1variants.sh ("-" = default library).  ROUNDS (default 3) interleaved passes beat box drift.
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
for a in "$@"; do
  n=${a%%:*}; envs=""; [ "$a" != "$n" ] && envs=$(echo "${a#*:}" | tr ',' ' ')
  lib=okvis2_amd/libokvfe_$n.so; [ "$n" = "-" ] && lib=okvis2_amd/libokvfe.so
  env $envs OKVFE_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
  python - "$a" <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_tmp.json")); print(sys.argv[1], "fps %.0f ms/step %.3f frac %.3f k1_ms %.4f"%(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
PY
done
done

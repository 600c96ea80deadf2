#!/bin/bash
# GPU box: K1 and its byte mover (bench.py roofline block) for library variants (names as for k1ab_env.sh)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
for a in "$@"; do
  n=${a%%:*}; envs=""; [ "$a" != "$n" ] && envs=$(echo "${a#*:}" | tr ',' ' ')
  lib=okvis2_amd/libokvfe_$n.so; [ "$n" = "-" ] && lib=okvis2_amd/libokvfe.so
  env $envs OKVFE_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
  python - "$a" <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_tmp.json")); f=r["roofline"]; print(sys.argv[1], "fps %.0f k1 %.4f iso %.4f mover %.4f frac %.3f"%(r["value"], f["avg_launch_ms"], f["isolated_launch_ms"], f["byte_mover_ms"] or 0, f["frac"]))
PY
done
done

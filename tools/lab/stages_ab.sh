#!/bin/bash
# GPU box: per-stage times of library variants (arguments as for k1ab_env.sh)
for a in "$@"; do
  n=${a%%:*}; envs=""; [ "$a" != "$n" ] && envs=$(echo "${a#*:}" | tr ',' ' ')
  lib=okvis2_amd/libokvfe_$n.so; [ "$n" = "-" ] && lib=okvis2_amd/libokvfe.so
  env $envs OKVFE_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
  python - "$a" <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_tmp.json")); print(sys.argv[1], "fps %.0f ms/step %.3f"%(r["value"], r["ms_per_step"]), {k: round(v,4) for k,v in r["stage_ms_per_launch"].items() if v is not None})
PY
done

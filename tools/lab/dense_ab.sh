#!/bin/bash
# GPU box: default and dense-content (checker) step of library variants (names as for k1ab_env.sh)
for a in "$@"; do
  lib=okvis2_amd/libokvfe_$a.so; [ "$a" = "-" ] && lib=okvis2_amd/libokvfe.so
  for c in corners checker; do
    OKVFE_LIB=$PWD/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --content $c > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
    python - "$a" $c <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_tmp.json")); s=r["stage_ms_per_launch"]; print(sys.argv[1], sys.argv[2], "fps %.0f ms/step %.3f match %.4f"%(r["value"], r["ms_per_step"], s["match"]))
PY
  done
done

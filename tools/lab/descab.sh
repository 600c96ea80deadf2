#!/bin/bash
# GPU box: describe-stage time for library variants x workloads
for wl in euroc mono640; do
for n in "$@"; do
  lib=okvis2_amd/libokvfe_$n.so; [ "$n" = "-" ] && lib=okvis2_amd/libokvfe.so
  OKVFE_LIB=$PWD/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --workload $wl > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
  python - "$n" $wl <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_tmp.json")); s=r["stage_ms_per_launch"]; print(sys.argv[2], sys.argv[1], "ms/step %.3f describe %.3f select %.3f kp/img %s"%(r["ms_per_step"], s["describe"], s["select"], r.get("mean_keypoints_per_image")))
PY
done
done

#!/bin/bash
# GPU box: describe-stage time of the default library under environment variants
# (arguments: ENV=VAL[,ENV=VAL] or "-"; WLS = workloads, default "euroc mono640")
for wl in ${WLS:-euroc mono640}; do
for a in "$@"; do
  envs=""; [ "$a" != "-" ] && envs=$(echo "$a" | tr ',' ' ')
  env $envs python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --workload $wl > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
  python - "$a" $wl <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_tmp.json")); s=r["stage_ms_per_launch"]; print(sys.argv[2], sys.argv[1], "ms/step %.3f describe %.3f fps %.0f"%(r["ms_per_step"], s["describe"], r["value"]))
PY
done
done

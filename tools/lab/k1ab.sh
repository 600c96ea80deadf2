#!/bin/bash
# GPU box: bench line summary for each variant library given (names as for k1variants.sh; "-" = default)
for n in "$@"; do
  lib=okvis2_amd/libokvfe_$n.so; [ "$n" = "-" ] && lib=okvis2_amd/libokvfe.so
  OKVFE_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$n.json 2>gpurun_out/ab_$n.err
  python - "$n" <<'PY'
import json,sys; r=json.load(open("gpurun_out/ab_%s.json"%sys.argv[1])); print(sys.argv[1], "fps %.0f ms/step %.3f frac %.3f k1_ms %.4f copyGBps %.0f"%(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["roofline"]["copy_kernel_GBps"]))
PY
done

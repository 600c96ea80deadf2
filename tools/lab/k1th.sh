#!/bin/bash
# GPU box: bench summary of the default library for several OKVFE_K1_TH settings
for th in "$@"; do
  OKVFE_LIB=$PWD/okvis2_amd/libokvfe_lab.so OKVFE_K1_TH=$th python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/th_$th.json 2>gpurun_out/th_$th.err
  python - "$th" <<'PY'
import json,sys; r=json.load(open("gpurun_out/th_%s.json"%sys.argv[1])); print("TH", sys.argv[1], "fps %.0f ms/step %.3f frac %.3f k1_ms %.4f"%(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
PY
done

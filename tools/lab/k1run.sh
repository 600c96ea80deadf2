python -m pytest tests/test_gpu_parity.py tests/test_gpu_detector_paths.py tests/test_gpu_golden_and_gather.py -x -q 2>&1 | tail -8
for v in "" "OKVFE_K1_TH=31"; do
env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2_b.json 2>gpurun_out/r2_b.err
python - <<'PY'
import json; r=json.load(open("gpurun_out/r2_b.json")); print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["stage_ms_per_launch"])
PY
done

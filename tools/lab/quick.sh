#!/bin/bash
# GPU box: parity suite + default bench line summary (+ optional select stats)
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python bench.py --steps 20 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/b.json 2>gpurun_out/b.err
python -c "
import json; r=json.load(open('gpurun_out/b.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac']); print(r['stage_ms_per_launch']); print(r['dense_content'])"

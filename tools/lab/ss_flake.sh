#!/bin/bash
# GPU box, lab library: how often does the 6-layer scale-space test fail under the bisecting knobs of detect_layers_concurrent?
export OKVFE_LIB=$PWD/okvis2_amd/libokvfe_lab.so
run() { f=0; for i in $(seq 1 ${N:-8}); do env "$@" timeout 120 python -m pytest "tests/test_gpu_octaves.py::test_scale_space_sizes_and_batches" -x -q 2>&1 | grep -q "failed" && f=$((f+1)); done; echo "$* : $f of ${N:-8} runs failed"; }
for v in "$@"; do run $v; done

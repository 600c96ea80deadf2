run() { make -C okvis2_amd/csrc -j8 2>&1 | grep -E "error" -A5; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --lanes 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['stage_ms_per_launch']['describe'])"; }
run "waves5 buf7680"
sed -i 's/amdgpu_waves_per_eu(5, 8)/amdgpu_waves_per_eu(6, 8)/; s/constexpr int kPatchBufBytes = 7680;/constexpr int kPatchBufBytes = 6144;/' okvis2_amd/csrc/k_describe.hip
run "waves6 buf6144"
sed -i 's/amdgpu_waves_per_eu(6, 8)/amdgpu_waves_per_eu(8, 8)/; s/constexpr int kPatchBufBytes = 6144;/constexpr int kPatchBufBytes = 4864;/' okvis2_amd/csrc/k_describe.hip
run "waves8 buf4864"

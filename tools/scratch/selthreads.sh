for t in 256 512 1024; do
  sed -i "s/^constexpr int kSelThreads = [0-9]*;/constexpr int kSelThreads = $t;/" okvis2_amd/csrc/k_select.hip
  make -C okvis2_amd/csrc -j8 2>&1 | grep -E "error" -A5
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --lanes 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads',$t, d['value'], d['stage_ms_per_launch']['select'])"
done

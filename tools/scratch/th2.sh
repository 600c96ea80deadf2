export OKVFE_NO_FUSED_NMS=1
for t in 30 60 120; do OKVFE_K1_TH=$t python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused TH',$t, d['stage_ms_per_launch']['harris'])"; done

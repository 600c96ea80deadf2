python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; grep -n "passed\|failed" gpurun_out/gputest.log | tail -2
for W in euroc tumvi hilti mono640 tumvi512 d455 d435i; do
  python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; r=json.loads(sys.stdin.read()); s=r.get('stage_ms_per_launch',{}); print('$W', round(r['value']), 'ms', round(r['ms_per_step'],3), {k:round(v,3) for k,v in s.items() if v})"
done
python bench.py --box-widen 1.73 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; r=json.loads(sys.stdin.read()); s=r.get('stage_ms_per_launch',{}); print('widen1.73', round(r['value']), 'ms', round(r['ms_per_step'],3), {k:round(v,3) for k,v in s.items() if v})"

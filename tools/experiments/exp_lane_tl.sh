timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "goldens" 2>&1 | tail -1
for v in 2 0 2 0; do python bench.py --no-cpu-baseline --kernel-variant $v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant', sys.argv[1], d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])" $v; done
python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -3

python tools/probe.py --variants 1,2 --check --frames 3 2>&1 | grep -E "frame |atrous"
python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'], d['kernels_us'])"
SVGF_STRIP_DBG=40 python tools/probe.py --variants 2 --frames 2 2>&1 | grep -E "strip dbg|wave|loader" | head -40 | awk 'NR<6 || (NR>17 && NR<22) || NR>33'
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2

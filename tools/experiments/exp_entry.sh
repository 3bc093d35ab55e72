timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_paper_steps.py -x -q 2>&1 | tail -1
line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], d['kernels_us'], 'iso', r['isolated']['mean_launch_us'])" "${CFG:-1080p-static}" "$@"; }
line; line; line --no-overlap; line --no-overlap; CFG=1080p-moving line; CFG=4k-static line; CFG=4k-moving line; CFG=config1 line
python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -5
python tools/probe.py --variants 2 --frames 6 2>&1 | grep -E "atrous" | head -5

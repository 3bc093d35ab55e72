# lane kernel: next iteration's centre / first tap row read before the barrier
line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])" "${CFG:-1080p-static}"; }
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -1
line; line; CFG=1080p-moving line; CFG=4k-static line
python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -5

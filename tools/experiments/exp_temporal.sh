timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_binary_gpu.py tests/test_reproj_scale.py -x -q 2>&1 | tail -2
line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], d['kernels_us'], 'iso', r['isolated']['mean_launch_us'])" "${CFG:-1080p-static}"; }
line; line; CFG=1080p-moving line; CFG=4k-static line; CFG=4k-moving line
python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "temporal|frame wall" | head -4
python tools/probe.py --size 3840x2160 --variants 0 --frames 6 2>&1 | grep -E "temporal|frame wall" | head -4

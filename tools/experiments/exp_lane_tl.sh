timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "goldens" 2>&1 | tail -1
for v in 2 0 2 0; do python bench.py --no-cpu-baseline --kernel-variant $v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant', sys.argv[1], d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])" $v; done
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
SVGF_LANE_DBG_SKIP=12 SVGF_LANE_DBG=40 python tools/probe.py --variants 0 --frames 8 2>&1 | grep -E "lane dbg|wave  0 it  [2-5]|wave  7 it  [2-5]" | head -9

timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_paper_steps.py -x -q 2>&1 | tail -1
line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], d['kernels_us'], 'iso', r['isolated']['mean_launch_us'])" "${CFG:-1080p-static}" "$@"; }
line; line; line --no-overlap; CFG=1080p-moving line; CFG=4k-static line
python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -5
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -1
SVGF_LANE_DBG=40 SVGF_LANE_DBG_SKIP=6 python tools/probe.py --variants 0 --frames 4 2>&1 | grep -E "lane dbg|prologue" | head -16

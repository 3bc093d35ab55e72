line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])" "NWC=${SVGF_LANE_NWC:-8}" "${CFG:-1080p-static}"; }
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_WPE=4" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
SVGF_LANE_NWC=4 timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k goldens 2>&1 | tail -1
for rep in 1 2; do
  SVGF_LANE_NWC=8 line; SVGF_LANE_NWC=4 line
done
SVGF_LANE_NWC=4 python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -5

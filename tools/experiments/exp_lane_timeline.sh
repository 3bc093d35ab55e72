# in-kernel timeline of the shipped lane kernel (stamps of waves 0, 7 = stage orders A, B; 8, 11 = loaders), iterations 0..7
# usage: exp_lane_timeline.sh [WxH]
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
for b in 40 200; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=5 python tools/probe.py --size ${1:-1920x1080} --variants 0 --frames 3 2>&1 | grep -E "lane dbg|prologue|it  [0-7]:" | head -80; done

SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
SVGF_LANE_DBG=40 SVGF_LANE_DBG_SKIP=6 python tools/probe.py --variants 0 --frames 4 2>&1 | grep -E "lane dbg|prologue|it  0|it  1:|it  2:|it 15|it 16" | head -40

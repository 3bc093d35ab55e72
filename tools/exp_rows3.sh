SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_STRIP_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 1080p warm"; SVGF_STRIP_DBG_SKIP=20 SVGF_STRIP_DBG=40 python tools/probe.py --variants 2 --frames 8 2>&1 | grep -E "strip dbg|wave  0 it  [0-3]|wave  7 it  [0-3]|loader  it  [0-2]" | head -60

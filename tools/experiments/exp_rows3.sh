SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LOADER_PRIO=3 -DSVGF_STRIP_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== ROWS=3 loader prio 3"; SVGF_STRIP_ROWS=3 python tools/probe.py --variants 2 --frames 6 2>&1 | grep -E "atrous|frame wall"
SVGF_STRIP_ROWS=3 SVGF_STRIP_DBG_SKIP=10 SVGF_STRIP_DBG=40 python tools/probe.py --variants 2 --frames 4 2>&1 | grep -E "strip dbg|wave  0 it  [1-3]|wave 11 it  [1-3]|loader  it  [1-3]" | head -10
echo "== ROWS=2 loader prio 3"; python tools/probe.py --variants 2 --frames 6 2>&1 | grep -E "atrous|frame wall"

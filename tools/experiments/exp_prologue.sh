python tools/probe.py --variants 1,2 --check --frames 4 2>&1 | grep -E "frame |atrous" | tail -9
python tools/probe.py --size 800x800 --nlevel 1 --variants 2 --frames 6 2>&1 | grep -E "atrous"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --config config1 2>/dev/null | tail -1 | cut -c1-200
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_STRIP_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
SVGF_STRIP_DBG_SKIP=20 SVGF_STRIP_DBG=40 python tools/probe.py --variants 2 --frames 8 2>&1 | grep -E "strip dbg|prologue" | head -12

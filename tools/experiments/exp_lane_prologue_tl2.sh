# in-kernel prologue timeline of the lane kernel: two-group prologue (mode 1) against geometry-first (mode 2)
for M in 1 2; do
echo "== SVGF_LANE_SPLIT_PROLOGUE=$M"
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE -DSVGF_LANE_SPLIT_PROLOGUE=$M" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
for b in 40 7 200; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=6 python tools/probe.py --variants 0 --frames 4 2>&1 | grep -E "lane dbg|prologue|it  [01]:" | head -14; done
done

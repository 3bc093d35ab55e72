# round 5: in-kernel timeline of the lane kernel with the half-iteration skew (SVGF_LANE_SKEW=1) and without
mkdir -p gpurun_out/r05
for sk in 1 0; do
echo "=== SVGF_LANE_SKEW=$sk"
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE -DSVGF_LANE_SKEW=$sk $EXTRA" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
for b in 40; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=5 python tools/probe.py --size ${1:-1920x1080} --variants 0 --frames 3 2>&1 | grep -E "lane dbg|prologue|it  [0-7]:" | head -60; done
done

# Does a THIRD computing wave per SIMD cost the two compute waves anything?  The lane kernel's loader waves (one per SIMD,
# mostly parked) get N x 16 private VALU instructions per iteration (-DSVGF_LANE_LOADER_BUSY=N; 19 ~ one tap row's mix).
for N in 0 10 19 29; do
  if [ "$N" = 0 ]; then export SVGF_EXTRA_HIPCC_FLAGS=""; else export SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_LOADER_BUSY=$N"; fi
  echo "== loader busy N=$N ($((N*16)) VALU per loader wave per iteration)"
  python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
  python tools/probe.py --variants 4 --frames 8 2>&1 | grep -E "atrous|frame wall" | head -6
done

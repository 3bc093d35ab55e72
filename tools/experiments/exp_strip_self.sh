# strip kernel without loader waves (SELF: 12 compute waves that stage their own rows) against the shipped shapes
python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "tuning_configurations" 2>&1 | tail -3
echo "== strip, shipped shape (ROWS=2 + loader waves)"
python tools/probe.py --variants 2 --frames 8 2>&1 | grep -E "atrous|frame wall" | head -6
echo "== strip ROWS=3 + loader waves"
SVGF_STRIP_ROWS=3 python tools/probe.py --variants 2 --frames 8 2>&1 | grep -E "atrous|frame wall" | head -6
echo "== strip ROWS=3 SELF (12 waves, all computing)"
SVGF_STRIP_ROWS=3 SVGF_STRIP_SELF=1 python tools/probe.py --variants 2 --frames 8 2>&1 | grep -E "atrous|frame wall" | head -6
echo "== 4K strip ROWS=3 SELF"
SVGF_STRIP_ROWS=3 SVGF_STRIP_SELF=1 python tools/probe.py --variants 2 --frames 6 --size 3840x2160 2>&1 | grep -E "atrous|frame wall" | head -6
echo "== 4K default"
python tools/probe.py --variants 0 --frames 6 --size 3840x2160 2>&1 | grep -E "atrous|frame wall" | head -6

for F in "-DSVGF_LANE_DEEP=3" "-DSVGF_LANE_DEEP=2" ""; do
SVGF_EXTRA_HIPCC_FLAGS="$F" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== flags: $F"
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "goldens or sequences or overlap" 2>&1 | tail -1
line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], d['kernels_us'], 'iso', r['isolated']['mean_launch_us'])" "${CFG:-1080p-static}"; }
line; line; CFG=4k-static line
python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -3
done

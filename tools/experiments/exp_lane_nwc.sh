# lane kernel: one 8-wave workgroup per CU (480 columns) vs two 4-wave workgroups per CU (240 columns) for S = 2, 4
line() { python bench.py --no-cpu-baseline --config ${CFG:-1080p-static} "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1:], d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])" "NWC=${SVGF_LANE_NWC:-8}" "${CFG:-1080p-static}"; }
SVGF_LANE_NWC=4 timeout 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -1
for rep in 1 2; do
  SVGF_LANE_NWC=8 line; SVGF_LANE_NWC=4 line
done
SVGF_LANE_NWC=8 python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -5
SVGF_LANE_NWC=4 python tools/probe.py --variants 0 --frames 6 2>&1 | grep -E "atrous" | head -5
CFG=4k-static SVGF_LANE_NWC=8 line; CFG=4k-static SVGF_LANE_NWC=4 line

run() { f="$1"
  SVGF_EXTRA_HIPCC_FLAGS="$f" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 120 python -m pytest tests/test_parity_gpu.py -x -q -k "goldens" 2>&1 | tail -1
  for i in 1 2; do timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])"; done
}
run "-DSVGF_LANE_ASYNC -DSVGF_LANE_ASYNC_NAP=12"
run "-DSVGF_LANE_ASYNC -DSVGF_LANE_ASYNC_NAP=40"
run "-DSVGF_LANE_ASYNC -DSVGF_LANE_ASYNC_NAP=4"
run ""

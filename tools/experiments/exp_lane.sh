run() { f="$1"; shift
  SVGF_EXTRA_HIPCC_FLAGS="$f" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
  for v in "$@"; do python bench.py --no-cpu-baseline --kernel-variant $v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f', 'variant', sys.argv[1], d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'])" $v; done
}
run "-DSVGF_LANE_G2=0" 2 4
run "-DSVGF_LANE_G2=2" 4 2 4
run "-DSVGF_LANE_G2=0" 4

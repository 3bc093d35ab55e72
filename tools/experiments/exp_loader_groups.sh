run() { f="$1"; shift
  SVGF_EXTRA_HIPCC_FLAGS="$f" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== $f | $@"
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'timed', r['mean_launch_us'], 'iso', r['isolated']['mean_launch_us'], d['kernels_us'])"
}
python tools/probe.py --variants 1,2 --check --frames 3 2>&1 | grep -E "frame |atrous" | tail -8
run ""
run "-DSVGF_STRIP_TIMELINE"
SVGF_STRIP_DBG=40 python tools/probe.py --variants 2 --frames 2 2>&1 | grep -E "strip dbg|wave|loader" | head -40 | awk 'NR<6 || (NR>17 && NR<22) || (NR>33 && NR < 38)'

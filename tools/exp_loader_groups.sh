mkdir -p gpurun_out
run() { # $1 = flags, $2.. = bench args
  f="$1"; shift
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
run "-DSVGF_LOADER_GROUPS=3" --config 4k-static
run "-DSVGF_LOADER_GROUPS=2" --config 4k-static
run "-DSVGF_LOADER_GROUPS=2" --config 1080p-moving
run "-DSVGF_LOADER_GROUPS=3" --config 1080p-moving
run "-DSVGF_LOADER_GROUPS=2"

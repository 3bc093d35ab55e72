# alternate builds with different SVGF_EXTRA_HIPCC_FLAGS on one box; two rounds
for round in 1 2; do for F in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -enable-post-misched=0" "-mllvm -amdgpu-sched-strategy=max-memory-clause"; do
SVGF_EXTRA_HIPCC_FLAGS="$F" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(repr(sys.argv[1]), d['value'], d['ms_per_step'], d['kernels_us'], 'iso', r['isolated']['mean_launch_us'])" "$F"; done
python bench.py --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(repr(sys.argv[1]), 'no-overlap', d['value'], d['kernels_us'])" "$F"
done; done

# A/B of two versions of one source file: $1 = repo-relative path, $2 / $3 = the two candidate files
T=$1
for round in 1 2; do for V in $2 $3; do
cp $V $T
python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], d['kernels_us'], 'iso', r['isolated']['mean_launch_us'])" $V; done
python bench.py --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(sys.argv[1], 'no-overlap', d['value'], d['kernels_us'])" $V
done; done

# config1 (800x800, one level, temporal off): lane kernel with forced segment lengths, and the strip kernel.
R=$GRAFT_REPO_ROOT; cd $R
run() { env $1 python bench.py --config config1 --steps 200 --warmup 20 --no-cpu-baseline --kernel-variant $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 variant $2: frame %.2f us, level %.2f us (isolated %.2f), frac %.4f' % (d['ms_per_step']*1e3, r['mean_launch_us'], r['isolated']['mean_launch_us'], r['frac']))"; }
for rep in 1 2; do
run X=0 0
for L in 4 5 6 7 8 10 13 25; do run SVGF_LANE_SEGROWS=$L 0; done
run X=0 2
for L in 4 6 8 13; do run SVGF_STRIP_SEGROWS=$L 2; done
done

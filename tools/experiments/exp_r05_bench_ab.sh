# round 5: A/B of prebuilt library variants (libsvgf_hip.so.<tag>) on the WHOLE-FRAME figure of bench.py with the frame pipeline (the
# chip at its power limit): usage: exp_r05_bench_ab.sh "A C" [rounds]
R=$GRAFT_REPO_ROOT
L=$R/cuda-path-tracer-denoising_amd/libsvgf_hip.so
E=$R/cuda-path-tracer-denoising_amd/libsvgf_hip_exp.so      # the variants are experiments builds: loaded as such (SVGF_USE_EXPERIMENTS_LIB)
export SVGF_USE_EXPERIMENTS_LIB=1
cp $E $E.orig
for i in $(seq 1 ${2:-3}); do for v in $1; do
  cp $L.$v $E; touch $E
  python $R/bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['telemetry']['timed_region']
print('$v', 'pipelined', d['ms_per_step'], 'ordered', (d.get('ordered') or {}).get('ms_per_step'), 'level alone', d['roofline']['isolated']['mean_launch_us'], 'power', (t.get('power_w') or {}).get('median'), 'sclk', (t.get('sclk_mhz') or {}).get('median'))"
done; done
cp $E.orig $E

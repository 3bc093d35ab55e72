set -x
mkdir -p gpurun_out/r06_base
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_base/bench_default_$i.json 2> gpurun_out/r06_base/bench_default_$i.err; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > gpurun_out/r06_base/bench_ordered_$i.json 2> gpurun_out/r06_base/bench_ordered_$i.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_base/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('ordered',{}) and d['ordered']['ms_per_step'], d['frame_pipeline'], d['frame_pipeline_trial'], d['latency_ms_sync'], d['roofline']['isolated'], d['telemetry']['timed_region'])
    except Exception as e: print(f, 'ERR', e)
PY

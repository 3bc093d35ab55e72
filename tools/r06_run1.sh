set -x
mkdir -p gpurun_out/r06_run1
python -m pytest tests/test_pipeline_gpu.py tests/test_farm_gloo.py tests/test_stream_gpu.py "tests/test_parity_gpu.py::test_back_to_back_asynchronous_frames_equal_synchronised_frames" -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r06_run1/tests.txt
cat gpurun_out/r06_run1/tests.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_run1/bench_$i.json 2> gpurun_out/r06_run1/bench_$i.err; tail -3 gpurun_out/r06_run1/bench_$i.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], 'pipe', d['frame_pipeline'], d['frame_pipeline_trial'], 'other', (d.get('ordered') or d.get('pipelined')), 'lat', d['latency_ms_sync'], 'roof', d['roofline']['frac'], d['roofline']['mean_launch_us'], d['roofline']['timed_region'], 'first', d['first_frame_ms'], d['context_create_ms'], 'cad', d['cadence'])
    except Exception as e: print(f, 'ERR', e)
PY

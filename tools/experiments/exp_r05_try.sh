# round 5: quick GPU pass over what changed (new tests, bench fields, the 4k-room config)
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_farm_gloo.py "tests/test_ref_scenes.py::test_room_4k_static_device_producer_to_denoiser_vs_oracle" "tests/test_parity_gpu.py::test_eight_host_threads_eight_contexts" "tests/test_parity_gpu.py::test_error_codes" -m gpu -x -q -s 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_try.json 2> gpurun_out/r05/bench_try.err; tail -c 2000 gpurun_out/r05/bench_try.err
python bench.py --no-cpu-baseline --config 4k-room --steps 10 > gpurun_out/r05/bench_try_room.json 2>> gpurun_out/r05/bench_try.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_try.json", "gpurun_out/r05/bench_try_room.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line:", e); continue
    cb = d.get("cpu_baseline", {})
    print(f, d["value"], d["ms_per_step"], d["state"], d["cold_ms_per_step"], d["per_rank"], d["roofline"]["frac"], d["roofline"]["traffic"],
          cb.get("value"), cb.get("min"), cb.get("max"), cb.get("one_thread"))
PY

# round 5: the frame pipeline — its tests, the bench line with and without it, rocprofv3 of both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_stream_gpu.py tests/test_farm_gloo.py "tests/test_parity_gpu.py::test_back_to_back_asynchronous_frames_equal_synchronised_frames" tests/test_profile_gpu.py -m gpu -x -q -rA 2>&1 | grep -E "passed|failed|Error|error|worst|assert" | tail -15
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_pipeline.json 2> gpurun_out/r05/bench_pipeline.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > gpurun_out/r05/bench_no_pipeline.json 2>> gpurun_out/r05/bench_pipeline.err
python bench.py --no-cpu-baseline --config 4k-static > gpurun_out/r05/bench_pipeline_4k.json 2>> gpurun_out/r05/bench_pipeline.err
python bench.py --no-cpu-baseline --config 1080p-moving > gpurun_out/r05/bench_pipeline_moving.json 2>> gpurun_out/r05/bench_pipeline.err
python bench.py --no-cpu-baseline --config config1 > gpurun_out/r05/bench_pipeline_config1.json 2>> gpurun_out/r05/bench_pipeline.err
tail -c 1500 gpurun_out/r05/bench_pipeline.err
python - <<'PY'
import json
for f in ("bench_pipeline", "bench_no_pipeline", "bench_pipeline_4k", "bench_pipeline_moving", "bench_pipeline_config1"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05/{f}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], "ordered", d.get("ordered"), "level", r["mean_launch_us"], "frac", r["frac"], "iso", r["isolated"]["mean_launch_us"], r["isolated"]["frac"],
          "in flight", r.get("kernels_in_flight_mean"), "lat", d["latency_ms_sync"], "temporal", d["kernels_us"]["temporal"])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05/rocprofv3_kernel_stats_pipeline.csv; rm -rf /tmp/pp
cut -d, -f1-4 $R/gpurun_out/r05/rocprofv3_kernel_stats_pipeline.csv | head -8

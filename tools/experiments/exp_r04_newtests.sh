#!/bin/bash
# round 4: the tests added after the last full-suite run + the bench line with latency and telemetry
O=gpurun_out/r04_newtests; mkdir -p $O
timeout 1500 python -m pytest tests/test_ref_scenes.py tests/test_planar_inputs.py tests/test_farm_gloo.py tests/test_reuse_gpu.py -m gpu -q -s 2>&1 | tail -15 > $O/tests.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err
python bench.py --no-cpu-baseline > $O/bench_200.json 2>> $O/bench.err

#!/bin/bash
# round 4: the prepare pass of the non-temporal mode fused into the first level (FUSED = 3): parity, config1 bench A/B, 1080p non-temporal
O=$GRAFT_REPO_ROOT/gpurun_out/r04_prep; mkdir -p $O; rm -f $O/*
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_prepare_fused_gpu.py -x -q 2>&1 | tail -8 > $O/tests.txt
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ref_scenes.py tests/test_vs_reference_binary_gpu.py -m gpu -x -q 2>&1 | tail -4 > $O/tests_parity.txt
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l["roofline"]; print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), "frac", r["frac"], "sync", l["latency_ms_sync"])'
for r in 1 2 3; do
  for v in 0 4; do timeout 200 python bench.py --config config1 --steps 200 --warmup 5 --no-cpu-baseline --kernel-variant $v 2>/dev/null | python -c "$pick" config1-variant$v >> $O/bench.log 2>&1; done
done

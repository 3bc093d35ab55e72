#!/bin/bash
# round 4: where did 0.266 -> 0.309 ms (driver command) come from?  (a) telemetry sampler on / off / slower, (b) round-3 library (A) vs tree (B)
O=gpurun_out/r04_regress; mkdir -p $O
for r in 1 2 3; do
  SVGF_BENCH_NO_TELEMETRY=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-telemetry ', l['value'], l['ms_per_step'], l['kernels_us'], l['latency_ms_sync'])" >> $O/bench_ab.log
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('telemetry 5ms ', l['value'], l['ms_per_step'], l['kernels_us'], l['latency_ms_sync'])" >> $O/bench_ab.log
  SVGF_BENCH_TELEMETRY_PERIOD=0.05 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('telemetry 50ms', l['value'], l['ms_per_step'], l['kernels_us'], l['latency_ms_sync'])" >> $O/bench_ab.log
done
bash tools/experiments/exp_ab_multi.sh "A B" 4 3 > $O/ab_r03_vs_tree.log 2>&1

#!/bin/bash
# round 4: bench.py with the telemetry sampler created before the warm-up (hypothesis: the idle gap in front of the timed region)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_bisect4; mkdir -p $O
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), "sync", l.get("latency_ms_sync"), "idle", l.get("idle_before_timed_region_ms"))'
for r in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/_r03 && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r03) >> $O/bench_ab.log 2>&1
  (cd $GRAFT_REPO_ROOT && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r04) >> $O/bench_ab.log 2>&1
done
(cd $GRAFT_REPO_ROOT && timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r04-200) >> $O/bench_ab.log 2>&1

#!/bin/bash
# round 4: chained profiling events + profiler re-arm without event re-creation: parity of the profiling tests, bench lines
O=$GRAFT_REPO_ROOT/gpurun_out/r04_events; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "profil or farm or bench or abi or determinism" 2>&1 | tail -4 > $O/tests.txt
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), "sync", l.get("latency_ms_sync"), "idle", l.get("idle_before_timed_region_ms"), "frac", l["roofline"]["frac"])'
for r in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r04 >> $O/bench.log 2>&1
done
timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r04-200 >> $O/bench.log 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --planar-inputs 2>/dev/null | python -c "$pick" r04-planar >> $O/bench.log 2>&1
timeout 300 python tools/clock_states.py --frames 20 2>/dev/null | head -6 > $O/clock_head.txt

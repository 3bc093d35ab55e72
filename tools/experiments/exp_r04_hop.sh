#!/bin/bash
# round 4: per-kernel timing through events attached to the dispatch (hipExtLaunchKernelGGL) vs rocprofv3's kernel durations, same command
O=$GRAFT_REPO_ROOT/gpurun_out/r04_hop; mkdir -p $O; rm -f $O/*
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_farm_gloo.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l["roofline"]; print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), "frac", r["frac"], "isolated", r["isolated"]["mean_launch_us"], "sync", l["latency_ms_sync"])'
for r in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" run$r >> $O/bench.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" under-rocprofv3 >> $O/bench.log 2>&1
python - $(find $O/prof -name '*kernel_stats.csv' | head -1) >> $O/bench.log <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "atrous" in r["Name"] or "temporal" in r["Name"]]
lv = [float(r["AverageNs"]) / 1e3 for r in rows if "atrous" in r["Name"]]
print("rocprofv3:", " ".join(f"{float(r['AverageNs'])/1e3:.2f}" for r in rows), "| mean level", round(sum(lv) / len(lv), 2))
PY
rm -rf $O/prof
cd $GRAFT_REPO_ROOT && timeout 300 python tools/probe.py --variants 0 --sustain 0.6 --reps 200 > $O/probe.log 2>&1
timeout 300 python tools/clock_states.py 2>/dev/null | head -5 > $O/clock_head.txt

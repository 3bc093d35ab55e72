#!/bin/bash
# round 4: where did the driver-command frame time go from 0.266 to 0.313 ms?  Whole trees (own bench.py + own library) of
# e66fa1e (_r03), 262e3d2 (_b1), 5747680 (_b2) and this tree, alternating on one box.
O=$GRAFT_REPO_ROOT/gpurun_out/r04_bisect; mkdir -p $O
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l["roofline"]; print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), "isolated", r.get("isolated",{}).get("mean_launch_us"), "sync", l.get("latency_ms_sync"))'
for r in 1 2; do
  for t in _r03 _b1 _b2 .; do
    (cd $GRAFT_REPO_ROOT/$t && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/err.log | python -c "$pick" $t) >> $O/bench_ab.log 2>&1
  done
done

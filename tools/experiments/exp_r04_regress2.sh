#!/bin/bash
# round 4: whole-bench A/B, round-3 tree (_r03/, its own bench.py + library) vs this tree, alternating on one box
O=$GRAFT_REPO_ROOT/gpurun_out/r04_regress2; mkdir -p $O
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), l.get("latency_ms_sync"), l.get("warmup_steps_run"))'
for r in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/_r03 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r03) >> $O/bench_ab.log
  (cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r04) >> $O/bench_ab.log
done
(cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) > $O/r04_line.json
(cd $GRAFT_REPO_ROOT/_r03 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) > $O/r03_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_r04 -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
(cd $GRAFT_REPO_ROOT/_r03 && rocprofv3 --kernel-trace --stats -d $O/prof_r03 -o p --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
for t in r03 r04; do f=$(find $O/prof_$t -name '*kernel_stats.csv' | head -1); echo "== $t"; head -12 $f | cut -d, -f1-8 | cut -c1-160; done > $O/kernel_stats.txt
rm -rf $O/prof_r03 $O/prof_r04

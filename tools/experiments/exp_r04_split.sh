#!/bin/bash
# round 4: the G-buffer split of temporal frames in the first level's loaders (FUSED = 4, SVGF_SPLIT_FUSED=1) against the default
O=$GRAFT_REPO_ROOT/gpurun_out/r04_split; mkdir -p $O; rm -f $O/*
cd $GRAFT_REPO_ROOT
SVGF_SPLIT_FUSED=1 timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ref_scenes.py tests/test_f4_options.py tests/test_planar_inputs.py -m gpu -x -q 2>&1 | tail -4 > $O/tests_split_on.txt
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l["roofline"]; print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), "frac", r["frac"], "sync", l["latency_ms_sync"])'
for r in 1 2 3; do
  for v in 0 1; do SVGF_SPLIT_FUSED=$v timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" split$v >> $O/bench.log 2>&1; done
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  SVGF_SPLIT_FUSED=$v rocprofv3 --kernel-trace --stats -d $O/prof$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/probe.py --variants 0 --frames 24 --reps 200 --sustain 0.6 > /dev/null 2>&1
  python - $(find $O/prof$v -name '*kernel_stats.csv' | head -1) $v >> $O/bench.log <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "atrous" in r["Name"] or "temporal" in r["Name"]]
print("rocprofv3 split" + sys.argv[2] + ":", " | ".join(f"{r['Name'][43:70]} {float(r['AverageNs'])/1e3:.2f}" for r in rows), "| sum", round(sum(float(r['AverageNs']) for r in rows) / 1e3 * 1.0 / 1, 1))
PY
  rm -rf $O/prof$v
done

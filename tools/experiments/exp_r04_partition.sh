#!/bin/bash
cd $GRAFT_REPO_ROOT
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["ms_per_step"], l["kernels_us"])'
timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" baseline
for n in 24 32 40 48 64; do
  SVGF_EXP_PARTITION=$n SVGF_EXP_NCU=$((256-n)) timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "$pick" "T=$n,L=$((256-n))" 2>&1 | tail -1
done
SVGF_EXP_PARTITION=32 SVGF_EXP_NCU=224 SVGF_EXP_TFIRST=1 timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "$pick" "T=first32" 2>&1 | tail -1
SVGF_EXP_NCU=224 timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" "no-partition,224-WG-tiling"

#!/bin/bash
# round 4: whole GPU suite on the current tree, then A/B of prebuilt libraries (A = tree, B = centre prefetch on every lane variant)
O=gpurun_out/r04_suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt
bash tools/experiments/exp_ab_multi.sh "A B" 4 3 > $O/ab_centre_prefetch.log 2>&1

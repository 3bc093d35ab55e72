#!/bin/bash
# round 4: fused kernel v5 after the rotation fix: full parity (fused tests + goldens with every variant), A/B timing
O=gpurun_out/r04_fused6; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -q 2>&1 | tail -8 > $O/test_fused.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -k "goldens or sequences_match or back_to_back" 2>&1 | tail -8 > $O/test_goldens.txt
for v in 6 4 5; do timeout 300 python tools/probe.py --variants $v --reps 200 > $O/probe_v$v.log 2>&1; done

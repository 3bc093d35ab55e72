#!/bin/bash
# round 4: first run of the fused temporal + first-level kernel: parity, then timing
O=gpurun_out/r04_fused1; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -25 > $O/test_fused.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -k "goldens or sequences_match" 2>&1 | tail -25 > $O/test_goldens.txt
for v in 4 5 6 0; do timeout 300 python tools/probe.py --variants $v > $O/probe_v$v.log 2>&1; done

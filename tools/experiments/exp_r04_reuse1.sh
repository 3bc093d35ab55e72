#!/bin/bash
# round 4: cross-level reuse of the geometric terms: parity, then A/B on one box (baseline: the default; SVGF_REUSE=1 turns the reuse on.
# The three logged runs were made while the reuse was the default and SVGF_NO_REUSE=1 the baseline.)
O=gpurun_out/r04_reuse1; mkdir -p $O
SVGF_REUSE=1 timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "goldens or sequences_match or randomised or 1080p_full or lane_kernel_at_steps" 2>&1 | tail -8 > $O/test_parity.txt
SVGF_REUSE=1 timeout 600 python -m pytest tests/test_ref_scenes.py tests/test_paper_steps.py -x -q -m gpu 2>&1 | tail -5 > $O/test_scenes.txt
for r in 1 2 3; do
  timeout 300 python tools/probe.py --variants 4 --reps 400 > $O/probe_noreuse_$r.log 2>&1
  SVGF_REUSE=1 timeout 300 python tools/probe.py --variants 4 --reps 400 > $O/probe_reuse_$r.log 2>&1
done
SVGF_REUSE=1 timeout 300 python tools/probe.py --variants 4 --reps 100 --size 3840x2160 --frames 8 > $O/probe_reuse_4k.log 2>&1
timeout 300 python tools/probe.py --variants 4 --reps 100 --size 3840x2160 --frames 8 > $O/probe_noreuse_4k.log 2>&1

#!/bin/bash
# round 4: fused kernel v5 (one pixel per sub-step): parity, timing (AoS and planar inputs), timeline of a loaded and a light workgroup
O=gpurun_out/r04_fused5; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -5 > $O/test_fused.txt
for v in 6 4; do timeout 300 python tools/probe.py --variants $v --reps 200 > $O/probe_v$v.log 2>&1; done
timeout 300 python tools/probe.py --variants 6 --reps 100 --planar > $O/probe_planar_v6.log 2>&1
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
for b in 200 40; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=5 timeout 300 python tools/probe.py --variants 6 --frames 3 2>&1 | grep -E "lane dbg|prologue|it +[0-9]+:" | head -45; done > $O/timeline_fused.log 2>&1

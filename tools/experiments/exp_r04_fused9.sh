#!/bin/bash
# round 4: fused kernel after the instruction diet (arena offsets, separable on-screen mask, one window computation, per-thread
# invariants of the loop's pixels, one shared division, vector-typed stage A) + whole-bench A/B against the round-3 tree (_r03/)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_fused9; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_planar_inputs.py -x -q 2>&1 | tail -5 > $O/test_fused.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "golden" 2>&1 | tail -3 > $O/test_goldens.txt
for v in 6 4; do timeout 300 python tools/probe.py --variants $v --reps 200 > $O/probe_v$v.log 2>&1; done
timeout 300 python tools/probe.py --variants 6 --reps 100 --planar > $O/probe_planar_v6.log 2>&1
pick='import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["ms_per_step"], l.get("kernels_us"), l.get("latency_ms_sync"), l.get("warmup_steps_run"))'
for r in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/_r03 && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/r03_err.log | python -c "$pick" r03) >> $O/bench_ab.log 2>&1
  (cd $GRAFT_REPO_ROOT && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" r04) >> $O/bench_ab.log 2>&1
done
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" timeout 900 python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
for b in 200 40; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=5 timeout 300 python tools/probe.py --variants 6 --frames 3 2>&1 | grep -E "lane dbg|prologue|it +[0-9]+:" | head -45; done > $O/timeline_fused.log 2>&1

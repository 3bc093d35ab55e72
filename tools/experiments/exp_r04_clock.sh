#!/bin/bash
# round 4: clock states (tools/clock_states.py) + the whole GPU suite on the arena build
O=$GRAFT_REPO_ROOT/gpurun_out/r04_clock; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/clock_states.py --json $O/clock_states.json > $O/clock_states.txt 2>$O/clock_err.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.txt

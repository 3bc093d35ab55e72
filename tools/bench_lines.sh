#!/bin/bash
# The bench lines of the evidence table, fresh process per line: bash tools/bench_lines.sh <tag> -> gpurun_out/final_<tag>/
TAG=${1:-r06}; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/final_$TAG; mkdir -p $O; cd $R
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 $([ $i != 1 ] && echo --no-cpu-baseline) > $O/bench_line_driver_cmd_$i.json 2>> $O/bench_driver_cmd.err; done
cp $O/bench_line_driver_cmd_1.json $O/bench_line_driver_cmd.json
python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline > $O/bench_line_driver_cmd_no_pipeline.json 2>> $O/bench_driver_cmd.err
GPU_MAX_HW_QUEUES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_driver_cmd_one_hw_queue.json 2>> $O/bench_driver_cmd.err
python bench.py --no-cpu-baseline > $O/bench_line.json 2>/dev/null
python bench.py --no-cpu-baseline --planar-inputs > $O/bench_line_planar_inputs.json 2>/dev/null
python bench.py --no-cpu-baseline --config 1080p-moving > $O/bench_line_1080p_moving.json 2>/dev/null
python bench.py --no-cpu-baseline --config 4k-static > $O/bench_line_4k_static.json 2>/dev/null
python bench.py --no-cpu-baseline --config 4k-room > $O/bench_line_4k_room.json 2>/dev/null
python bench.py --config config1 > $O/bench_line_config1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cadence-hz 144 --cadence-frames 100 > $O/bench_line_cadence_144hz.json 2>/dev/null

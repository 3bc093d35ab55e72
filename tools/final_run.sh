#!/bin/bash
# Evidence run for profiles/: tests, bench lines (the driver's command in fresh processes, the other BASELINE configs), rocprofv3 kernel
# stats of the bench command at 1080p AND at 4K (ordered = the kernels' own durations, and as the bench runs), PMC passes (SQ + HBM
# traffic, FETCH and WRITE in separate passes as MI355X_MICROARCH.md prescribes) at 1080p AND at 4K.  Run on the GPU box:
#   bash tools/final_run.sh <tag> [quick]     -> everything lands under gpurun_out/final_<tag>/
# then, back in the build container, tools/final_collect.sh <tag> copies what is to be judged into profiles/ as <tag>_* and rebuilds
# profiles/pmc_traffic.json.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final_$TAG
mkdir -p $O
cd $R
if [ "$2" != "quick" ]; then
timeout 2400 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|error|worst|pipelined|PASSED.*(4k|room|1080p)|max rel" | tail -40 > $O/gpu_tests.txt
fi
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
python tools/clock_states.py --json $O/clock_states.json > $O/clock_states.txt 2>/dev/null
python tools/probe.py --variants 0,4,2 --reps 200 --sustain 0.6 > $O/probe_1080p.log 2>&1
python tools/probe.py --variants 0,2 --size 3840x2160 --frames 8 --reps 60 --sustain 0.6 > $O/probe_4k.log 2>&1
python tools/region_dynamics.py > $O/region_dynamics.txt 2>&1
python tools/mode_switch_timeline.py > $O/mode_switch_timeline.txt 2>&1
examples/farm 8 32 1920 1080 > $O/farm_cpp_8_contexts_1080p.txt 2>&1
(examples/pipeline 400 1920 1080; examples/pipeline 600 1280 720; examples/pipeline 200 3840 2160) > $O/pipeline_cpp_two_streams.txt 2>&1
python tools/soak.py --frames 6000 --random 97 > $O/pipeline_soak.log 2>&1
python tools/soak.py --frames 1500 --random 97 --size 801x603 >> $O/pipeline_soak.log 2>&1
cd /tmp && export TMPDIR=/tmp
# ---- 1080p: kernel trace of the driver's command, SQ pass, HBM passes ----
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --cadence-hz 0 > $O/bench_line_under_rocprofv3.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench.csv 2>/dev/null
rm -rf $O/prof
# the same with every frame ordered on the one stream: one kernel at a time on the GPU, the kernels' own durations
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --cadence-hz 0 > $O/bench_line_no_pipeline_under_rocprofv3.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_ordered.csv 2>/dev/null
rm -rf $O/prof
rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq "atrous" > $O/pmc_sq.txt
python $R/tools/pmc_summary.py $O/pmc_sq "k_temporal" >> $O/pmc_sq.txt
rm -rf $O/pmc_sq
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm "atrous" > $O/pmc_hbm.txt
python $R/tools/pmc_summary.py $O/pmc_hbm "k_temporal" >> $O/pmc_hbm.txt
rm -rf $O/pmc_hbm
# ---- 3840x2160 (configs[3]): the same three ----
rocprofv3 --kernel-trace --stats -d $O/prof4k -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --config 4k-static --cadence-hz 0 > $O/bench_line_4k_no_pipeline_under_rocprofv3.json 2>/dev/null
cp $(find $O/prof4k -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_4k_ordered.csv 2>/dev/null
rm -rf $O/prof4k
rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq4k -o p --output-format csv -- python $R/tools/probe.py --variants 0 --size 3840x2160 --frames 6 --reps 10 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq4k "atrous" > $O/pmc_sq_4k.txt
rm -rf $O/pmc_sq4k
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm4k -o p --output-format csv -- python $R/tools/probe.py --variants 0 --size 3840x2160 --frames 6 --reps 10 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm4k "atrous" > $O/pmc_hbm_4k.txt
python $R/tools/pmc_summary.py $O/pmc_hbm4k "k_temporal" >> $O/pmc_hbm_4k.txt
rm -rf $O/pmc_hbm4k
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/gpu_box.txt; nproc >> $O/gpu_box.txt; grep -m1 "model name" /proc/cpuinfo >> $O/gpu_box.txt
ls -la $O

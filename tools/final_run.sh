#!/bin/bash
# Evidence run for profiles/ (round 5): tests, bench lines (the driver's command, the other BASELINE configs), rocprofv3 kernel
# stats of the bench command at 1080p AND at 4K, PMC passes (SQ + HBM traffic, FETCH and WRITE in separate passes as
# MI355X_MICROARCH.md prescribes) at 1080p AND at 4K, the 4K tile / segment A/B.  Run on the GPU box:
#   bash tools/final_run.sh [quick]     -> everything lands under gpurun_out/final_r05/
# then copy what is to be judged into profiles/ as r05_* and run tools/pmc_traffic_update.py (tools/final_collect.sh does both).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final_r05
mkdir -p $O
cd $R
if [ "$1" != "quick" ]; then
timeout 2400 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|error|worst|pipelined|PASSED.*(4k|room|1080p)|max rel" | tail -40 > $O/gpu_tests.txt
fi
python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --steps 20 --warmup 5 --no-pipeline > $O/bench_line_driver_cmd_no_pipeline.json 2>> $O/bench_driver_cmd.err
python bench.py --no-cpu-baseline > $O/bench_line.json 2>/dev/null
python bench.py --no-cpu-baseline --planar-inputs > $O/bench_line_planar_inputs.json 2>/dev/null
python bench.py --no-cpu-baseline --config 1080p-moving > $O/bench_line_1080p_moving.json 2>/dev/null
python bench.py --no-cpu-baseline --config 4k-static > $O/bench_line_4k_static.json 2>/dev/null
python bench.py --no-cpu-baseline --config 4k-room > $O/bench_line_4k_room.json 2>/dev/null
python bench.py --config config1 > $O/bench_line_config1.json 2>/dev/null
python bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_line_200_steps.json 2>/dev/null
python tools/clock_states.py --json $O/clock_states.json > $O/clock_states.txt 2>/dev/null
python tools/probe.py --variants 0,4,2 --reps 200 --sustain 0.6 > $O/probe_1080p.log 2>&1
# configs[3] / the "LDS-tile sizing" run: the lane kernel's 480-column x 6-row ring tiles against the strip kernel's 256-column x 2-row tiles
python tools/probe.py --variants 0,2 --size 3840x2160 --frames 8 --reps 60 --sustain 0.6 > $O/probe_4k.log 2>&1
examples/farm 8 32 1920 1080 > $O/farm_cpp_8_contexts_1080p.txt 2>&1
(examples/pipeline 400 1920 1080; examples/pipeline 600 1280 720; examples/pipeline 200 3840 2160) > $O/pipeline_cpp_two_streams.txt 2>&1
python tools/experiments/exp_r05_pipeline.py > $O/pipeline_vs_ordered_1080p.log 2>&1
python tools/experiments/exp_r05_pipeline.py --moving --soak 20000 > $O/pipeline_soak.log 2>&1
cd /tmp && export TMPDIR=/tmp
# ---- 1080p: kernel trace of the driver's command, SQ pass, HBM passes ----
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_under_rocprofv3.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench.csv 2>/dev/null
rm -rf $O/prof
# the same with every frame ordered on the one stream: one kernel at a time on the GPU, the kernels' own durations
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > $O/bench_line_no_pipeline_under_rocprofv3.json 2>/dev/null
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
rocprofv3 --kernel-trace --stats -d $O/prof4k -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --config 4k-static > $O/bench_line_4k_under_rocprofv3.json 2>/dev/null
cp $(find $O/prof4k -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_4k.csv 2>/dev/null
rm -rf $O/prof4k
rocprofv3 --kernel-trace --stats -d $O/prof4k -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --config 4k-static > $O/bench_line_4k_no_pipeline_under_rocprofv3.json 2>/dev/null
cp $(find $O/prof4k -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_4k_ordered.csv 2>/dev/null
rm -rf $O/prof4k
rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq4k -o p --output-format csv -- python $R/tools/probe.py --variants 0 --size 3840x2160 --frames 6 --reps 10 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq4k "atrous" > $O/pmc_sq_4k.txt
rm -rf $O/pmc_sq4k
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm4k -o p --output-format csv -- python $R/tools/probe.py --variants 0 --size 3840x2160 --frames 6 --reps 10 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm4k "atrous" > $O/pmc_hbm_4k.txt
python $R/tools/pmc_summary.py $O/pmc_hbm4k "k_temporal" >> $O/pmc_hbm_4k.txt
rm -rf $O/pmc_hbm4k
# planar boundary: kernel trace
rocprofv3 --kernel-trace --stats -d $O/prof2 -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --planar-inputs > /dev/null 2>&1
cp $(find $O/prof2 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_planar.csv 2>/dev/null
rm -rf $O/prof2
# ---- 4K segment length (experiments build: svgf_exp_set("lane_segrows")): the automatic choice (68 rows, one round of 256 workgroups)
#      against its neighbours 34 (two rounds), 45 (a round and a half), 136 (half the CUs) ----
export SVGF_USE_EXPERIMENTS_LIB=1
for rows in 0 34 45 68 136; do
  if [ $rows = 0 ]; then unset SVGF_LANE_SEGROWS; else export SVGF_LANE_SEGROWS=$rows; fi
  rocprofv3 --kernel-trace --stats -d $O/seg4k -o p --output-format csv -- python $R/tools/probe.py --size 3840x2160 --variants 4 --frames 8 --reps 60 --sustain 0.6 > /dev/null 2>&1
  python - "$rows" "$O" >> $O/segment_length_4k.log <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[2] + "/seg4k/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_atrous_lane" in r["Name"]]
step = lambda n: 1 << int(re.search(r"k_atrous_lane<(\d+)", n).group(1))
rows.sort(key=lambda r: step(r["Name"]))
print(f"3840x2160 seg_rows {sys.argv[1]:>3} (0 = automatic = 68):", " ".join(f"step{step(r['Name'])}={float(r['AverageNs'])/1e3:.1f}us" for r in rows),
      f"| sum {sum(float(r['AverageNs']) for r in rows)/1e3:.1f} us")
PY
  rm -rf $O/seg4k
done
unset SVGF_LANE_SEGROWS SVGF_USE_EXPERIMENTS_LIB
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/gpu_box.txt; nproc >> $O/gpu_box.txt; grep -m1 "model name" /proc/cpuinfo >> $O/gpu_box.txt
ls -la $O

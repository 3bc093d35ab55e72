#!/bin/bash
# Evidence run for profiles/ (round 4): tests, bench lines (same command the driver uses + the long run), rocprofv3 kernel
# stats of the bench command, PMC passes (SQ + HBM traffic, FETCH and WRITE in separate passes as MI355X_MICROARCH.md
# prescribes).  Run on the GPU box:  bash tools/final_run.sh   -> everything lands under gpurun_out/final_r04/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final_r04
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --no-cpu-baseline > $O/bench_line.json 2>/dev/null
python bench.py --no-cpu-baseline --planar-inputs > $O/bench_line_planar_inputs.json 2>/dev/null
python bench.py --no-cpu-baseline --config 1080p-moving > $O/bench_line_1080p_moving.json 2>/dev/null
python bench.py --no-cpu-baseline --config 4k-static > $O/bench_line_4k_static.json 2>/dev/null
python bench.py --config config1 > $O/bench_line_config1.json 2>/dev/null
python bench.py --config config1 --no-cpu-baseline --kernel-variant 4 > $O/bench_line_config1_unfused_prepare.json 2>/dev/null
python bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_line_200_steps.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-variant 6 > $O/bench_line_fused_forced.json 2>/dev/null
python tools/clock_states.py --json $O/clock_states.json > $O/clock_states.txt 2>/dev/null
python tools/probe.py --variants 0,4,5,6 --reps 200 > $O/probe_1080p.log 2>&1
python tools/probe.py --variants 0 --size 3840x2160 --frames 8 > $O/probe_4k.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench.csv 2>/dev/null
rm -rf $O/prof
rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq "atrous" > $O/pmc_sq.txt
python $R/tools/pmc_summary.py $O/pmc_sq "k_temporal" >> $O/pmc_sq.txt
rm -rf $O/pmc_sq
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm "atrous" > $O/pmc_hbm.txt
python $R/tools/pmc_summary.py $O/pmc_hbm "k_temporal" >> $O/pmc_hbm.txt
rm -rf $O/pmc_hbm
SVGF_NO_VARIANCE_PLANE=1 rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm2 -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm2 "atrous" > $O/pmc_hbm_no_variance_plane.txt
rm -rf $O/pmc_hbm2
rocprofv3 --kernel-trace --stats -d $O/prof2 -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --planar-inputs > /dev/null 2>&1
cp $(find $O/prof2 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_planar.csv 2>/dev/null
rm -rf $O/prof2
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm3 -o p --output-format csv -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --planar-inputs --min-warmup-seconds 0.1 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm3 "k_temporal" > $O/pmc_hbm_planar_temporal.txt
rm -rf $O/pmc_hbm3
# the fused temporal + first-level kernel (kernel_variant 6, parked): its own rocprofv3 line and PMC passes
rocprofv3 --kernel-trace --stats -d $O/prof3 -o p --output-format csv -- python $R/tools/probe.py --variants 6 --frames 24 > /dev/null 2>&1
cp $(find $O/prof3 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_fused_probe.csv 2>/dev/null
rm -rf $O/prof3
rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq6 -o p --output-format csv -- python $R/tools/probe.py --variants 6 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq6 "atrous" > $O/pmc_sq_fused.txt
rm -rf $O/pmc_sq6
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm6 -o p --output-format csv -- python $R/tools/probe.py --variants 6 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm6 "atrous" > $O/pmc_hbm_fused.txt
rm -rf $O/pmc_hbm6
# cross-level reuse of the geometric terms (SVGF_REUSE=1, parked)
SVGF_REUSE=1 rocprofv3 --kernel-trace --stats -d $O/prof4 -o p --output-format csv -- python $R/tools/probe.py --variants 4 --frames 24 > /dev/null 2>&1
cp $(find $O/prof4 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_reuse_probe.csv 2>/dev/null
rm -rf $O/prof4
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/gpu_box.txt; nproc >> $O/gpu_box.txt; grep -m1 "model name" /proc/cpuinfo >> $O/gpu_box.txt
ls -la $O

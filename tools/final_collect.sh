#!/bin/bash
# After tools/final_run.sh came back (gpurun merges gpurun_out/final_r05/ into this container): copy what is to be judged into
# profiles/ as r05_* and rebuild the hashed PMC record bench.py reads (profiles/pmc_traffic.json).  Run HERE, from the repo root.
O=gpurun_out/final_r05
P=profiles
for f in gpu_tests.txt bench_line_driver_cmd.json bench_line.json bench_line_planar_inputs.json bench_line_1080p_moving.json bench_line_4k_static.json \
         bench_line_4k_room.json bench_line_config1.json bench_line_200_steps.json clock_states.json clock_states.txt probe_1080p.log probe_4k.log \
         farm_cpp_8_contexts_1080p.txt rocprofv3_kernel_stats_bench.csv rocprofv3_kernel_stats_4k.csv rocprofv3_kernel_stats_bench_planar.csv \
         pmc_sq.txt pmc_hbm.txt pmc_sq_4k.txt pmc_hbm_4k.txt segment_length_4k.log gpu_box.txt bench_line_under_rocprofv3.json bench_line_4k_under_rocprofv3.json \
         bench_line_driver_cmd_no_pipeline.json rocprofv3_kernel_stats_bench_ordered.csv rocprofv3_kernel_stats_4k_ordered.csv \
         bench_line_no_pipeline_under_rocprofv3.json bench_line_4k_no_pipeline_under_rocprofv3.json pipeline_vs_ordered_1080p.log pipeline_soak.log pipeline_cpp_two_streams.txt; do
  [ -s $O/$f ] && cp $O/$f $P/r05_$f
done
sclk() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["telemetry"]["timed_region"]["sclk_mhz"]["median"])
except Exception:
    print(2390)
PY
}
python tools/pmc_traffic_update.py hbm $P/r05_pmc_hbm.txt 1920x1080
python tools/pmc_traffic_update.py hbm $P/r05_pmc_hbm_4k.txt 3840x2160
# (the launch durations the SQ counters are divided by: the ORDERED runs' — the counter passes run one kernel at a time too)
python tools/pmc_traffic_update.py sq $P/r05_pmc_sq.txt $P/r05_rocprofv3_kernel_stats_bench_ordered.csv $(sclk $P/r05_bench_line_no_pipeline_under_rocprofv3.json) 1920x1080
python tools/pmc_traffic_update.py sq $P/r05_pmc_sq_4k.txt $P/r05_rocprofv3_kernel_stats_4k_ordered.csv $(sclk $P/r05_bench_line_4k_no_pipeline_under_rocprofv3.json) 3840x2160

#!/bin/bash
# After tools/final_run.sh <tag> came back (gpurun merges gpurun_out/final_<tag>/ into this container): copy what is to be judged into
# profiles/ as <tag>_* and rebuild the hashed PMC record bench.py reads (profiles/pmc_traffic.json).  Run HERE, from the repo root.
TAG=${1:-r06}
O=gpurun_out/final_$TAG
P=profiles
for f in $(ls $O); do
  case $f in *.err|prof*|pmc_sq|pmc_hbm) continue;; esac
  [ -s $O/$f ] && cp $O/$f $P/${TAG}_$f
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
python tools/pmc_traffic_update.py hbm $P/${TAG}_pmc_hbm.txt 1920x1080
python tools/pmc_traffic_update.py hbm $P/${TAG}_pmc_hbm_4k.txt 3840x2160
# (the launch durations the SQ counters are divided by: the ORDERED runs' — the counter passes run one kernel at a time too)
python tools/pmc_traffic_update.py sq $P/${TAG}_pmc_sq.txt $P/${TAG}_rocprofv3_kernel_stats_bench_ordered.csv $(sclk $P/${TAG}_bench_line_no_pipeline_under_rocprofv3.json) 1920x1080
python tools/pmc_traffic_update.py sq $P/${TAG}_pmc_sq_4k.txt $P/${TAG}_rocprofv3_kernel_stats_4k_ordered.csv $(sclk $P/${TAG}_bench_line_4k_no_pipeline_under_rocprofv3.json) 3840x2160

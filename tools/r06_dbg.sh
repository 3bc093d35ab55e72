mkdir -p gpurun_out/r06_dbg
for i in 1 2; do SVGF_BENCH_DEBUG=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --cadence-hz 0 > gpurun_out/r06_dbg/b$i.json 2> gpurun_out/r06_dbg/b$i.err; grep debug gpurun_out/r06_dbg/b$i.err; python -c "
import json; d=json.loads(open('gpurun_out/r06_dbg/b$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['frame_pipeline_trial']['pipelined_ms'], d['frame_pipeline_trial']['ordered_ms'], d['idle_before_timed_region_ms'])"; done

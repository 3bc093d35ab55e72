#!/bin/bash
# round 4, first call: what telemetry does the GPU box offer, and the baseline numbers of HEAD on this box
O=gpurun_out/r04_box; mkdir -p $O
ls -la /sys/class/drm/ > $O/sysfs.txt 2>&1
for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $d"; ls $d; for f in freq1_input freq1_label power1_average power1_input temp1_input temp1_label; do echo -n "$f: "; cat $d/$f 2>&1; done; done >> $O/sysfs.txt 2>&1
rocm-smi --showclocks --showpower --showtemp --json > $O/smi_idle.json 2>&1
python tools/telemetry.py 0.5 > $O/telemetry_idle.json 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd.json 2> $O/bench.err
python tools/probe.py --variants 0 > $O/probe_1080p.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.txt

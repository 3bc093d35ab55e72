python tools/probe.py --variants 1,2 --check --frames 4 2>&1 | grep -E "frame |atrous" | tail -9
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --config 4k-static 2>/dev/null | tail -1 | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2

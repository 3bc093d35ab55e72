timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -k "lattice or sequences or randomised" 2>&1 | tail -3
python tools/probe.py --variants 0,1 --frames 4 --nlevel 7 2>&1 | grep -E "atrous|variant|frame" | head -24
python tools/probe.py --size 3840x2160 --variants 0 --frames 4 --nlevel 7 2>&1 | grep -E "atrous" | head -8

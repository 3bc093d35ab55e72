for F in "" "-DSVGF_LATTICE_NT=512 -DSVGF_LATTICE_LDS_KB=75" "-DSVGF_LATTICE_NT=256 -DSVGF_LATTICE_LDS_KB=37"; do
SVGF_EXTRA_HIPCC_FLAGS="$F" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== flags: $F"
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -k "lattice" 2>&1 | tail -1
python tools/probe.py --variants 0 --frames 4 --nlevel 7 2>&1 | grep -E "step  64|step 128"
python tools/probe.py --size 3840x2160 --variants 0 --frames 4 --nlevel 7 2>&1 | grep -E "step  64|step 128"
done

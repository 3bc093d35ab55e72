# where does the DMA kernel's time go: (a) as built, (b) without the A / B dword gathers in the loop, (c) without any staging
# in the loop (b, c: wrong results, timing only)
for F in "" "-DSVGF_DMA_EXP_NOGATHER" "-DSVGF_DMA_EXP_NOSTAGE"; do
  export SVGF_EXTRA_HIPCC_FLAGS="$F"
  echo "== flags: $F"
  python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v "amdgpu.ids\|inline asm\|Reserved" | tail -2
  python tools/probe.py --variants 0 --frames 8 2>&1 | grep -E "atrous|frame wall" | head -6
done
export SVGF_EXTRA_HIPCC_FLAGS=""
python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v "amdgpu.ids\|inline asm\|Reserved" | tail -2
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -k "goldens" 2>&1 | tail -25

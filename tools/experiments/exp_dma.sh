# LDS-DMA 12-wave kernel: parity, then per-level timings against the shipped lane / strip selection (SVGF_NO_DMA=1)
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -5
echo "== default (DMA kernel for steps 2..16)"
python tools/probe.py --variants 0 --frames 8 2>&1 | grep -E "atrous|frame wall|temporal" | head -7
echo "== SVGF_NO_DMA=1 (lane for 2-8, strip for 16-32)"
SVGF_NO_DMA=1 python tools/probe.py --variants 0 --frames 8 2>&1 | grep -E "atrous|frame wall|temporal" | head -7
echo "== 4K default"
python tools/probe.py --variants 0 --frames 6 --size 3840x2160 2>&1 | grep -E "atrous|frame wall|temporal" | head -7
echo "== 4K SVGF_NO_DMA=1"
SVGF_NO_DMA=1 python tools/probe.py --variants 0 --frames 6 --size 3840x2160 2>&1 | grep -E "atrous|frame wall|temporal" | head -7

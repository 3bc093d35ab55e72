echo "== share off"; python tools/probe.py --variants 2 --frames 6 2>&1 | grep -E "atrous" | head -3
echo "== share on"; SVGF_STRIP_SHARE=1 python tools/probe.py --variants 2 --frames 6 2>&1 | grep -E "atrous" | head -3

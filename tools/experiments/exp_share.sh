SVGF_SHARE_DBG=40 python tools/probe.py --variants 3 --frames 2 2>&1 | grep -E "share dbg|wave|loader" | head -80

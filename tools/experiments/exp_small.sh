for tx in 256 128; do for sr in 0 13 17 25 34 50 100; do
  echo "== TX=$tx SEGROWS=$sr"; SVGF_STRIP_TX=$tx SVGF_STRIP_SEGROWS=$sr python tools/probe.py --size 800x800 --nlevel 2 --variants 2 --frames 6 2>&1 | grep -E "atrous" | tr '\n' ' '; echo
done; done

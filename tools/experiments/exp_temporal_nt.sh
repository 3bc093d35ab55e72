for F in "-DSVGF_TEMPORAL_NT" ""; do
SVGF_EXTRA_HIPCC_FLAGS="$F" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== flags: $F"
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "goldens or overlap" 2>&1 | tail -1
for c in 1080p-static 1080p-moving 4k-static 4k-moving; do for f in "" "--no-overlap"; do python bench.py --no-cpu-baseline --config $c $f 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_step'], d['kernels_us'])" "$c" "$f"; done; done
done

cd $GRAFT_REPO_ROOT
cp cuda-path-tracer-denoising_amd/libsvgf_hip.so.T cuda-path-tracer-denoising_amd/libsvgf_hip_exp.so
export SVGF_USE_EXPERIMENTS_LIB=1
for b in 40 7 200; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=10 python tools/probe.py --variants 4 --frames 6 2>&1 | grep -E "lane dbg|prologue|it  ?[0-9]+:" | head -120; done

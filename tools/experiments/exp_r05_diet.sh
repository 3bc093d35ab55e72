# round 5: the loader / prologue diet of the lane kernel (SVGF_LANE_DIET): parity, A/B against round 4's staging, in-kernel timeline
# libraries built in the container: exp_r05_build_variants.sh "A=-DSVGF_LANE_DIET=0" "B=-DSVGF_LANE_DIET=1" "T=-DSVGF_LANE_TIMELINE"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
if [ "${1:-all}" != "ab" ]; then
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_prepare_fused_gpu.py tests/test_fuzz_gpu.py tests/test_planar_inputs.py -m gpu -x -q 2>&1 | tail -8
fi
bash tools/experiments/exp_ab_multi.sh "A B" 4 ${2:-3} 2>&1 | grep -v amdgpu.ids
cp cuda-path-tracer-denoising_amd/libsvgf_hip.so.T cuda-path-tracer-denoising_amd/libsvgf_hip_exp.so
export SVGF_USE_EXPERIMENTS_LIB=1
for b in 40 200; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=10 python tools/probe.py --variants 4 --frames 6 2>&1 | grep -E "lane dbg|prologue|it  ?[0-3]:" | head -80; done

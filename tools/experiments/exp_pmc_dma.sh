# SQ counters of the 12-wave DMA kernel (built without in-loop staging: pure compute, wrong results) beside the lane / strip kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_DMA_EXP_NOSTAGE"
(cd $R && python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v "amdgpu.ids\|inline asm\|Reserved" | tail -2)
for P in pmc1 pmc2; do
  rocprofv3 -i $R/tools/$P.txt -d $R/gpurun_out/pmc_dma_$P -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_dma_$P "k_atrous_dma<3, 256, 3, true"
  SVGF_NO_DMA=1 rocprofv3 -i $R/tools/$P.txt -d $R/gpurun_out/pmc_old_$P -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_old_$P "lane<3, true"
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_old_$P "strip<4, 256, 2, true"
  rm -rf $R/gpurun_out/pmc_dma_$P $R/gpurun_out/pmc_old_$P
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -i $R/tools/pmc2.txt -d $R/gpurun_out/pmc_off -o p --output-format csv -- python $R/tools/probe.py --variants 2 --frames 6 > /dev/null 2>&1
SVGF_STRIP_SHARE=1 rocprofv3 -i $R/tools/pmc2.txt -d $R/gpurun_out/pmc_on -o p --output-format csv -- python $R/tools/probe.py --variants 2 --frames 6 > /dev/null 2>&1
echo "== share off"; python $R/tools/pmc_summary.py $R/gpurun_out/pmc_off "strip<2, 256, 2, true"
echo "== share on"; python $R/tools/pmc_summary.py $R/gpurun_out/pmc_on "strip<2, 256, 2, true"

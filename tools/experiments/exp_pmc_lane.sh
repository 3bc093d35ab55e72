cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -i $R/tools/pmc2.txt -d $R/gpurun_out/pmc_lane -o p --output-format csv -- python $R/tools/probe.py --variants 4 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_lane "lane<2, true"
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_lane "strip<4, 256, 2, true"
rm -rf $R/gpurun_out/pmc_lane

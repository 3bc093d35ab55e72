cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -i $R/tools/pmc2.txt -d $R/gpurun_out/pmc_lat -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 4 --nlevel 7 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_lat "lattice<true"
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_lat "lane<2, true"
rm -rf $R/gpurun_out/pmc_lat

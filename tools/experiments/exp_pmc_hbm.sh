cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -i $R/tools/pmc_traffic.txt -d $R/gpurun_out/pmc_hbm -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_hbm "atrous" > $R/gpurun_out/pmc_hbm.txt
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_hbm "k_temporal" >> $R/gpurun_out/pmc_hbm.txt
rm -rf $R/gpurun_out/pmc_hbm
cat $R/gpurun_out/pmc_hbm.txt | cut -c1-140

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BLUR=${1:-0}
rocprofv3 -i $R/tools/pmc2.txt -d $R/gpurun_out/pmc_pw -o p --output-format csv -- python $R/tools/probe.py --variants 5 --frames 6 --blur $BLUR > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_pw "pwILi1ELb1" 
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_pw "pw<1, true"
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_pw "pw<5, false"
rm -rf $R/gpurun_out/pmc_pw

#!/bin/bash
# round 4: PMC passes of the lane kernels with cross-level reuse of geometric terms (SVGF_REUSE=1), beside the default path's (profiles/r04_pmc_sq.txt)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_reuse_pmc; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SVGF_REUSE=1 rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq -o p --output-format csv -- python $R/tools/probe.py --variants 4 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq "atrous" > $O/pmc_sq_reuse.txt
rm -rf $O/pmc_sq
SVGF_REUSE=1 rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm -o p --output-format csv -- python $R/tools/probe.py --variants 4 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_hbm "atrous" > $O/pmc_hbm_reuse.txt
rm -rf $O/pmc_hbm

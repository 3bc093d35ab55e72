#!/bin/bash
# round 4: why a VMEM instruction of the fused kernel's loader waves takes ~300 ticks: counters, AoS vs planar inputs, finer timeline
O=gpurun_out/r04_fused4; mkdir -p $O
R=$PWD
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TCC|SQ|TD)_[A-Za-z0-9_]+" | sort -u > $O/counters_available.txt
for v in 6 4; do timeout 300 python tools/probe.py --variants $v --reps 100 --planar > $O/probe_planar_v$v.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 -i $R/tools/pmc_vmem.txt -d $R/$O/pmc_vmem -o p --output-format csv -- python $R/tools/probe.py --variants 6 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/$O/pmc_vmem "atrous" > $R/$O/pmc_vmem.txt; python $R/tools/pmc_summary.py $R/$O/pmc_vmem "k_temporal" >> $R/$O/pmc_vmem.txt
rm -rf $R/$O/pmc_vmem
timeout 600 rocprofv3 -i $R/tools/pmc_vmem.txt -d $R/$O/pmc_vmem4 -o p --output-format csv -- python $R/tools/probe.py --variants 4 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/$O/pmc_vmem4 "atrous" > $R/$O/pmc_vmem_v4.txt; python $R/tools/pmc_summary.py $R/$O/pmc_vmem4 "k_temporal" >> $R/$O/pmc_vmem_v4.txt
rm -rf $R/$O/pmc_vmem4
cd $R
SVGF_EXTRA_HIPCC_FLAGS="-DSVGF_LANE_TIMELINE" python -c "
import sys
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True)" 2>&1 | grep -v amdgpu.ids | tail -2
for b in 200 201; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=5 timeout 300 python tools/probe.py --variants 6 --frames 3 2>&1 | grep -E "lane dbg|prologue|it +[0-9]+:" | head -45; done > $O/timeline_fused.log 2>&1
for b in 200; do SVGF_LANE_DBG=$b SVGF_LANE_DBG_SKIP=5 timeout 300 python tools/probe.py --variants 6 --frames 3 --planar 2>&1 | grep -E "lane dbg|prologue|it +[0-9]+:" | head -45; done > $O/timeline_fused_planar.log 2>&1

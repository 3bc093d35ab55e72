# round-end evidence run: tests, smoke, bench (3 configs), rocprofv3 kernel stats, PMC (SQ + HBM traffic), per-kernel probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_1080p_static.json
python bench.py --no-cpu-baseline --config 1080p-moving 2>/dev/null | tail -1 > $O/bench_1080p_moving.json
python bench.py --no-cpu-baseline --config 4k-static 2>/dev/null | tail -1 > $O/bench_4k_static.json
cut -c1-200 $O/bench_*.json
python tools/probe.py --variants 1,0 --check --frames 6 2>&1 | grep -v amdgpu.ids > $O/probe_1080p.log
python tools/probe.py --size 3840x2160 --variants 0 --frames 6 2>&1 | grep -v amdgpu.ids > $O/probe_4k.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r01 -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 -i $R/tools/pmc2.txt -d $O/pmc_sq -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
rocprofv3 -i $R/tools/pmc_traffic.txt -d $O/pmc_hbm -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq "k_" > $O/pmc_sq_summary.txt
python $R/tools/pmc_summary.py $O/pmc_hbm "k_" > $O/pmc_hbm_summary.txt
cp $O/stats/r01_kernel_stats.csv $O/kernel_stats_bench.csv
rm -rf $O/stats $O/pmc_sq $O/pmc_hbm
head -8 $O/kernel_stats_bench.csv | cut -c1-140; grep -E "FETCH|WRITE" $O/pmc_hbm_summary.txt | cut -c1-140

# round 5: the per-launch fixed cost of a 1080p a-trous level on the lane kernel, by a sweep of the segment
# length (lattice rows per workgroup) in the sustained clock state; rocprofv3 kernel averages over ~7 000 launches per point.
# 540 lattice rows per (strip, y-phase) column at step 2, 8 such columns: 17 rows -> 256 workgroups (one per CU, the shipped choice),
# 26 -> 168, 34 -> 128 (one round, fewer CUs busy: T = F + rows x slope); 13 -> 336, 9 -> 480 (two rounds on part of the CUs).
# usage: exp_r05_fixed_cost.sh [WxH]     (needs libsvgf_hip_exp.so: the knob is svgf_exp_set("lane_segrows") of the experiments build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SVGF_USE_EXPERIMENTS_LIB=1
for rows in 0 9 13 17 21 26 34 45; do
  if [ $rows = 0 ]; then unset SVGF_LANE_SEGROWS; else export SVGF_LANE_SEGROWS=$rows; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fc_prof -o p --output-format csv -- python $R/tools/probe.py --size ${1:-1920x1080} --variants 4 --frames 24 --reps 200 --sustain 0.6 > /tmp/fc_probe.log 2>&1
  python - "$rows" <<PY
import csv,glob,sys,re
f=glob.glob("$R/gpurun_out/fc_prof/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_atrous_lane" in r["Name"]]
def step(n):
    m=re.search(r"k_atrous_lane<(\d+)", n); return 1<<int(m.group(1))
rows.sort(key=lambda r: step(r["Name"]))
print(f"seg_rows {sys.argv[1]:>3} (0 = automatic):", " ".join(f"step{step(r['Name'])}={float(r['AverageNs'])/1e3:.2f}us(n={r['Calls']})" for r in rows))
PY
  grep -E "frame wall|telemetry" /tmp/fc_probe.log | head -2
  rm -rf $R/gpurun_out/fc_prof
done

# A/B of an environment switch on one box: alternating runs, rocprofv3 kernel averages.  usage: exp_env_ab.sh "ENV=1" [rounds] [probe args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in $(seq 1 ${2:-3}); do for v in base "$1"; do
  if [ "$v" = base ]; then E=""; else E="$v"; fi
  env $E rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_prof -o p --output-format csv -- python $R/tools/probe.py --variants 0 --frames 24 ${3:-} > /dev/null 2>&1
  python - "$v" <<PY
import csv,glob,sys
f=glob.glob("$R/gpurun_out/ab_prof/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "atrous" in r["Name"] or "temporal" in r["Name"]]
tot=sum(float(r['AverageNs']) for r in rows if "atrous" in r["Name"])
print(sys.argv[1], " ".join(f"{r['Name'].split('::')[-1][:34]}={float(r['AverageNs'])/1e3:.2f}" for r in sorted(rows,key=lambda r:r['Name'])), f"| sum a-trous {tot/1e3:.1f}")
PY
  rm -rf $R/gpurun_out/ab_prof
done; done

# A/B/C/... of prebuilt libraries (libsvgf_hip.so.A, .B, ... next to the real one), alternating on one box; rocprofv3 kernel averages.
# usage: exp_ab_multi.sh "A B C" [kernel_variant] [rounds]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/cuda-path-tracer-denoising_amd/libsvgf_hip.so
cp $L $L.orig
for i in $(seq 1 ${3:-3}); do for v in $1; do
  cp $L.$v $L; touch $L
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_prof -o p --output-format csv -- python $R/tools/probe.py --variants ${2:-0} --frames 24 > /dev/null 2>&1
  python - "$v" <<PY
import csv,glob,sys
f=glob.glob("$R/gpurun_out/ab_prof/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "atrous" in r["Name"] or "temporal" in r["Name"]]
tot=sum(float(r['AverageNs']) for r in rows if "atrous" in r["Name"])
print(sys.argv[1], " ".join(f"{r['Name'].split('::')[-1][:22]}={float(r['AverageNs'])/1e3:.2f}" for r in sorted(rows,key=lambda r:r['Name'])), f"| sum a-trous {tot/1e3:.1f}")
PY
  rm -rf $R/gpurun_out/ab_prof
done; done
cp $L.orig $L

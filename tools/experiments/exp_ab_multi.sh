# (round 4: every run holds the sustained clock state, probe.py --sustain 0.6; the kernel averages are over ~7 000 sustained launches)
# A/B/C/... of prebuilt libraries (libsvgf_hip.so.A, .B, ... next to the real one), alternating on one box; rocprofv3 kernel averages.
# usage: exp_ab_multi.sh "A B C" [kernel_variant] [rounds]
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/ab_tm_*.jsonl
R=$GRAFT_REPO_ROOT
L=$R/cuda-path-tracer-denoising_amd/libsvgf_hip.so
cp $L $L.orig
for i in $(seq 1 ${3:-3}); do for v in $1; do
  cp $L.$v $L; touch $L
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_prof -o p --output-format csv -- python $R/tools/probe.py --variants ${2:-0} --frames 24 --reps 200 --sustain 0.6 --telemetry-json /tmp/ab_tm_$v.jsonl > /dev/null 2>&1
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
# Clock check: the runs of two libraries are comparable only when the shader clock they ran at agrees to
# 2 % (median over the wall-time loop of every run, tools/telemetry.py).  Otherwise the lines above are NOT an A/B.
python - $1 <<PY
import json, statistics, sys
sys.path.insert(0, "$R/tools")
import telemetry
tags = sys.argv[1:]
summ = {}
for t in tags:
    try:
        rows = [json.loads(l) for l in open(f"/tmp/ab_tm_{t}.jsonl")]
    except OSError:
        rows = []
    clk = [r["telemetry"]["sclk_mhz"]["median"] for r in rows if r["telemetry"].get("sclk_mhz") and r["telemetry"]["sclk_mhz"].get("median")]
    pw = [r["telemetry"]["power_w"]["median"] for r in rows if r["telemetry"].get("power_w") and r["telemetry"]["power_w"].get("median")]
    summ[t] = {"sclk_mhz": {"median": statistics.median(clk)} if clk else None, "power_w": {"median": statistics.median(pw)} if pw else None,
               "runs": len(rows), "frame_us": [round(r["frame_us"], 1) for r in rows]}
    print(f"clock {t}: median sclk {summ[t]['sclk_mhz']['median'] if clk else None} MHz, power {summ[t]['power_w']['median'] if pw else None} W over {len(rows)} runs; frame wall {summ[t]['frame_us']} us")
for i in range(len(tags)):
    for j in range(i + 1, len(tags)):
        ok = telemetry.comparable(summ[tags[i]], summ[tags[j]], tol=0.02)
        print(f"{tags[i]} vs {tags[j]}: " + ("comparable (sclk within 2 %, power within 10 %)" if ok else "REFUSED: shader clocks differ by more than 2 % or socket power by more than 10 % (or were not sampled) -- not an A/B"))
PY
rm -f /tmp/ab_tm_*.jsonl

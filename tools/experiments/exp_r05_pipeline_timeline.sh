# round 5: what the frame pipeline looks like on the GPU — rocprofv3 kernel trace of bench.py, three consecutive frames of the timed
# region as a table (kernel, hardware queue, begin / end in us): frame n's levels 2-5 and frame n+1's temporal pass + level 1 overlap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "" "--no-pipeline"; do
rm -rf /tmp/ktl
rocprofv3 --kernel-trace -d /tmp/ktl -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $mode > /dev/null 2>&1
python - "$mode" <<'PY'
import csv, glob, sys, re
f = glob.glob("/tmp/ktl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_temporal" in r["Kernel_Name"] or "k_atrous" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    m = re.search(r"k_atrous_lane<(\d+)", r["Kernel_Name"])
    return f"level step {1 << int(m.group(1)):2d}" if m else "temporal pass "
# a window in the middle of the run (sustained, pipelined or ordered as asked)
i0 = len(rows) // 2
while "k_temporal" not in rows[i0]["Kernel_Name"]:
    i0 += 1
t0 = int(rows[i0]["Start_Timestamp"])
print(f"== bench.py --steps 20 --warmup 5 {sys.argv[1] or '(frame pipeline)'}: 18 consecutive launches, times in us from the first one's begin")
qs = {}
for r in rows[i0:i0 + 18]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    b, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"  queue {q}  {name(r)}  {b:8.1f} .. {e:8.1f}  ({e - b:5.1f} us)  " + " " * int(b / 8) + "#" * max(1, int((e - b) / 8)))
PY
done

#!/bin/bash
# round 4: inter-kernel gaps inside bench.py's pipelined frames (rocprofv3 kernel trace: start / end / queue of every dispatch)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_bisect3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in _r03 .; do
  n=$(echo $t | tr -d '._'); [ -z "$n" ] && n=r04
  (cd $GRAFT_REPO_ROOT/$t && timeout 400 rocprofv3 --kernel-trace -d $O/prof_$n -o p --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2>$O/err_$n.log)
  f=$(find $O/prof_$n -name '*kernel_trace.csv' | head -1)
  python - "$f" $n >> $O/gaps.txt <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
print("==", sys.argv[2], len(rows), "dispatches; columns", list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady pipelined part: dispatches 6000..12000
sel = rows[6000:12000]
gaps = [int(sel[i + 1]["Start_Timestamp"]) - int(sel[i]["End_Timestamp"]) for i in range(len(sel) - 1)]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
qs = sorted(set(r.get("Queue_Id", "?") for r in sel))
print("queues", qs, "mean duration", statistics.mean(dur) / 1e3, "us; gap between consecutive dispatches: median", statistics.median(gaps) / 1e3, "mean", statistics.mean(gaps) / 1e3, "p90", sorted(gaps)[int(0.9 * len(gaps))] / 1e3, "us")
for r0, r1 in zip(sel[:14], sel[1:15]):
    print("  ", r0["Kernel_Name"][:50], "dur", (int(r0["End_Timestamp"]) - int(r0["Start_Timestamp"])) / 1e3, "gap to next", (int(r1["Start_Timestamp"]) - int(r0["End_Timestamp"])) / 1e3, "queue", r0.get("Queue_Id"), "lds", r0.get("LDS_Block_Size"), "scratch", r0.get("Scratch_Size"), "vgpr", r0.get("VGPR_Count"), "wg", r0.get("Workgroup_Size"), "grid", r0.get("Grid_Size"))
PY
  rm -rf $O/prof_$n
done

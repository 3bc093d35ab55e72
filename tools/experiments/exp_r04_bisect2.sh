#!/bin/bash
# round 4: rocprofv3 kernel durations inside bench.py, round-3 tree vs this tree (same box)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_bisect2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in _r03 .; do
  n=$(echo $t | tr -d '._'); [ -z "$n" ] && n=r04
  (cd $GRAFT_REPO_ROOT/$t && timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o p --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2>$O/err_$n.log)
  f=$(find $O/prof_$n -name '*kernel_stats.csv' | head -1)
  echo "== $t" >> $O/kernel_stats.txt
  python - "$f" >> $O/kernel_stats.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
  rm -rf $O/prof_$n
done

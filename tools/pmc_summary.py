"""Summarise rocprofv3 --pmc counter CSVs: median per (kernel, counter).  usage: pmc_summary.py <dir> [substr]"""
import csv, glob, sys, statistics, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "atrous"
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if sub in k:
            acc[(k[:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:70s} {c:22s} n={len(v):3d} median={statistics.median(v):14.0f}")

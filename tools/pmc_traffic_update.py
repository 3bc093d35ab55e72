#!/usr/bin/env python3
"""Rebuild profiles/pmc_traffic.json (HBM bytes per a-trous launch, what bench.py reports as roofline.traffic) from a
pmc_summary.py listing of separate FETCH_SIZE / WRITE_SIZE passes.  usage: pmc_traffic_update.py <pmc_hbm_summary.txt>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
px = 1920 * 1080
f = {}
for line in open(sys.argv[1]):
    m = re.search(r"k_atrous_(lane|strip)<(\d).*?(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ median=\s*(\d+)", line)
    if m:
        f[(int(m.group(2)), m.group(3))] = (float(m.group(4)) * 1024 / px, m.group(1))
lv, tot = {}, 0.0
for L in range(1, 6):
    fe, wr = 2 * f[(L, "FETCH_SIZE")][0], f[(L, "WRITE_SIZE")][0]        # FETCH_SIZE x2: MI355X_MICROARCH.md, gfx950 correction
    lv[f"step{1 << L}"] = {"kernel": "k_atrous_" + f[(L, "FETCH_SIZE")][1], "fetch_x2": round(fe, 1), "write": round(wr, 1)}
    tot += fe + wr
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
d["per_level_bytes_per_pixel"] = lv
d["mean_bytes_per_pixel_per_launch"] = round(tot / 5, 1)
d["mean_bytes_per_launch"] = int(tot / 5 * px)
d["_comment"] = ("HBM-side traffic of the a-trous kernels (default path: k_atrous_lane for steps 2-8, k_atrous_strip for 16-32) from rocprofv3 "
                 "PMC, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes "
                 "(profiles/r01_pmc_hbm_final.txt), KiB units, FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B). "
                 "Bytes per pixel per launch, 1920x1080.")
json.dump(d, open(p, "w"), indent=2)
print(json.dumps(lv), d["mean_bytes_per_pixel_per_launch"], d["mean_bytes_per_launch"])

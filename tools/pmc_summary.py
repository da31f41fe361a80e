#!/usr/bin/env python3
"""Condense the rocprofv3 PMC passes written by tools/pmc_run.sh into profiles/<tag>_pmc_summary.csv:
one row per counter = mean over the launches of the solve kernel (all other kernels are dropped).
usage: tools/pmc_summary.py [tag] [kernel-name substring, default mpc_solve]"""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
kernel_pat = sys.argv[2] if len(sys.argv) > 2 else "mpc_solve"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmc_{tag}")
rows = []
for d in sorted(glob.glob(os.path.join(src, "*/"))):
    name = os.path.basename(d.rstrip("/"))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch -> value
    kern = set()
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:            # gpurun_out/ keeps the files of earlier runs: only the newest pass counts
        for r in csv.DictReader(open(f)):
            if kernel_pat not in r["Kernel_Name"]:
                continue
            kern.add(r["Kernel_Name"].split("(")[0])
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c in sorted(acc):
        v = list(acc[c].values())
        rows.append((name, c, sum(v) / len(v), len(v), ";".join(sorted(kern))))
out = os.path.join(root, "profiles", f"{tag}_pmc_summary.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["pass", "counter", "mean_per_launch", "launches", "kernel"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.6g}", r[3], r[4]])
print(out, len(rows), "rows")

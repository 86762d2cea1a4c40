#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by profiles/run_profiles.sh) into profiles/<tag>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.csv (per-launch mean of every
counter for the render_frames kernel).  usage: python profiles/summarize.py <tag>"""
import csv, glob, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
rows = []
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "render_frames" not in k and "direction_checkpoints" not in k:
            continue
        for c, v in cs.items():
            rows.append((k, c, len(v), sum(v) / len(v)))
with open(os.path.join(dst, "pmc_summary.csv"), "w") as o:
    o.write("kernel,counter,launches,mean_per_launch\n")
    for r in rows:
        o.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f}\n")
        print(r)

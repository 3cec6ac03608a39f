"""Aggregate rocprofv3 --pmc CSV passes per kernel: python tools/pmc_summary.py <dir> [kernel-substring[|kernel-substring...]]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sub and not any(x in k for x in sub.split("|")):
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        n = cnt[k][c]
        print(f"   {c:28s} per-dispatch {acc[k][c] / n:16.1f}   (dispatches {n})")

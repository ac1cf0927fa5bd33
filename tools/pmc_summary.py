"""Summarise rocprofv3 --pmc counter_collection CSVs: per counter, mean/min/max over the launches of kernels matching a substring.
usage: pmc_summary.py <dir> <kernel-substring> [skip_first_n]"""
import csv, glob, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    vals = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in vals.items():
        v = v[skip:]
        if v:
            print(f"{c}: launches {len(v)} mean {sum(v) / len(v):.1f} min {min(v):.1f} max {max(v):.1f}")

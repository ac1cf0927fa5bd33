"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup) -> text table.  usage: prof_summary.py <dir> <out.txt> [header]"""
import csv, glob, re, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
header = sys.argv[3] if len(sys.argv) > 3 else ""
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void |tg::|\(.*$", "", r["Kernel_Name"])
        name = re.sub(r"\.kd$", "", name)
        key = (name, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(a[1] for a in agg.values())
with open(out, "w") as o:
    if header:
        o.write("# " + header + "\n")
    o.write(f"# total kernel time {tot / 1e3:.1f} ms; grid sizes in threads\n")
    for (name, gx, gy, wg), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{name[:44]:44s} grid({gx:8d},{gy:3d}) wg {wg:4d} calls {n:6d} avg_us {us / n:8.2f} tot_ms {us / 1e3:8.2f} {100 * us / tot:5.1f}%\n")
print(open(out).read()[:6000])

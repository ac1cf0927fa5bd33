"""One training iteration as an ordered kernel timeline, from a rocprofv3 --kernel-trace CSV.

usage: iter_timeline.py <rocprof dir> <out.txt> [marker kernel = iter_head_kernel] [iteration index from the end = 3] [min ns between markers = 1000000]

The iteration boundaries are the launches of the marker kernel that opens every iteration (the generator's RNG advance).  For the
chosen iteration prints every launch in start order: start offset, duration, gap to the previous kernel's end, name, grid; then
per-name totals and the sum of gaps (idle time between dependent launches)."""
import csv
import glob
import re
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "iter_head_kernel"
back = int(sys.argv[4]) if len(sys.argv) > 4 else 3
min_gap = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void |tg::|\(.*$", "", r["Kernel_Name"])
        name = re.sub(r"\.kd$", "", name)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]),
                     int(r["Workgroup_Size_X"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
# the generator and the discriminator each advance their RNG at the top of an iteration: keep the first of each close pair
starts = [m for j, m in enumerate(marks) if j == 0 or rows[m][0] - rows[marks[j - 1]][0] > min_gap]
assert len(starts) > back + 1, (len(marks), len(starts))
lo, hi = starts[-back - 1], starts[-back]
it = rows[lo:hi]
t0 = it[0][0]
with open(out, "w") as o:
    o.write(f"# iteration of {len(it)} launches, {(it[-1][1] - t0) / 1e3:.1f} us from first start to last end\n")
    o.write("#   start_us   dur_us   gap_us  kernel  grid(threads) wg\n")
    prev_end, gaps, busy = t0, 0.0, 0.0
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, name, gx, gy, wg in it:
        gap = (s - prev_end) / 1e3
        if gap > 0:
            gaps += gap
        busy += (e - s) / 1e3
        o.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {name[:52]:52s} ({gx},{gy}) {wg}\n")
        prev_end = max(prev_end, e)
        a = agg[name]
        a[0] += 1
        a[1] += (e - s) / 1e3
    o.write(f"# sum of kernel durations {busy:.1f} us, sum of positive gaps {gaps:.1f} us\n")
    o.write("# per kernel name: launches, total us\n")
    for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"#   {name[:52]:52s} {n:4d} {us:9.1f}\n")
print(open(out).read()[-3500:])

#!/bin/bash
# Does the next graph segment start while a gradient bucket's RCCL kernel is still running?  (bench.py --force-ddp, one rank; the unnamed
# rows of the timeline are RCCL's kernels, a NEGATIVE gap on the row after one means overlap)  usage: bash tools/ddp_overlap_probe.sh <tag> [ENV=VAL ...]
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-ddp > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --force-ddp > gpurun_out/${tag}_prof.log 2>&1
python3 tools/iter_timeline.py /tmp/prof_$tag gpurun_out/${tag}_timeline.txt > /dev/null
set -e
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print("${tag}", "$@", "ms/step", round(d["ms_per_step"], 3))
rows = [l.split(None, 3) for l in open("gpurun_out/${tag}_timeline.txt") if not l.startswith("#")]
for i, r in enumerate(rows):
    if len(r) == 4 and r[3].strip().startswith("("):          # unnamed kernel = RCCL
        nxt = rows[i + 1] if i + 1 < len(rows) else None
        print("   RCCL kernel at", r[0], "us, dur", r[1], "-> next kernel", nxt[3].split()[0] if nxt else None, "starts at", nxt[0] if nxt else None, "gap", nxt[2] if nxt else None)
PY

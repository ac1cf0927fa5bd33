#!/bin/bash
# round 5: the data-parallel code path on ONE rank (bench.py --force-ddp) against the plain single-GPU path, same box, interleaved:
# hardware queues x RCCL stream priority.   usage: bash tools/r5_ddp.sh <tag>
tag=${1:-r5_ddp}
out=gpurun_out/${tag}.txt
: > $out
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d.get('ddp',{}).get('collectives',''))"; }
# (first sweep, profiles/r5_f_ddp.txt: q8 prio0 4.87-4.88, q4 prio1 5.45, q4 prio0 4.81, q8 prio1 4.89 against plain 4.50 ms: a high-priority RCCL stream is worse)
for r in 1 2; do
  echo "round $r plain                           $(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms)" | tee -a $out
  echo "round $r ddp q4 prio0 both forks, 2 cuts $(TG_DDP_HW_QUEUES=4 python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --force-ddp 2>/dev/null | ms)" | tee -a $out
  echo "round $r ddp q4 prio0 fwd fork only      $(TG_DDP_HW_QUEUES=4 TG_DDP_BWD_FORK=0 python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --force-ddp 2>/dev/null | ms)" | tee -a $out
done

#!/bin/bash
# round 5: collectives captured inside the iteration's graph (TG_DDP_CAPTURE=1) with torch's NCCL event cache off: does the watchdog abort of
# rounds 3-4 come back, and what does the single-rank data-parallel path cost then?   usage: bash tools/r5_ddp_capture.sh <tag> [runs]
tag=${1:-r5_cap}; runs=${2:-6}
out=gpurun_out/${tag}.txt
: > $out
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d.get('ddp',{}))"; }
echo "plain $(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms)" | tee -a $out
for r in $(seq 1 $runs); do
  TG_DDP_CAPTURE=1 TG_DDP_HW_QUEUES=4 python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --force-ddp > /tmp/cap_$r.json 2> /tmp/cap_$r.err
  rc=$?
  echo "run $r rc=$rc $(cat /tmp/cap_$r.json | ms 2>/dev/null) $(grep -c -i "abort\|terminate\|not permitted" /tmp/cap_$r.err) error lines" | tee -a $out
  [ $rc -ne 0 ] && tail -5 /tmp/cap_$r.err | tee -a $out
done
echo "plain $(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms)" | tee -a $out

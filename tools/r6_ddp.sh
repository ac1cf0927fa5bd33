#!/bin/bash
# round 6: the data-parallel code path on ONE rank (bench.py --force-ddp) across its modes against the plain path, interleaved, same box:
#   bash tools/r6_ddp.sh  ->  gpurun_out/r6_ddp.txt
out=gpurun_out/r6_ddp.txt
: > $out
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), (d.get('ddp') or {}).get('collectives'), (d.get('ddp') or {}).get('rejected'))"; }
run() { label=$1; shift; r=$(env "$@" python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary $FLAGS 2>/dev/null | tail -1 | ms); echo "$label: $r" | tee -a $out; }
for round in 1 2; do
  FLAGS="" run "plain" X=1
  FLAGS="--force-ddp" run "ddp default (segments, backward fork, side rows)" X=1
  FLAGS="--force-ddp" run "ddp captured collectives" TG_DDP_CAPTURE=1
  FLAGS="--force-ddp" run "ddp segments, no side rows" TG_DDP_TN_SIDE=0
  FLAGS="--force-ddp" run "ddp segments, covered order (round 5)" TG_DDP_BWD_FORK=0
  FLAGS="--force-ddp" run "ddp captured, covered order" TG_DDP_CAPTURE=1 TG_DDP_BWD_FORK=0
done

#!/bin/bash
# Same-box A/B of one environment switch of the package (default value vs the given one), interleaved.   usage: bash tools/ab_env.sh <tag> VAR=VALUE [rounds]
tag=$1; kv=$2; rounds=${3:-3}
out=gpurun_out/${tag}_ab_env.txt
: > $out
ms() { python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for r in $(seq 1 $rounds); do
  a=$(env $kv python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  b=$(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  echo "round $r  $kv $a ms   default $b ms" | tee -a $out
done

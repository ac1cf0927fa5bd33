#!/bin/bash
# Same-box comparison of several environment settings of the package, interleaved.   usage: bash tools/ab_env2.sh <tag> <rounds> "VAR=V ..." "VAR=V ..." ...
# ("-" = the defaults)
tag=$1; rounds=$2; shift 2
out=gpurun_out/${tag}_ab_env.txt
: > $out
ms() { python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for r in $(seq 1 $rounds); do
  line="round $r"
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then v=$(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
    else v=$(env $kv python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms); fi
    line="$line | $kv $v ms"
  done
  echo "$line" | tee -a $out
done

#!/bin/bash
# Same-box A/B of the training iteration: the current library against another build of it (default: libtrimodal_hip_base.so = the previous
# round's final sources, built by hand into the package directory), interleaved runs.   usage: bash tools/ab_bench.sh <tag> [rounds] [extra bench args]
tag=${1:-ab}; rounds=${2:-3}; shift 2
pk=gesture-generation-from-trimodal-context_amd
base=${BASE_LIB:-$PWD/$pk/libtrimodal_hip_base.so}
out=gpurun_out/${tag}_ab.txt
: > $out
for r in $(seq 1 $rounds); do
  a=$(TG_LIB_PATH=$base TG_TN_MW_WS=0 python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "round $r  base $a ms   new $b ms" | tee -a $out
done

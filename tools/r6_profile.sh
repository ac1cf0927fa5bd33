#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/r6_profile.sh <tag>   -> gpurun_out/<tag>_{bench.json,by_shape.txt,timeline.txt,kernel_stats.csv}
# BENCH_ARGS: extra bench.py arguments (e.g. --force-ddp for the data-parallel code path on one rank)
set -u
tag=${1:-r6}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline ${BENCH_ARGS:-} > $out/${tag}_bench.json 2> $out/${tag}_bench.err
head -c 330 $out/${tag}_bench.json; echo
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $out/${tag}_prof.log 2>&1
python3 tools/prof_summary.py /tmp/prof_$tag $out/${tag}_by_shape.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}" > /dev/null
python3 tools/iter_timeline.py /tmp/prof_$tag $out/${tag}_timeline.txt > /dev/null
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
head -24 $out/${tag}_by_shape.txt
tail -25 $out/${tag}_timeline.txt

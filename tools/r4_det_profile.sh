#!/bin/bash
# kernel-time breakdown of the deterministic mode (bench.py --deterministic): gpurun_out/<tag>_by_shape.txt
tag=${1:-r4_det}
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --deterministic > gpurun_out/${tag}_prof.log 2>&1
python3 tools/prof_summary.py /tmp/prof_$tag gpurun_out/${tag}_by_shape.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --deterministic" > /dev/null
head -30 gpurun_out/${tag}_by_shape.txt

#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench: tools/dev/prof_bench.sh <tag> [bench args...]
# writes gpurun_out/<tag>_stats.csv (condensed per-kernel table, ms/step over the 19 steps of --steps 10 --warmup 5)
tag=$1; shift
export TMPDIR=/tmp
repo=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $repo/gpurun_out
cd $repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline "$@" > /tmp/prof_$tag.log 2>&1
grep '"metric"' /tmp/prof_$tag.log | cut -c1-260
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python tools/dev/prof_summary.py "$f" 19 gpurun_out/${tag}_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 5 $* (19 steps incl. 4 set-up steps without optimizer; eager head)"

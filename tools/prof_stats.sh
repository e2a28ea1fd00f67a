#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command (no PMC in this run; a long warm-up so that nearly all traced
# launches run at working clocks; --no-inflight: only the timed
# single-stream launches, so that the trace's average duration is that of the kernel running alone).
# usage: tools/prof_stats.sh <tag> [bench args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 200 --no-inflight "$@" > $OUT/bench.log 2>&1
echo "exit=$?" >> $OUT/bench.log
tail -2 $OUT/bench.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -r cat

#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command (no PMC in this run).
# usage: tools/prof_stats.sh <tag> [bench args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 "$@" > $OUT/bench.log 2>&1
echo "exit=$?" >> $OUT/bench.log
tail -2 $OUT/bench.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -r cat

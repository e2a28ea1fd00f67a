#!/bin/bash
# Same-call A/B of two builds of librt355 on several kbench configurations (interleaved rounds).
# usage: tools/ab.sh <out file under gpurun_out/> <lib A> <lib B> <config>[:iters] ...
OUT=gpurun_out/$1; A=$2; B=$3; shift 3
: > $OUT
for spec in "$@"; do
  cfg=${spec%%:*}; iters=${spec##*:}; [ "$iters" = "$spec" ] && iters=8
  timeout -k 10 500 python tools/kbench.py --libs $A $B --config $cfg --rounds 2 --iters $iters >> $OUT 2>&1 || exit 1
done
python tools/ab_summary.py "$OUT"

#!/bin/bash
# Same-call A/B of two builds of librt355 on several kbench configurations (interleaved rounds).
# usage: tools/ab.sh <out file under gpurun_out/> <lib A> <lib B> <config>[:iters] ...
OUT=gpurun_out/$1; A=$2; B=$3; shift 3
: > $OUT
for spec in "$@"; do
  cfg=${spec%%:*}; iters=${spec##*:}; [ "$iters" = "$spec" ] && iters=8
  timeout -k 10 500 python tools/kbench.py --libs $A $B --config $cfg --rounds 2 --iters $iters >> $OUT 2>&1 || exit 1
done
python - "$OUT" <<'PY'
import json, sys, collections
best = collections.OrderedDict()
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        k = (d["config"], d["lib"])
        best[k] = min(best.get(k, (1e30,))[0], d["min_ms"]), d["median_ms"], d["sha"], d["vgprs"]
for (cfg, lib), (mn, med, sha, vg) in best.items():
    print(f"{cfg:24s} {lib:40s} min {mn:10.4f} ms  median {med:10.4f}  vgprs {vg}  sha {sha}")
PY

#!/bin/bash
# Development sweep (needs build/dev/librt355.so): cooperative-trace threshold x pixel spread on the
# hierarchy configurations.  usage: bash tools/coop_sweep.sh "<configs>" "<shifts>" "<coop thresholds>"
CFGS=${1:-"spheres256_1080 mixed1024_1080 tori64_1080"}
SHIFTS=${2:-"6 3 0"}
COOPS=${3:-"0 8 16"}
mkdir -p gpurun_out/r2f
OUT=gpurun_out/r2f/coop_sweep.txt
for cfg in $CFGS; do
  for sh in $SHIFTS; do
    for c in $COOPS; do
      line=$(RT_SPREAD_SHIFT=$sh RT_COOP_RAYS=$c timeout -k 10 200 python tools/kbench.py --libs build/dev/librt355.so --config $cfg --rounds 1 --iters 3 2>&1 | grep '^{') || exit 1
      echo "$cfg spread_shift=$sh coop_rays=$c $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("min_ms %.3f median_ms %.3f sha %s" % (d["min_ms"], d["median_ms"], d["sha"]))')" | tee -a $OUT
    done
  done
done

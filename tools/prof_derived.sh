#!/bin/bash
# rocprofv3 derived metrics of the render kernel (one pass): VALUBusy, VALUUtilization.
# usage: tools/prof_derived.sh <lib.so> <tag> [config]
LIB=$1; TAG=$2; CFG=${3:-4k_d8s3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/derived_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc VALUBusy VALUUtilization --output-format csv -d $OUT/d1 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --child --lib $GRAFT_REPO_ROOT/$LIB --iters 6 --config $CFG > $OUT/d1.log 2>&1
echo "exit=$?"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/d1/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rt_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k, "n=%d" % len(v), "median=%.4g" % sorted(v)[len(v)//2])
PY
tail -3 $OUT/d1.log

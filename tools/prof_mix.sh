#!/bin/bash
# VALU instruction-mix counters for one kbench config.  usage: tools/prof_mix.sh <lib> <tag> [config] [extra kbench args]
LIB=$1; TAG=$2; CFG=${3:-4k_d8s3}; shift 3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mix_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU --output-format csv -d $OUT/m1 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --child --lib $GRAFT_REPO_ROOT/$LIB --iters 6 --config $CFG "$@" > $OUT/m1.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/m1/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rt_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k, "per-wave=%.1f" % (sorted(v)[len(v)//2] / 129600.0))
PY
tail -3 $OUT/m1.log

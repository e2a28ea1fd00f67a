#!/bin/bash
# PMC passes over the headline kernel (one counter group per run, no tracing domains mixed in).
# usage: tools/prof_pmc.sh <lib.so> <tag> [config]
LIB=$1; TAG=$2; CFG=${3:-4k_d8s3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() {
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --child --lib $GRAFT_REPO_ROOT/$LIB --iters 6 --config $CFG > $OUT/$1.log 2>&1 || return 1
}
run p1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" &&
run p2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" 
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rt_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(p, k, "n=%d" % len(v), "median=%.6g" % sorted(v)[len(v)//2])
PY

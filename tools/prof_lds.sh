#!/bin/bash
# LDS counters of one kbench configuration's render kernel (one --pmc pass, no tracing).
# usage: tools/prof_lds.sh <kbench config> [iters]
CFG=${1:?config}; ITERS=${2:-3}
LIB=ray-tracing_amd/librt355.so
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lds_$CFG
mkdir -p $OUT
timeout -s KILL 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --child --lib $GRAFT_REPO_ROOT/$LIB --iters $ITERS --config $CFG > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$CFG", {k: sorted(v)[len(v) // 2] for k, v in agg.items()})
PY

#!/bin/bash
# HBM traffic of the render kernel: FETCH_SIZE and WRITE_SIZE in separate counter passes
# (they do not fit one pass), per launch.  usage: tools/prof_hbm.sh <lib.so> <tag> [config]
LIB=$1; TAG=$2; CFG=${3:-4k_d8s3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/hbm_$TAG
mkdir -p $OUT
run() {
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --child --lib $GRAFT_REPO_ROOT/$LIB --iters 6 --config $CFG > $OUT/$1.log 2>&1 || return 1
}
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE" &&
python3 - <<PY
import csv, glob, collections, json
res = {}
for p in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rt_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            res[k] = {"launches": len(v), "median_kb": sorted(v)[len(v) // 2]}
# FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes
fetch = res.get("FETCH_SIZE", {}).get("median_kb", 0.0) * 1024 * 2
write = res.get("WRITE_SIZE", {}).get("median_kb", 0.0) * 1024
res["config"] = "$CFG"
res["hbm_bytes_per_launch"] = fetch + write
res["read_bytes_corrected_x2"] = fetch
res["write_bytes"] = write
print(json.dumps(res))
PY

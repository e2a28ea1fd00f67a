#!/bin/bash
# rocprofv3 counter summary of one benchmark workload's render kernel -> gpurun_out/pmc_<workload>.json
# (copy it to profiles/<round>/; bench.py reads the latest one for roofline.executed).
# One counter group per run, no tracing domains mixed in.
# usage: tools/prof_workload.sh <workload: default4k|tori64_4k|spheres256_4k|mixed1024_8k> [iters]
WL=${1:?workload}; ITERS=${2:-6}
case $WL in
  default4k) CFG=4k_d8s3;; tori64_4k) CFG=tori64_4k;; spheres256_4k) CFG=spheres256_4k;; mixed1024_8k) CFG=mixed1024_8k_16spp;;
  *) echo "unknown workload $WL"; exit 2;;
esac
LIB=ray-tracing_amd/librt355.so
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcw_$WL
mkdir -p $OUT
run() {
  timeout -s KILL 280 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --child --lib $GRAFT_REPO_ROOT/$LIB --iters $ITERS --config $CFG > $OUT/$1.log 2>&1 || return 1
}
run p1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" &&
run p2 "VALUBusy VALUUtilization" &&
run p3 "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" || exit 1
python3 - <<PY
import csv, glob, collections, json
out = {"workload": "$WL", "kbench_config": "$CFG", "launches_per_pass": $ITERS + 2}
for p in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rt_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["kernel"] = r["Kernel_Name"].split("(")[0]
        for k, v in agg.items():
            out[k] = sorted(v)[len(v) // 2]
    for line in open("$OUT/%s.log" % p):
        if line.startswith("{"):
            out["kernel_ms" if p == "p1" else "kernel_ms_" + p] = json.loads(line)["median_ms"]
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import rt355
out.update(rt355.api.library_fingerprint())     # the library these counters belong to
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_$WL.json", "w"), indent=1)
print(json.dumps(out))
PY

#!/bin/bash
# The other three benchmark workloads, in one GPU call: bench line (with roofline), rocprofv3 kernel
# stats of the same command, and the counter summary bench.py's executed-work figures come from.
#   gpurun --timeout 1200 -- 'bash tools/profile_workloads.sh <tag>'
# then copy gpurun_out/profile_<tag>/ and gpurun_out/pmc_*.json into profiles/<tag>/.
TAG=${1:?usage: tools/profile_workloads.sh <tag>}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
: > $OUT/bench_other_workloads.jsonl
for spec in "tori64_4k 40 5 6" "spheres256_4k 4 1 4" "mixed1024_8k 3 1 3"; do
  set -- $spec
  W=$1; STEPS=$2; WARM=$3; PMC_ITERS=$4
  bash tools/prof_workload.sh $W $PMC_ITERS > $OUT/prof_workload_$W.txt 2>&1 || exit 1
  cp gpurun_out/pmc_$W.json $OUT/
  mkdir -p profiles/$TAG && cp gpurun_out/pmc_$W.json profiles/$TAG/      # bench.py reads the newest summary from profiles/
  bash tools/prof_stats.sh ${TAG}_$W --workload $W --roofline --steps $STEPS --warmup $WARM > $OUT/prof_stats_$W.txt 2>&1 || exit 1
  cp "$(ls -t gpurun_out/stats_${TAG}_$W/vm/*kernel_stats.csv | head -1)" $OUT/kernel_stats_$W.csv
  grep '^{' gpurun_out/stats_${TAG}_$W/bench.log >> $OUT/bench_other_workloads.jsonl
  echo "$W done"
done
ls -la $OUT

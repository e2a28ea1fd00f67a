#!/bin/bash
# Everything profiles/<tag>/ holds for the headline workload, in one GPU call:
#   gpurun --timeout 1200 -- 'bash tools/profile_all.sh <tag>'
# then copy gpurun_out/profile_<tag>/ into profiles/<tag>/ (the box's tree does not persist).
# One profiler pass per run; counters never share a run with tracing.
TAG=${1:?usage: tools/profile_all.sh <tag>}
LIB=ray-tracing_amd/librt355.so
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
# counters first: bench.py's roofline.executed reads the newest summary under profiles/ and refuses one from other kernel sources
bash tools/prof_workload.sh default4k 6 > $OUT/prof_workload_default4k.txt 2>&1 || exit 1
mkdir -p profiles/$TAG && cp gpurun_out/pmc_default4k.json profiles/$TAG/ && cp gpurun_out/pmc_default4k.json $OUT/
timeout -k 10 300 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_20_steps.json 2>> $OUT/bench_n1.err || exit 1
bash tools/prof_stats.sh $TAG > $OUT/prof_stats.txt 2>&1 || exit 1
cp "$(ls -t gpurun_out/stats_$TAG/vm/*kernel_stats.csv | head -1)" $OUT/bench_kernel_stats.csv
bash tools/prof_hbm.sh $LIB $TAG > $OUT/prof_hbm.txt 2>&1 || exit 1
tail -1 $OUT/prof_hbm.txt > $OUT/hbm_4k_d8s3_strict.json
bash tools/prof_derived.sh $LIB $TAG > $OUT/prof_derived.txt 2>&1 || exit 1
grep "^VALU" $OUT/prof_derived.txt > $OUT/derived_4k_d8s3_strict.txt
bash tools/prof_pmc.sh $LIB $TAG > $OUT/prof_pmc.txt 2>&1 || exit 1
grep "^p[12] " $OUT/prof_pmc.txt > $OUT/pmc_4k_d8s3_strict.txt
timeout -k 10 200 python tools/part_streams.py > $OUT/part_streams_4k_d8s3.txt 2>/dev/null || exit 1
ls -la $OUT

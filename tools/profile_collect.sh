#!/bin/bash
# Copy the summaries of tools/profile_round.sh <tag> from gpurun_out/ into profiles/ (tracked).
TAG=${1:-vX}
O=gpurun_out; P=profiles
cp $O/bench_$TAG.json $P/r01_${TAG}_bench.json
cp $O/bench_${TAG}_single_stream.json $P/r01_${TAG}_bench_single_stream.json
for c in cfg1 cfg3 cfg4 cfg5; do cp $O/bench_${TAG}_$c.json $P/r01_${TAG}_bench_$c.json; done
cp "$(find $O/prof_$TAG -name '*kernel_stats.csv' | head -1)" $P/r01_${TAG}_kernel_stats.csv
cp "$(find $O/prof_${TAG}_select -name '*kernel_stats.csv' | head -1)" $P/r01_${TAG}_select_kernel_stats.csv
cp $O/traffic_$TAG.json $P/r01_${TAG}_pmc_traffic.json
cp $O/traffic_$TAG.json $P/traffic.json
if [ -f $O/sq_counters_$TAG.json ]; then cp $O/sq_counters_$TAG.json $P/r01_${TAG}_sq_counters.json; cp $O/sq_counters_$TAG.json $P/sq_counters.json; fi
ls -la $P | grep $TAG

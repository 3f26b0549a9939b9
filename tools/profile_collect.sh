#!/bin/bash
# Copy the summaries of tools/profile_round.sh <tag> from gpurun_out/ into profiles/ (tracked).
TAG=${1:-vX}
O=gpurun_out; P=profiles; R=r06
cp $O/bench_$TAG.json $P/${R}_${TAG}_bench.json
for v in steps20 steps200 single_stream 2x2 2x8 3x1 rccl_1rank cfg1 cfg3 cfg5 cfg4_shard32 cfg4_256_rccl_1rank cfg4_257_rccl_1rank; do [ -s $O/bench_${TAG}_$v.json ] && cp $O/bench_${TAG}_$v.json $P/${R}_${TAG}_bench_$v.json; done
cp "$(find $O/prof_$TAG -name '*kernel_stats.csv' | head -1)" $P/${R}_${TAG}_kernel_stats.csv
for v in select cfg3 cfg4 cfg5; do f="$(find $O/prof_${TAG}_$v -name '*kernel_stats.csv' | head -1)"; [ -n "$f" ] && cp "$f" $P/${R}_${TAG}_${v}_kernel_stats.csv; done
cp $O/traffic_$TAG.json $P/${R}_${TAG}_pmc_traffic.json
cp $O/traffic_$TAG.json $P/traffic.json
if [ -f $O/sq_counters_$TAG.json ]; then cp $O/sq_counters_$TAG.json $P/${R}_${TAG}_sq_counters.json; cp $O/sq_counters_$TAG.json $P/sq_counters.json; fi
[ -s $O/api_probe_$TAG.json ] && cp $O/api_probe_$TAG.json $P/${R}_${TAG}_api_probe.json
[ -s $O/api_profile_$TAG.txt ] && cp $O/api_profile_$TAG.txt $P/${R}_${TAG}_api_profile.txt
[ -s $O/api_timeline_$TAG.txt ] && cp $O/api_timeline_$TAG.txt $P/${R}_${TAG}_api_timeline.txt
[ -s $O/traffic_${TAG}_cfg4.json ] && cp $O/traffic_${TAG}_cfg4.json $P/${R}_${TAG}_cfg4_pmc_traffic.json
[ -s $O/mem_counters_$TAG.json ] && [ $O/mem_counters_$TAG.json -nt $O/bench_$TAG.json ] && cp $O/mem_counters_$TAG.json $P/${R}_${TAG}_mem_counters.json
for c in cfg3 cfg5; do [ -s $O/sq_counters_${TAG}_$c.json ] && cp $O/sq_counters_${TAG}_$c.json $P/${R}_${TAG}_${c}_sq_counters.json; done
ls -la $P | grep ${R}_$TAG

#!/bin/bash
# One profiling pass on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh <tag>
# Writes under gpurun_out/: the default bench line, the other BASELINE configs, rocprofv3 kernel stats of the cfg-2 step
# (one stream, so that kernel times are not stretched by a co-running pair) and of the selection, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as gpurun requires) that tools/pmc_traffic.py turns into HBM bytes per launch.
# tools/profile_collect.sh <tag> then copies the summaries into profiles/.
TAG=${1:-vX}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python3 bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
python3 bench.py --inflight 1 --no-cpu-baseline > $O/bench_${TAG}_single_stream.json 2>> $O/bench_$TAG.err
for c in cfg1 cfg3 cfg4 cfg5; do python3 bench.py --config $c > $O/bench_${TAG}_$c.json 2>> $O/bench_$TAG.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python3 bench.py --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_select -o run -- python3 tools/profile_select.py 50 > $O/prof_${TAG}_select.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_fetch -o run -- python3 bench.py --inflight 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_write -o run -- python3 bench.py --inflight 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/pmc_${TAG}_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write $O/traffic_$TAG.json > /dev/null 2>> $O/bench_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_${TAG}_sq1 -o run -- python3 bench.py --inflight 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/pmc_${TAG}_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_${TAG}_sq2 -o run -- python3 bench.py --inflight 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/pmc_${TAG}_sq2.log 2>&1
python3 tools/pmc_sq.py $O/sq_counters_$TAG.json $O/pmc_${TAG}_sq1 $O/pmc_${TAG}_sq2 > /dev/null 2>> $O/bench_$TAG.err
find $O/prof_$TAG $O/prof_${TAG}_select -name "*kernel_stats.csv" | head
tail -c 600 $O/bench_$TAG.json

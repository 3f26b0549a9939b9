#!/bin/bash
# One profiling pass on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh <tag>
# Writes under gpurun_out/: the default bench line, the other BASELINE configs, rocprofv3 kernel stats of the cfg-2 step
# (one stream, so that kernel times are not stretched by a co-running pair), of the selection and of cfg-3 / cfg-4 / cfg-5, the
# two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as gpurun requires) that tools/pmc_traffic.py turns into HBM bytes
# per launch, and the SQ counter passes (tools/pmc_sq.py).  The counter passes run FIRST and their summaries go into profiles/ on the box,
# so that the bench lines written afterwards quote counters of the tree they ran on.  Every rocprofv3 call sits under a timeout: a
# refused counter set makes rocprofv3 abort and then hang.  tools/profile_collect.sh <tag> then copies the summaries into profiles/.
TAG=${1:-vX}
export TMPDIR=/tmp
export KLT_PROFILE_TAG=r06_$TAG
export KLT_PROFILE_BATCH=8       # pairs per launch of the cfg-2 passes below (bench.py --batch default)
O=gpurun_out
mkdir -p $O
PROF="--min-timed-s 0 --repeats 5 --no-cpu-baseline --no-extras"
PMC="--inflight 1 --resident-pairs 8 --steps 3 --warmup 1 $PROF"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_fetch -o run -- python3 bench.py $PMC > $O/pmc_${TAG}_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_write -o run -- python3 bench.py $PMC > $O/pmc_${TAG}_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write $O/traffic_$TAG.json > /dev/null 2>> $O/bench_$TAG.err
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${TAG}_sq$i -o run -- python3 bench.py $PMC > $O/pmc_${TAG}_sq$i.log 2>&1
done
python3 tools/pmc_sq.py $O/sq_counters_$TAG.json $O/pmc_${TAG}_sq1 $O/pmc_${TAG}_sq2 $O/pmc_${TAG}_sq3 $O/pmc_${TAG}_sq4 $O/pmc_${TAG}_sq5 $O/pmc_${TAG}_sq6 > /dev/null 2>> $O/bench_$TAG.err
# the counters just collected describe this tree: the bench lines below quote them (bench.py reads profiles/traffic.json and
# profiles/sq_counters.json and drops them when the kernel sources they were collected for are not the ones in the tree)
cp $O/traffic_$TAG.json profiles/traffic.json
[ -s $O/sq_counters_$TAG.json ] && cp $O/sq_counters_$TAG.json profiles/sq_counters.json
python3 bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
python3 bench.py --steps 20 --warmup 5 --no-config-sweep > $O/bench_${TAG}_steps20.json 2>> $O/bench_$TAG.err
python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_${TAG}_steps200.json 2>> $O/bench_$TAG.err
python3 bench.py --inflight 1 --batch 1 --no-cpu-baseline --no-api --no-config-sweep > $O/bench_${TAG}_single_stream.json 2>> $O/bench_$TAG.err
python3 bench.py --inflight 2 --batch 2 --no-cpu-baseline --no-api --no-config-sweep > $O/bench_${TAG}_2x2.json 2>> $O/bench_$TAG.err
python3 bench.py --inflight 2 --batch 8 --resident-pairs 64 --no-cpu-baseline --no-api --no-config-sweep > $O/bench_${TAG}_2x8.json 2>> $O/bench_$TAG.err
python3 bench.py --inflight 3 --batch 1 --resident-pairs 66 --no-cpu-baseline --no-api --no-config-sweep > $O/bench_${TAG}_3x1.json 2>> $O/bench_$TAG.err
KLT_FORCE_DIST=1 python3 bench.py --gpus 1 --no-cpu-baseline --no-api --no-config-sweep > $O/bench_${TAG}_rccl_1rank.json 2>> $O/bench_$TAG.err
for c in cfg1 cfg3 cfg5; do python3 bench.py --config $c > $O/bench_${TAG}_$c.json 2>> $O/bench_$TAG.err; done
python3 bench.py --config cfg4 --pairs 32 --steps 50 --warmup 5 > $O/bench_${TAG}_cfg4_shard32.json 2>> $O/bench_$TAG.err
KLT_FORCE_DIST=1 python3 bench.py --config cfg4 --gpus 1 --pairs 256 --steps 10 --warmup 2 > $O/bench_${TAG}_cfg4_256_rccl_1rank.json 2>> $O/bench_$TAG.err
KLT_FORCE_DIST=1 python3 bench.py --config cfg4 --gpus 1 --pairs 257 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_${TAG}_cfg4_257_rccl_1rank.json 2>> $O/bench_$TAG.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python3 bench.py --inflight 1 --steps 6 --warmup 2 $PROF > $O/prof_$TAG.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_select -o run -- python3 tools/profile_select.py 50 > $O/prof_${TAG}_select.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_cfg3 -o run -- python3 bench.py --config cfg3 --steps 100 --warmup 10 --min-timed-s 0 --repeats 5 --no-cpu-baseline > $O/prof_${TAG}_cfg3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_cfg4 -o run -- python3 bench.py --config cfg4 --pairs 32 --steps 30 --warmup 3 --min-timed-s 0 --repeats 5 --no-cpu-baseline > $O/prof_${TAG}_cfg4.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_cfg5 -o run -- python3 bench.py --config cfg5 --frames 32 --steps 31 --min-timed-s 0 --repeats 5 --no-cpu-baseline > $O/prof_${TAG}_cfg5.log 2>&1
python3 tools/api_probe.py --4k > $O/api_probe_$TAG.json 2>> $O/bench_$TAG.err
python3 tools/api_profile.py > $O/api_profile_$TAG.txt 2>> $O/bench_$TAG.err
python3 tools/api_timeline.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/api_timeline_$TAG.txt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_cfg4_fetch -o run -- python3 bench.py --config cfg4 --pairs 32 --steps 10 --warmup 2 --min-timed-s 0 --repeats 3 --no-cpu-baseline > $O/pmc_${TAG}_cfg4_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_cfg4_write -o run -- python3 bench.py --config cfg4 --pairs 32 --steps 10 --warmup 2 --min-timed-s 0 --repeats 3 --no-cpu-baseline > $O/pmc_${TAG}_cfg4_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_${TAG}_cfg4_fetch $O/pmc_${TAG}_cfg4_write $O/traffic_${TAG}_cfg4.json > /dev/null 2>> $O/bench_$TAG.err
# (memory-path counters of the tracker: bash tools/pmc_mem.sh $TAG -- eleven more passes, run separately when the tracker changes)
find $O/prof_$TAG $O/prof_${TAG}_select -name "*kernel_stats.csv" | head
tail -c 600 $O/bench_$TAG.json

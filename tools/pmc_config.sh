#!/bin/bash
# SQ counter passes (kernel-trace only, as gpurun requires) of one of the other BASELINE configs:  tools/pmc_config.sh <tag> cfg3|cfg5
# -> gpurun_out/sq_counters_<tag>_<cfg>.json (tools/pmc_sq.py: counters per launch of every kernel family, largest launch)
TAG=${1:-vX}; CFG=${2:-cfg3}
O=gpurun_out
ST=20; [ $CFG = cfg5 ] && ST="14 --frames 8"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${TAG}_${CFG}_sq$i -o run -- python3 bench.py --config $CFG --steps $ST --warmup 2 --min-timed-s 0 --repeats 5 --no-cpu-baseline > $O/pmc_${TAG}_${CFG}_sq$i.log 2>&1
done
python3 tools/pmc_sq.py $O/sq_counters_${TAG}_$CFG.json $O/pmc_${TAG}_${CFG}_sq1 $O/pmc_${TAG}_${CFG}_sq2 $O/pmc_${TAG}_${CFG}_sq3 $O/pmc_${TAG}_${CFG}_sq4 $O/pmc_${TAG}_${CFG}_sq5 > /dev/null
ls -la $O/sq_counters_${TAG}_$CFG.json

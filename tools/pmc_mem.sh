#!/bin/bash
# Memory-path counters of the cfg-2 kernels (tracker first of all):  tools/pmc_mem.sh <tag>   (through gpurun, from the repo root)
# Separate rocprofv3 --pmc passes (kernel-trace only; two counters of a block per pass -- four TA counters at once are refused and the
# refused rocprofv3 then hangs, hence the timeout around every pass), reduced per launch by tools/pmc_sq.py into gpurun_out/mem_counters_<tag>.json:
# texture-addresser / L1 (TCP) / L2 (TCC) activity, the L1's request latency towards the L2 and the address-translation misses.
TAG=${1:-vX}
export TMPDIR=/tmp
export KLT_PROFILE_TAG=r04_$TAG
export KLT_PROFILE_BATCH=8
O=gpurun_out
mkdir -p $O
PMC="--inflight 1 --resident-pairs 8 --steps 3 --warmup 1 --min-timed-s 0 --repeats 5 --no-cpu-baseline --no-extras"
i=0
dirs=""
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${TAG}_mem$i -o run -- python3 bench.py $PMC > $O/pmc_${TAG}_mem$i.log 2>&1
  dirs="$dirs $O/pmc_${TAG}_mem$i"
done
python3 tools/pmc_sq.py $O/mem_counters_$TAG.json $dirs > /dev/null 2> $O/pmc_${TAG}_mem.err
python3 - <<PY
import json
d = json.load(open("$O/mem_counters_$TAG.json"))
for fam in ("track", "smooth_grad_l0", "gradients", "pyramid_reduce"):
    print(fam, json.dumps(d.get(fam)))
PY

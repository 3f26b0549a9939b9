#!/bin/bash
# A/B of two builds of libkltgpu.so on one box, alternating:  tools/ab_libs.sh <other.so> [bench.py arguments]
# (the tree's library against <other.so> through KLT_GPU_LIB; prints ms per step, features/s and the kernel durations of each run)
OTHER=$1; shift
O=gpurun_out
mkdir -p $O
for rep in 1 2; do
  for which in tree other; do
    if [ $which = other ]; then export KLT_GPU_LIB=$OTHER; else unset KLT_GPU_LIB; fi
    timeout 400 python3 bench.py --no-extras --no-cpu-baseline --no-api "$@" > $O/ab_${which}_$rep.json 2> $O/ab_${which}_$rep.err
    python3 - <<PY
import json
d = json.loads(open("$O/ab_${which}_$rep.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]
print("$which", $rep, "ms_per_step", round(d["ms_per_step"], 5), "M feat/s", round(d["value"] / 1e6, 2), {n: round(v["us_per_launch"], 2) for n, v in k.items()}, "parity", d.get("parity_checked"))
PY
  done
done
unset KLT_GPU_LIB

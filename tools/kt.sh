#!/bin/bash
# per-kernel event timings of the cfg-2 step (compact); usage: tools/kt.sh [bench args]
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline "$@" 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    j = json.loads(line)
    r = j["roofline"]
    print("ms/step %.4f  value %.1fM  frac %.3f  select_ms %.4f  single-stream ms %.4f  latency ms %.4f" % (j["ms_per_step"], j["value"]/1e6, r["frac"], j["extra"]["ms_per_select_5000"], j["extra"]["single_stream_ms_per_pair"], j["extra"]["latency_ms_per_pair_synchronised"]))
    for k, v in r["kernels"].items(): print("   %-16s %6.2f us x %.0f" % (k, v["us_per_launch"], v["launches_per_step"]))
'

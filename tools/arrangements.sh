#!/bin/bash
# cfg-2 headline under other arrangements of contexts x pairs per launch (same resident pairs, same work per pair):  tools/arrangements.sh
for arr in "--inflight 2 --batch 8 --resident-pairs 64" "--inflight 3 --batch 8 --resident-pairs 72" "--inflight 2 --batch 16 --resident-pairs 64" "--inflight 4 --batch 4 --resident-pairs 64"; do
  timeout 400 python3 bench.py --no-extras --no-cpu-baseline --no-api $arr 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$arr', 'ms_per_pair', round(d['config']['ms_per_pair'], 5), 'M feat/s', round(d['value'] / 1e6, 2), 'parity', d.get('parity_checked'))"
done

#!/bin/bash
# cfg-5 at a glance on the GPU box: ms per frame (twice) and the average duration of every kernel of the sequence (rocprofv3 --stats).
#   tools/cfg5_kernels.sh [frames=256]
F=${1:-256}
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --config cfg5 --frames $F --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg5 ms/frame', round(d['ms_per_step'],4), d['parity_checked'])"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5k -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --frames 64 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c5k.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/c5k/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"].replace("(anonymous namespace)::","")[:56].ljust(56), r["Calls"].rjust(7), r["AverageNs"][:8].rjust(9), r["Percentage"])
PY
rm -rf gpurun_out/c5k

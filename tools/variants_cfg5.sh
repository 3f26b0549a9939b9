#!/bin/bash
# A/B on one GPU box: cfg-5 with the fused column pass + eigenvalue kernel off / on (KLT_FUSED_COLS_EIGEN), alternating; then the fused
# kernel's average duration inside the sequence (rocprofv3 --stats).      tools/variants_cfg5.sh [frames=256]
F=${1:-256}
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 0 1; do
  KLT_FUSED_COLS_EIGEN=$v python bench.py --config cfg5 --frames $F --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fused=$v rep $rep cfg5 ms/frame', round(d['ms_per_step'],4), d['parity_checked'])"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5k -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --frames 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/c5k/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "cols_eigen" in r["Name"] or "sat_rows" in r["Name"]: print(r["Name"][23:45], r["AverageNs"][:8])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/c5k

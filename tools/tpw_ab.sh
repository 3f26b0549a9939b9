cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 1 4; do
  KLT_MIS_TPW=$v python bench.py --config cfg5 --frames 256 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('tpw=$v rep $rep cfg5 ms/frame', round(d['ms_per_step'],4), d['parity_checked'])"
done; done
for v in 1 4 1 4; do KLT_MIS_TPW=$v python bench.py --no-sequences --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['extra']; print('tpw=$v', round(d['value']/1e6,2), 'ms_per_select_5000', round(e['ms_per_select_5000'],4))"; done
for m in "" --sequence --batch; do timeout 300 python tests/fuzz/fuzz_parity.py --trials 400 --seed 91 $m 2>&1 | tail -1; done

cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/${KLT_FUZZ_OUT:-r06_fuzz_long.txt}
: > $O
t() { local name="$1"; shift; local s=$(date +%s); r=$(timeout 900 python tests/fuzz/fuzz_parity.py "$@" 2>&1 | tail -1); echo "fuzz $name: $r" | tee -a $O; }
t "default (seed 31337)" --trials 6000 --seed 31337
t "sequence (seed 31337)" --trials 8000 --seed 31337 --sequence
t "sequence, frames up to 3 Mpx (seed 31338)" --trials 300 --seed 31338 --sequence --max-pixels 3000000 --max-n 6000 --max-side 2200
t "batch (seed 31337)" --trials 2000 --seed 31337 --batch
t "huge (seed 4243, frames up to 8.5 Mpx)" --trials 80 --seed 4243 --max-pixels 8500000 --max-n 20000 --max-side 3900
t "prepared replacement vs oracle, frames of 0.3-0.7 Mpx (seed 777)" --trials 250 --seed 777 --prepared --min-pixels 300000 --max-pixels 700000 --max-n 3000 --max-side 1100
t "Python API over random call sequences vs oracle (seed 8088)" --trials 3000 --seed 8088 --api
t "Python API on Pillow images over random call sequences vs oracle (seed 8089)" --trials 3000 --seed 8089 --api --pil

# More seeds on the final tree of round 5 (run through gpurun from the repo root): bash tests/fuzz/fuzz_seeds_r05.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r05_fuzz_seeds.txt
: > $O
t() { local name="$1"; shift; r=$(timeout 1200 python tests/fuzz/fuzz_parity.py "$@" 2>&1 | tail -1); echo "fuzz $name: $r" | tee -a $O; }
t "default (seed 50501)" --trials 15000 --seed 50501
t "sequence (seed 50502)" --trials 15000 --seed 50502 --sequence
t "affine (seed 50503)" --trials 4000 --seed 50503 --affine
t "batch (seed 50504)" --trials 4000 --seed 50504 --batch
t "prepared replacement vs oracle (seed 50505)" --trials 1500 --seed 50505 --prepared --min-pixels 300000 --max-pixels 900000 --max-n 4000 --max-side 1300

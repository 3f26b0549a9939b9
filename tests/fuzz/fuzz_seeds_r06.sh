# More seeds on the final tree of round 6 (run through gpurun from the repo root): bash tests/fuzz/fuzz_seeds_r06.sh [base seed] [tag]
# As fuzz_seeds_r05.sh, plus the Python API on Pillow images (the reference's image type, read through Pillow's row tables) and sequences
# through KLTTrackSequence's two helper threads (frames of 4 MB and more: --max-pixels 5000000).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
B=${1:-60600}
O=gpurun_out/r06_fuzz_seeds${2:+_$2}.txt
: > $O
t() { local name="$1"; shift; r=$(timeout 1500 python tests/fuzz/fuzz_parity.py "$@" 2>&1 | tail -1); echo "fuzz $name: $r" | tee -a $O; }
t "default (seed $((B+1)))" --trials ${FUZZ_DEFAULT:-15000} --seed $((B+1))
t "sequence (seed $((B+2)))" --trials ${FUZZ_SEQUENCE:-15000} --seed $((B+2)) --sequence
t "sequence, frames of 4-5 Mpx: two staging threads (seed $((B+7)))" --trials ${FUZZ_SEQUENCE_BIG:-120} --seed $((B+7)) --sequence --min-pixels 4200000 --max-pixels 5000000 --max-n 6000 --max-side 2800
t "affine (seed $((B+3)))" --trials ${FUZZ_AFFINE:-4000} --seed $((B+3)) --affine
t "batch (seed $((B+4)))" --trials ${FUZZ_BATCH:-4000} --seed $((B+4)) --batch
t "prepared replacement vs oracle (seed $((B+5)))" --trials ${FUZZ_PREPARED:-1500} --seed $((B+5)) --prepared --min-pixels 300000 --max-pixels 900000 --max-n 4000 --max-side 1300
t "random call sequences through the Python API (seed $((B+6)))" --trials ${FUZZ_API:-6000} --seed $((B+6)) --api
t "random call sequences through the Python API on Pillow images (seed $((B+8)))" --trials ${FUZZ_API_PIL:-6000} --seed $((B+8)) --api --pil
true

#!/bin/bash
# The randomised HIP-vs-oracle checks on the tree's kernels, ~10 minutes on one GPU:  tests/fuzz/fuzz_round.sh [seed]
S=${1:-301}
O=gpurun_out
mkdir -p $O
for mode in "" "--sequence" "--affine" "--batch"; do
  timeout 600 python3 tests/fuzz/fuzz_parity.py --trials 500 --seed $S $mode > $O/fuzz_${S}_${mode#--}.log 2>&1; echo "fuzz $mode rc=$? $(tail -1 $O/fuzz_${S}_${mode#--}.log)"
done
timeout 600 python3 tests/fuzz/fuzz_parity.py --trials 60 --seed $((S + 1)) --max-pixels 3000000 --max-n 6000 --max-side 2200 > $O/fuzz_${S}_large.log 2>&1; echo "fuzz large rc=$? $(tail -1 $O/fuzz_${S}_large.log)"
timeout 600 python3 tests/fuzz/fuzz_parity.py --trials 60 --seed $((S + 2)) --max-pixels 3000000 --max-n 6000 --max-side 2200 --batch > $O/fuzz_${S}_large_batch.log 2>&1; echo "fuzz large batch rc=$? $(tail -1 $O/fuzz_${S}_large_batch.log)"

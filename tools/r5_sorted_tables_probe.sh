#!/bin/bash
# round 5: tables by sorted position (symmetrize_pairs=2) against tables by row (=1): bit equality + stage times on C3, then the tests of the tail
cd "$(dirname "$0")/.."
O=gpurun_out/r5_sorted_tables
mkdir -p $O
export TMPDIR=/tmp
for kind in mix manifold; do
  GT_REPS=5 GT_VARIANTS="symmetrize_pairs=1;symmetrize_pairs=2" timeout 900 python tools/gpu_ab_probe.py 1000000 64 $kind > $O/ab_$kind.txt 2>&1
  tail -4 $O/ab_$kind.txt
done
timeout 1500 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_symm_bins.py tests/test_gpu_full_reference.py -m gpu -x -q > $O/tests.txt 2>&1
tail -5 $O/tests.txt

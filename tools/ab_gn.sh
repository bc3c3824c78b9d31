#!/bin/bash
# same-box A/B of the aligner kernels: _abtmp/base (a built copy of a reference commit) against the working tree
# usage (GPU box, repo root): bash tools/ab_gn.sh [batch] [runs]
B=${1:-55296}
N=${2:-2}
for i in $(seq $N); do
  echo "base:"; (cd _abtmp/base && python tools/bench_gn.py $B 2>&1 | tail -2)
  echo "new:";  python tools/bench_gn.py $B 2>&1 | tail -2
done

#!/bin/bash
# parity evidence of a round on the GPU box: randomised sweeps against the CPU checker + closed loops along KITTI 00
# usage: bash tools/evidence_round.sh r03   (writes gpurun_out/evidence_r03/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd $R
{
  echo "Randomised parity sweeps, GPU path vs CPU checker (MI355X, $TAG)"
  echo "tools/fuzz_align.py 600 3131; tools/fuzz_matcher.py 500 31; tools/fuzz_merge.py 200 313; tools/fuzz_rows.py 200 3"
  timeout 1500 python tools/fuzz_align.py 600 3131 2>&1 | tail -3
  timeout 900 python tools/fuzz_matcher.py 500 31 2>&1 | tail -2
  timeout 900 python tools/fuzz_merge.py 200 313 2>&1 | tail -2
  timeout 900 python tools/fuzz_rows.py 200 3 2>&1 | tail -4
} > $OUT/fuzz_sweeps.txt 2>&1
timeout 900 python tools/bench_tracking.py --batch 256 --frames 240 --check 2 > $OUT/closed_loop_kitti00_240frames.json 2> $OUT/cl240.err
timeout 900 python tools/bench_tracking.py --batch 4096 --frames 60 --check 1 > $OUT/closed_loop_kitti00_4096x60.json 2> $OUT/cl4096.err
timeout 2400 python tools/bench_tracking.py --batch 64 --frames 4541 --keypoints 1000 --check 2 > $OUT/closed_loop_kitti00_full_4541frames.json 2> $OUT/clfull.err
tail -n 2 $OUT/cl240.err $OUT/cl4096.err $OUT/clfull.err
cat $OUT/fuzz_sweeps.txt

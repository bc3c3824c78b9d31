#!/bin/bash
# tools/latency_b1 on a fresh frames file, once per environment given as arguments ("-" = none), e.g.
#   bash tools/latency_quick.sh - PRS_FUSED_ALIGN=1 "PRS_FUSED_ALIGN=1 PRS_STAMPS=1"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lat_quick
mkdir -p $OUT
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import bench
from srrg2_proslam_amd import configs, synthetic as syn
cfg = configs.get("kitti")
frames = bench.make_unique_frames(cfg, 16, 2000, 2000, syn.seed_for(1, 0))
bench.write_latency_frames("$OUT/frames.bin", bench.latency_params(cfg), [frames[k % 16] for k in range(32)])
PY
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo "== env: $e"
  env $e $R/tools/bin/latency_b1 $OUT/frames.bin 4 2>&1 | tail -${LAT_TAIL:-3}
  cp $OUT/frames.bin.poses "$OUT/poses_$(echo $e | tr ' =' '__').bin" 2>/dev/null
done
rm -f $OUT/frames.bin

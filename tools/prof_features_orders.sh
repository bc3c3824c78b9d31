#!/bin/bash
# rocprofv3 kernel statistics of the feature extractor on the real KITTI frames, both selection orders
# usage on the GPU box: bash tools/prof_features_orders.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ORDER in canonical libstdcxx; do
  OUT=$R/gpurun_out/prof_features_$ORDER
  rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_features.py 4096 kitti $ORDER > $OUT/run.log 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-70s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  tail -1 $OUT/run.log
done

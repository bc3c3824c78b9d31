#!/bin/bash
# rocprofv3 kernel statistics of the closed loop (tools/bench_tracking.py, 4096 sequences x 30 frames); usage on the GPU box: bash tools/prof_tracking.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_tracking
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_tracking.py --batch 4096 --frames 30 --check 0 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-90s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -2 $OUT/run.log | cut -c1-600

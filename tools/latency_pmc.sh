#!/bin/bash
# PMC counters of the one-frame-at-a-time program (tools/latency_b1.cpp); usage: bash tools/latency_pmc.sh "COUNTER1 COUNTER2" [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lat_pmc_${2:-x}
rm -rf $OUT; mkdir -p $OUT
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import bench
from srrg2_proslam_amd import configs, synthetic as syn
cfg = configs.get("kitti")
frames = bench.make_unique_frames(cfg, 16, 2000, 2000, syn.seed_for(1, 0))
bench.write_latency_frames("$OUT/frames.bin", bench.latency_params(cfg), [frames[k % 16] for k in range(8)])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $1 --output-format csv -d $OUT/pmc -- $R/tools/bin/latency_b1 $OUT/frames.bin 1 > $OUT/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
tail -3 $OUT/pmc.log
rm -f $OUT/frames.bin*

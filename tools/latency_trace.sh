#!/bin/bash
# kernel trace of the one-frame-at-a-time program (tools/latency_b1.cpp): which launches a frame costs and how long each takes
# usage (on the GPU box, from the repo root): bash tools/latency_trace.sh r05
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/lat_$TAG
rm -rf $OUT; mkdir -p $OUT
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import bench
from srrg2_proslam_amd import configs, synthetic as syn
cfg = configs.get("kitti")
frames = bench.make_unique_frames(cfg, 16, 2000, 2000, syn.seed_for(1, 0))
bench.write_latency_frames("$OUT/frames.bin", bench.latency_params(cfg), [frames[k % 16] for k in range(32)])
PY
cd /tmp && export TMPDIR=/tmp
$R/tools/bin/latency_b1 $OUT/frames.bin 4 > $OUT/plain.json 2> $OUT/plain.err
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- $R/tools/bin/latency_b1 $OUT/frames.bin 4 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))))
    # the LAST frame of the C-ABI flavour: everything after the last stereo matcher launch
    last = max(i for i, r in enumerate(rows) if "stereo_match" in r[2])
    t0 = rows[last][0]
    lines = ["%8.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, k[:110]) for s, e, k in rows[last:]]
    open(out + "/last_frame_timeline.txt", "w").write("\n".join(lines) + "\n")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, k in rows:
        agg[k[:110]][0] += 1
        agg[k[:110]][1] += (e - s) / 1e3
    open(out + "/kernel_totals.txt", "w").write("\n".join("%6d launches %10.1f us total %8.2f us mean  %s" % (n, t, t / n, k) for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])) + "\n")
print(open(out + "/plain.json").read())
print(open(out + "/last_frame_timeline.txt").read())
print(open(out + "/kernel_totals.txt").read())
PY
rm -f $OUT/frames.bin $OUT/frames.bin.poses

#!/bin/bash
# collects the rocprofv3 evidence for a round: kernel-trace stats of bench.py + PMC passes (own runs)
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
python3 - <<PY
import csv, glob, collections, json, os
out = "$OUT"
summary = {}
for f in glob.glob(out + "/stats/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [r for r in rows if "prs::" in r["Name"]]
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(out + "/" + name + "/*/*_counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "prs::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summary[name] = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
summary["frames_per_launch"] = int(os.environ.get("PRS_PROFILE_BATCH", "18432"))
summary["keypoints_per_image"] = int(os.environ.get("PRS_PROFILE_KEYPOINTS", "2000"))
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
PY

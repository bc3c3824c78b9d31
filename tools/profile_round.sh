#!/bin/bash
# collects the rocprofv3 evidence for a round: kernel trace of bench.py's HEADLINE workload alone (no steady-state leg, no other
# configurations, no CPU legs: every step the profiler sees is the step the bench times) + PMC passes (own runs)
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r03
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-steady-state --no-other-configs"
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
# per-launch durations: a step = one matcher launch followed by the rounds of (search, Gauss-Newton) launches of the aligner
for f in glob.glob(out + "/stats/*/*_kernel_trace.csv"):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    steps, cur = [], None
    for s, e, k in rows:
        if "stereo_match5_kernel" in k:
            cur = {"matcher": (e - s) / 1e6, "search": [], "gn": []}
            steps.append(cur)
        elif cur is not None and "prs::align_kernel" in k:
            cur["search"].append((e - s) / 1e6)
        elif cur is not None and "prs::gn_kernel" in k:
            cur["gn"].append((e - s) / 1e6)
    if steps:
        rounds = max(len(st["search"]) for st in steps)
        mean = lambda xs: sum(xs) / max(len(xs), 1)
        summary["per_launch_ms"] = {
            "steps_profiled": len(steps),
            "matcher": mean([st["matcher"] for st in steps]),
            "search_by_round": [mean([st["search"][r] for st in steps if len(st["search"]) > r]) for r in range(rounds)],
            "gn_by_round": [mean([st["gn"][r] for st in steps if len(st["gn"]) > r]) for r in range(rounds)],
            "search_per_step": mean([sum(st["search"]) for st in steps]),
            "gn_per_step": mean([sum(st["gn"]) for st in steps]),
        }
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(out + "/" + name + "/*/*_counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "prs::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summary[name] = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
summary["frames_per_launch"] = int(os.environ.get("PRS_PROFILE_BATCH", "55296"))
summary["keypoints_per_image"] = int(os.environ.get("PRS_PROFILE_KEYPOINTS", "2000"))
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary.get("per_launch_ms", {}), indent=1))
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_feat -- python3 $R/tools/bench_features.py 2048 > $R/gpurun_out/prof_feat.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/prof_feat/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "prs::" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY

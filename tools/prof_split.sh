cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_split -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_split.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/prof_split/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "prs::" in r["Name"]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for f in glob.glob("$R/gpurun_out/prof_split/*/*_kernel_trace.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "prs::" in r["Kernel_Name"]]
    rows=rows[-16:]
    for r in rows: print(r["Kernel_Name"][:40], int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_stereo -- python3 $R/tools/bench_stereo.py 2048 2000 > $R/gpurun_out/pmc_stereo.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$R/gpurun_out/pmc_stereo/*/*_counter_collection.csv"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "stereo" in r["Kernel_Name"]: acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in acc.items():
        print(k, {c: round(sum(v)/len(v)/2048) for c,v in d.items()})
PY

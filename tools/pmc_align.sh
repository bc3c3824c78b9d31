#!/bin/bash
# PMC passes (<= 3 counters each) over the bench's aligner kernels; usage on the GPU box: bash tools/pmc_align.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_align
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --batch 6144 --no-cpu-baseline --no-steady-state --no-other-configs --timing-steps 1 > $OUT/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "align_kernel" in k or "gn_kernel" in k: acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-24s %16.0f total over %d launches, %12.0f per frame-launch" % (c, sum(v), len(v), sum(v)/len(v)/6144))
PY

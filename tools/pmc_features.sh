#!/bin/bash
# PMC passes (<= 3 counters each) over the feature extractor's kernels; usage on the GPU box: bash tools/pmc_features.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_features
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=512
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_features.py $B kitti libstdcxx > $OUT/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "prs::" in k: acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-24s %16.0f total over %d launches, %12.0f per image-launch" % (c, sum(v), len(v), sum(v)/len(v)/$B))
PY

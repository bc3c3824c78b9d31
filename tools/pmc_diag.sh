#!/bin/bash
# stall diagnosis of the stereo matcher: small PMC passes (<=3 counters each, each under its own timeout)
# usage (GPU box, repo root): bash tools/pmc_diag.sh [extra env for the bench]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_diag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ATOMIC_RETURN" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_stereo.py 2048 2000 > $OUT/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stereo" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-28s %14.0f per frame (launches %d)" % (c, sum(v)/len(v)/2048, len(v)))
PY

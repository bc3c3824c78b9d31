#!/bin/bash
# same-box A/B of the brute-force matcher: _abtmp/base (a built copy of a reference commit) against the working tree
# usage (GPU box, repo root): bash tools/ab_bruteforce.sh [dense mode 0 / 2]
MODE=${1:-0}
cat > /tmp/ab_bf.py <<PY
import sys, os
sys.path.insert(0, "tools")
import bench_bruteforce as bb
mode = $MODE
for name, call in (("real 1024 x ~750", lambda: bb.run_real(1024, 50.0, quiet=True, target=1000, capacity=65536, dense=mode)),
                   ("real 256 x ~1350", lambda: bb.run_real(256, 50.0, quiet=True, target=2000, capacity=262144, dense=mode)),
                   ("real 8 x ~1350", lambda: bb.run_real(8, 50.0, quiet=True, target=2000, capacity=262144, dense=mode)),
                   ("real 1 x ~1350", lambda: bb.run_real(1, 50.0, quiet=True, target=2000, capacity=262144, dense=mode)),
                   ("real 1 x ~750", lambda: bb.run_real(1, 50.0, quiet=True, target=1000, capacity=65536, dense=mode)),
                   ("random 1024 x 2000", lambda: bb.run(1024, 2000, 50.0, quiet=True, dense=mode)),
                   ("random 1024 x 1000", lambda: bb.run(1024, 1000, 50.0, quiet=True, dense=mode)),
                   ("random 8 x 2000", lambda: bb.run(8, 2000, 50.0, quiet=True, dense=mode)),
                   ("random 1 x 2000", lambda: bb.run(1, 2000, 50.0, quiet=True, dense=mode))):
    print("   %-20s %.3f ms" % (name, call()["ms_per_launch"]))
PY
echo "base:"; (cd _abtmp/base && python /tmp/ab_bf.py 2>&1 | grep " ms")
echo "new:";  python /tmp/ab_bf.py 2>&1 | grep " ms"

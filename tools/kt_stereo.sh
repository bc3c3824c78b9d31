cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_stereo -- python3 $R/tools/bench_stereo.py 2048 2000 > $R/gpurun_out/kt_stereo.log 2>&1
echo "kernel-trace rc=$?"
grep -h "stereo" $R/gpurun_out/kt_stereo/*/*_kernel_stats.csv | cut -c1-160
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2_stereo -- python3 $R/tools/bench_stereo.py 256 2000 > $R/gpurun_out/pmc2_stereo.log 2>&1
echo "pmc rc=$?"
tail -3 $R/gpurun_out/pmc2_stereo.log | cut -c1-200

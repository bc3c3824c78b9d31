cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/tools/bench_stereo.py 2048 2000 > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/prof_pmc1 -- python3 $R/tools/bench_stereo.py 2048 2000 > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/prof_pmc2 -- python3 $R/tools/bench_stereo.py 2048 2000 > $R/gpurun_out/prof_pmc2.log 2>&1
ls -R $R/gpurun_out | head -40

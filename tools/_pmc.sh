cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc/p1 -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc/log1.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc/p2 -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc/log2.txt 2>&1
true

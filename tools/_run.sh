cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/stosa2
timeout -k 10 300 python -m pytest tests/test_wide_kernels.py -k wasserstein_attention -q 2>&1 | tail -2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stosa2 -o p -- python3 tools/bench_wide.py stosa --steps 10 > gpurun_out/stosa2/log.txt 2>&1
rm -f gpurun_out/stosa2/p_kernel_trace.csv
grep wattn gpurun_out/stosa2/p_kernel_stats.csv | cut -c1-140

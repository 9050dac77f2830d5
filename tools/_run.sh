cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof6
(timeout -k 10 200 python tools/check_seq_vs_staged.py 2 200 6; timeout -k 10 200 python tools/check_seq_vs_staged.py 4 48 3; timeout -k 10 200 python tools/check_seq_vs_staged.py 1 100 4) > gpurun_out/r2_seqcheck6.log 2>&1
timeout -k 10 400 python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py tests/test_dp_gpu.py -q > gpurun_out/r2_t18.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o p -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof6/log.txt 2>&1

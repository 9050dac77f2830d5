cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof2
timeout -k 10 300 python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py -q > gpurun_out/r2_t12.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o p -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof2/log.txt 2>&1

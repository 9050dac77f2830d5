cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/stats -o p -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r02/stats_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02/f -o fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02/f_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02/w -o write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02/w_log.txt 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/r02/f/fetch_counter_collection.csv gpurun_out/r02/w/write_counter_collection.csv k_seqtt_dec_fwd gpurun_out/r02/r02_dec_fwd_pmc.json
rm -f gpurun_out/r02/f/*kernel_trace* gpurun_out/r02/w/*kernel_trace* gpurun_out/r02/stats/*kernel_trace*

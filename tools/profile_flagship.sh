# Round-3 profile of the bench step on the GPU box: kernel stats, HBM traffic of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes),
# SQ / MFMA counters of the flagship, the fused logits + CE and the wide steps.  The program sits directly behind `--` (no env / bash -c hop).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/stats -o p -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-ndcg > gpurun_out/r03/stats_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03/f -o fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ndcg > gpurun_out/r03/f_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03/w -o write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ndcg > gpurun_out/r03/w_log.txt 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/r03/f/fetch_counter_collection.csv gpurun_out/r03/w/write_counter_collection.csv k_seqtt_dec_fwd gpurun_out/r03/r03_dec_fwd_pmc.json
python tools/pmc_traffic.py gpurun_out/r03/f/fetch_counter_collection.csv gpurun_out/r03/w/write_counter_collection.csv "k_seqtt_attn_pre_bwd<32, 1, false>" gpurun_out/r03/r03_attn_pre_bwd_pmc.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/r03/sq -o sq -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ndcg > gpurun_out/r03/sq_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/r03/sq_lce -o sq -- python3 tools/bench_lce.py > gpurun_out/r03/sq_lce_log.txt 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/r03/sq_bert -o sq -- python3 tools/bench_wide.py bert --steps 4 > gpurun_out/r03/sq_bert_log.txt 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/r03/sq_stosa -o sq -- python3 tools/bench_wide.py stosa --steps 4 > gpurun_out/r03/sq_stosa_log.txt 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/r03/sq_s256 -o sq -- python3 tools/bench_wide.py sasrec256 --steps 4 > gpurun_out/r03/sq_s256_log.txt 2>&1 || exit 1
python tools/pmc_mfma.py gpurun_out/r03/r03_mfma.json flagship=gpurun_out/r03/sq/sq_counter_collection.csv lce=gpurun_out/r03/sq_lce/sq_counter_collection.csv bert=gpurun_out/r03/sq_bert/sq_counter_collection.csv stosa=gpurun_out/r03/sq_stosa/sq_counter_collection.csv sasrec256=gpurun_out/r03/sq_s256/sq_counter_collection.csv > gpurun_out/r03/r03_mfma.txt
rm -f gpurun_out/r03/*/*kernel_trace* gpurun_out/r03/*/*counter_collection.csv

# Round-4 profile of the bench step on the GPU box: kernel stats + one-step timelines (resident and host-fed leg), HBM traffic of the dominant kernels and of the whole step
# (separate FETCH_SIZE / WRITE_SIZE passes), SQ / MFMA counters of the flagship and the fused logits + CE, small-batch probe.
# The program sits directly behind `--` (no env / bash -c hop).  Outputs under gpurun_out/r04p/ ; copy the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r04p && mkdir -p $O
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
B="python3 bench.py --no-cpu-baseline --no-ndcg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $B --steps 50 --warmup 10 > $O/stats_log.txt 2>&1 || exit 1
cp $O/stats/p_kernel_stats.csv $O/r04_kernel_stats.csv
python tools/trace_overlap.py $O/stats/p_kernel_trace.csv -25 > $O/r04_side_stream_timeline.txt
echo "---- host-fed leg (under the profiler the host is not ahead of the GPU, so the next batch is not published in time for the prefetch and k_step_begin reads the ring over PCIe: 23 us; unprofiled the prefetch hits -- 398 of 400 steps -- and k_step_begin takes 9 us, the two prefetch halves add ~7 us to k_loss_seeds + k_embed_bwd3)" >> $O/r04_side_stream_timeline.txt
python tools/trace_overlap.py $O/stats/p_kernel_trace.csv 25 >> $O/r04_side_stream_timeline.txt
echo stats done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o fetch -- $B --steps 6 --warmup 2 --no-roofline > $O/f_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o write -- $B --steps 6 --warmup 2 --no-roofline > $O/w_log.txt 2>&1 || exit 1
python tools/pmc_traffic.py $O/f/fetch_counter_collection.csv $O/w/write_counter_collection.csv k_seqtt_dec_fwd $O/r04_dec_fwd_pmc.json
python tools/pmc_traffic.py $O/f/fetch_counter_collection.csv $O/w/write_counter_collection.csv "k_seqtt_attn_pre_bwd<32, 1, false>" $O/r04_attn_pre_bwd_pmc.json
python tools/pmc_step_traffic.py $O/f/fetch_counter_collection.csv $O/w/write_counter_collection.csv $O/r04_step_traffic.json > $O/r04_step_traffic.txt
echo traffic done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq -o sq -- $B --steps 6 --warmup 2 --no-roofline > $O/sq_log.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_lce -o sq -- python3 tools/bench_lce.py > $O/sq_lce_log.txt 2>&1 || exit 1
python tools/pmc_mfma.py $O/r04_mfma.json flagship=$O/sq/sq_counter_collection.csv lce=$O/sq_lce/sq_counter_collection.csv > $O/r04_mfma.txt
echo sq done
python tools/small_batch_probe.py > $O/r04_small_batch.txt 2>&1
echo small done
rm -rf $O/stats $O/f $O/w $O/sq $O/sq_lce

/* adt_host.h -- C ABI of libadt_host.so: the host half of the SASRec-ADT training data path (plain C++ / OpenMP, no GPU calls).
 *
 * Reference counterparts (paths relative to the reference root): WarpDataset.sample_data + random_neq (sasrec/utils.py:73-77, 288-307) build
 * one (seq, dec, pos, neg) sample per user in Python inside 4 DataLoader workers (sasrec/main.py:88), and the loop turns each batch into device
 * tensors (sasrec/main.py:144-145, sasrec/model.py:34,37,40,53).  Here a batch is sampled natively, written as ONE packed int32 block straight
 * into a slot of the trainer's pinned ring, and fetched by the step's first kernel (include/adt_hip.h: adt_sasrec_step_begin_ring).
 * All pointers are HOST pointers; return 0 on success, negative on bad arguments / timeout.
 */
#ifndef ADT_HOST_H
#define ADT_HOST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int adt_host_version(void);

/* One batch of WarpDataset samples.  offsets: usernum + 2 entries, history of user u (1-based) = items[offsets[u] .. offsets[u+1]);
 * users: B user ids; seq / dec / pos / neg: B x L int32 row-major outputs (right-aligned history without its last item; the same shifted
 * right by one, dec[0] = 0; the next item; a uniformly random item the user has not interacted with, 0 where pos == 0).  The negative
 * stream of a row depends on (seed, user, global row index) only. */
int adt_host_sample_batch(const int64_t* offsets, const int32_t* items, const int32_t* users, int B, int L, int itemnum, uint64_t seed,
                          int32_t* seq, int32_t* dec, int32_t* pos, int32_t* neg, int nthreads);
/* The same for rows [b0, b0 + B) of a global batch: what a data-parallel rank samples (the union over ranks is the 1-process batch). */
int adt_host_sample_rows(const int64_t* offsets, const int32_t* items, const int32_t* users, int B, int b0, int L, int itemnum,
                         uint64_t seed, int32_t* seq, int32_t* dec, int32_t* pos, int32_t* neg, int nthreads);
/* sum over users of min(len(u) - 1, L): the number of pos != 0 positions of the batch = the BCE normaliser of sasrec/main.py:150-153 */
int64_t adt_host_count_targets(const int64_t* offsets, const int32_t* users, int B, int L);
/* dst = [seq | dec | pos | neg] (T ids each) + (n_bce, n_mse, n_nll) as float bits + 0.  A source equal to its destination is skipped. */
int adt_host_pack_batch(int32_t* dst, const int32_t* seq, const int32_t* dec, const int32_t* pos, const int32_t* neg, int64_t T, float n_bce,
                        float n_mse, float n_nll);
/* spin until the uint32 counter at p (stored by the GPU into pinned host memory) has reached v (wrap-around safe); -1 after timeout_us */
int adt_host_wait_ge(const volatile uint32_t* p, uint32_t v, int64_t timeout_us);
/* release-store of a counter the GPU polls in pinned host memory (the producer's "batches written" word of the id ring) */
int adt_host_store_release(volatile uint32_t* p, uint32_t v);

#ifdef __cplusplus
}
#endif
#endif

// Host-side batch sampler for the SASRec-ADT trainer (libadt_host.so, plain C ABI, OpenMP).
// Replaces the per-sample Python loops of the reference's WarpDataset.sample_data / random_neq
// (sasrec/utils.py:73-77, 288-307): right-aligned history as `seq`, the same shifted right by one as `dec`
// (dec[0] = 0), the next item as `pos`, and for every real position a uniformly random item the user has NOT
// interacted with as `neg` (0 where pos == 0).  At GPU step rates (a 256-sequence batch every ~1.5 ms) the Python
// sampler (~1 ms per user) bounds the whole job; this one fills a batch in ~0.2 ms.
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <sched.h>
#include <vector>
#include "../../include/adt_host.h"
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

extern "C" {

int adt_host_version(void) { return 3; }

// offsets: usernum + 2 entries (users are 1-based; history of user u = items[offsets[u] .. offsets[u+1]))
// users: B user ids; outputs: B x L int32, row-major.  Returns 0, or -1 on bad arguments.
// b0: index of users[0] in the GLOBAL batch (a data-parallel rank samples only its rows [b0, b0 + B): the negative stream of a row
// depends on (seed, user, global row) alone, so the shards of all ranks together are exactly the batch one process would have drawn)
int adt_host_sample_rows(const int64_t* offsets, const int32_t* items, const int32_t* users, int B, int b0, int L, int itemnum,
                         uint64_t seed, int32_t* seq, int32_t* dec, int32_t* pos, int32_t* neg, int nthreads);

int adt_host_sample_batch(const int64_t* offsets, const int32_t* items, const int32_t* users, int B, int L, int itemnum,
                          uint64_t seed, int32_t* seq, int32_t* dec, int32_t* pos, int32_t* neg, int nthreads) {
  return adt_host_sample_rows(offsets, items, users, B, 0, L, itemnum, seed, seq, dec, pos, neg, nthreads);
}

// number of positions with a target (pos != 0) the rows of `users` will have: sum of min(len(u) - 1, L) -- the BCE normaliser of a batch
// (sasrec/main.py:150-153) without sampling it
int64_t adt_host_count_targets(const int64_t* offsets, const int32_t* users, int B, int L) {
  int64_t n = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t len = offsets[users[b] + 1] - offsets[users[b]];
    const int64_t k = len - 1 < L ? len - 1 : L;
    n += k > 0 ? k : 0;
  }
  return n;
}

// one packed id block of the trainer's ring (include/adt_hip.h: adt_sasrec_step_begin_ring): [seq | dec | pos | neg] (T ids each), the three
// loss normalisers as float bits, a zero word.  Any of the four sources may already BE its part of dst (sampled in place): then it is skipped.
int adt_host_pack_batch(int32_t* dst, const int32_t* seq, const int32_t* dec, const int32_t* pos, const int32_t* neg, int64_t T, float n_bce,
                        float n_mse, float n_nll) {
  if (!dst || T < 0) return -1;
  const int32_t* src[4] = {seq, dec, pos, neg};
  for (int k = 0; k < 4; ++k)
    if (src[k] && src[k] != dst + k * T) memcpy(dst + k * T, src[k], sizeof(int32_t) * (size_t)T);
  const float nm[3] = {n_bce, n_mse, n_nll};
  memcpy(dst + 4 * T, nm, sizeof(nm));
  dst[4 * T + 3] = 0;
  return 0;
}

// publish a 32-bit counter the GPU reads from pinned host memory: everything the caller wrote before (the batch's ids and normalisers) is
// visible to a reader that sees the new value
int adt_host_store_release(volatile uint32_t* p, uint32_t v) { __atomic_store_n(const_cast<uint32_t*>(p), v, __ATOMIC_RELEASE); return 0; }

// wait until the 32-bit counter at p (written by the GPU into pinned host memory) has reached v, wrap-around safe; 0 ok, -1 timeout
int adt_host_wait_ge(const volatile uint32_t* p, uint32_t v, int64_t timeout_us) {
  if ((int32_t)(*p - v) >= 0) return 0;
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t spin = 0;; ++spin) {
    if ((int32_t)(*p - v) >= 0) return 0;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((spin & 1023) == 1023) {
      clock_gettime(CLOCK_MONOTONIC, &t);
      const int64_t us = (int64_t)(t.tv_sec - t0.tv_sec) * 1000000 + (t.tv_nsec - t0.tv_nsec) / 1000;
      if (us > timeout_us) return -1;
      if (us > 200) sched_yield();
    }
  }
}

int adt_host_sample_rows(const int64_t* offsets, const int32_t* items, const int32_t* users, int B, int b0, int L, int itemnum,
                         uint64_t seed, int32_t* seq, int32_t* dec, int32_t* pos, int32_t* neg, int nthreads) {
  if (!offsets || !items || !users || B < 0 || L <= 0 || itemnum <= 0) return -1;
  const int words = (itemnum + 64) / 64;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    std::vector<uint64_t> seen((size_t)words);
#pragma omp for schedule(dynamic, 8)
    for (int b = 0; b < B; ++b) {
      int32_t* s = seq + (size_t)b * L;
      int32_t* d = dec + (size_t)b * L;
      int32_t* p = pos + (size_t)b * L;
      int32_t* ng = neg + (size_t)b * L;
      memset(s, 0, sizeof(int32_t) * L);
      memset(d, 0, sizeof(int32_t) * L);
      memset(p, 0, sizeof(int32_t) * L);
      memset(ng, 0, sizeof(int32_t) * L);
      const int u = users[b];
      const int64_t lo = offsets[u], hi = offsets[u + 1];
      const int64_t len = hi - lo;
      int n = (int)(len - 1 < L ? len - 1 : L);
      if (n <= 0) continue;
      const int32_t* h = items + hi - (n + 1);          // last n+1 items of the history
      for (int k = 0; k < n; ++k) {
        s[L - n + k] = h[k];
        p[L - n + k] = h[k + 1];
        if (L - n + k + 1 < L) d[L - n + k + 1] = h[k];
      }
      memset(seen.data(), 0, sizeof(uint64_t) * words);
      int distinct = 0;
      for (int64_t i = lo; i < hi; ++i) {
        const int it = items[i];
        if (it >= 0 && it <= itemnum) {
          uint64_t& wd = seen[it >> 6];
          const uint64_t bit = 1ull << (it & 63);
          distinct += (wd & bit) ? 0 : 1;
          wd |= bit;
        }
      }
      if (distinct >= itemnum) continue;                // nothing left to draw (degenerate): leave neg = 0
      uint64_t st = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(u + 1)) ^ ((uint64_t)(b + b0) << 32);
      for (int k = 0; k < n; ++k) {
        int t;
        do {
          t = 1 + (int)(splitmix64(st) % (uint64_t)itemnum);
        } while (seen[t >> 6] & (1ull << (t & 63)));
        ng[L - n + k] = t;
      }
    }
  }
  return 0;
}

}  // extern "C"

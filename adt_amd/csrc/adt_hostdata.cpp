// Host-side batch sampler for the SASRec-ADT trainer (libadt_host.so, plain C ABI, OpenMP).
// Replaces the per-sample Python loops of the reference's WarpDataset.sample_data / random_neq
// (sasrec/utils.py:73-77, 288-307): right-aligned history as `seq`, the same shifted right by one as `dec`
// (dec[0] = 0), the next item as `pos`, and for every real position a uniformly random item the user has NOT
// interacted with as `neg` (0 where pos == 0).  At GPU step rates (a 256-sequence batch every ~1.5 ms) the Python
// sampler (~1 ms per user) bounds the whole job; this one fills a batch in ~0.2 ms.
#include <stdint.h>
#include <string.h>
#include <vector>
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

int adt_host_version(void) { return 1; }

// offsets: usernum + 2 entries (users are 1-based; history of user u = items[offsets[u] .. offsets[u+1]))
// users: B user ids; outputs: B x L int32, row-major.  Returns 0, or -1 on bad arguments.
int adt_host_sample_batch(const int64_t* offsets, const int32_t* items, const int32_t* users, int B, int L, int itemnum,
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
      uint64_t st = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(u + 1)) ^ ((uint64_t)b << 32);
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

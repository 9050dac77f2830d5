// Deterministic item-table / positional-table gradient (sasrec/model.py:34-41, :53-59, :72-76 reversed).
//
// The reference's autograd sums the rows that hit one item (embedding index_add) in a fixed order on the CPU.  The float-atomic scatters of
// adt_misc.cuh (k_logits_bwd_scatter, k_embed_bwd64) are bound by the memory-side atomic unit (~20 ns per request to one 64-byte line: a
// popular item's row is thousands of requests deep even with 16 replicas) and add in arrival order, so two runs differ in the last bits.
// Here the step's ids are SORTED once (they are known when the step begins: the sort runs beside the forward) and every item's rows are
// summed by one owner in sorted order:
//   k_isort_hist / k_isort_scan_chunks / k_isort_place : stable counting sort of the entries e = src * T + t
//       (src: which id array, t: token) by item -- 256 chunks of consecutive entries, one wave each, an LDS histogram per chunk, ranks of
//       equal keys inside a wave from ballots (lane order), so the order inside an item is the entry order: a pure function of the ids.
//   k_item_segsum : segmented sums over the sorted list, 1,024 entries per workgroup: 64 sub-ranges of 16 entries, one 16-lane group each
//       (a lane holds four of the 64 features); a segment inside a sub-range is written by its group, pieces of segments that cross
//       sub-ranges are joined by wave 0 in order, pieces that cross workgroups go to a carry area;
//   k_item_carry  : the carry chains, one wave per chain, in workgroup order.
//   k_posemb_sum  : dP[l] = sum_b of the masked rows of position l, b ascending, one workgroup per position.
// No float atomics: every sum has one fixed order.  HBM-bound gathers of 256-byte rows; no MFMA.
#pragma once
#include "adt_common.cuh"

namespace adt {

constexpr int IS_NCH = 256;          // chunks of the entry list in the counting sort (one wave each)
constexpr int IS_MAXV1 = 16000;      // item_num + 1 the LDS histogram of a chunk holds (< 64 KB)
constexpr int IG_WAVES = 16;         // k_item_segsum: waves per workgroup
constexpr int IG_PER_GROUP = 16;     // sorted entries per 16-lane group (a sub-range)
constexpr int IG_SUBS = IG_WAVES * 4;                     // sub-ranges per workgroup
constexpr int IG_PER_BLOCK = IG_SUBS * IG_PER_GROUP;      // 1,024 sorted entries per workgroup
constexpr int IG_U = 8;              // rows in flight per group

struct ItemSortArgs {
  const int* ids[4];                 // the id arrays (T entries each); entry e = src * T + t
  int nsrc, T, V1;                   // V1 = item_num + 1 (id 0 = padding: not an entry)
  int* hist;                         // [IS_NCH][V1]: counts per chunk, rewritten to the exclusive prefix over the chunks
  int* base;                         // [V1 + 1]: per item the prefix of the totals inside its 64-item group ; base[V1] = number of entries
  int* bsum;                         // [ceil(V1 / 64)]: totals of the 64-item groups
  int* perm;                         // [nsrc * T]: the entries sorted by item (stable)
  int* pitem;                        // [nsrc * T]: item of each sorted entry
  // the gather plan of k_item_segsum, written with the sorted list (everything about an entry that does not depend on gradient VALUES):
  const float* rows[4]; const float* coef[4]; int kind[4]; uint32_t row_offset;
  uint64_t* prow;                    // [nsrc * T]: address of the entry's 64-float row
  uint64_t* pcoef;                   // [nsrc * T]: address of its scale factor (kind 1) or of a zero
  uint32_t* pmeta;                   // [nsrc * T]: (t + row_offset) * 64 | src << 1 | (kind == 1)
};
static __device__ __attribute__((aligned(256))) const float ig_zero_row[64] = {};

ADT_DEVICE_INLINE int is_key(const ItemSortArgs& a, int e) {
  typedef const int __attribute__((address_space(1))) * gi;
  const int T = a.T;
  const int src = (e >= T) + (e >= 2 * T) + (e >= 3 * T);
  const int* p = src == 0 ? a.ids[0] : (src == 1 ? a.ids[1] : (src == 2 ? a.ids[2] : a.ids[3]));
  return *(gi)(p + (e - src * T));
}
constexpr int IS_MAXR = 16;          // rounds of 64 entries a chunk's wave keeps in registers (16 x 64 x 256 chunks = 262,144 entries)

// the keys of a chunk, every load issued before the first use (a round per loop iteration was a dependent round trip per 64 entries:
// 13 of them per wave at the flagship step)
ADT_DEVICE_INLINE void is_load_keys(const ItemSortArgs& a, int e0, int e1, int lane, int (&key)[IS_MAXR]) {
#pragma unroll
  for (int r = 0; r < IS_MAXR; ++r) {
    const int e = e0 + r * 64 + lane;
    int k = e < e1 ? is_key(a, e < e1 ? e : e0) : 0;
    key[r] = (k > 0 && k < a.V1) ? k : 0;
  }
}

__global__ __launch_bounds__(64) void k_isort_hist(ItemSortArgs a) {
  extern __shared__ int sh[];
  const int lane = threadIdx.x, c = blockIdx.x, N = a.nsrc * a.T;
  const int CH = (N + IS_NCH - 1) / IS_NCH, e0 = c * CH, e1 = min(N, e0 + CH);
  int key[IS_MAXR];
  is_load_keys(a, e0, e1, lane, key);
  for (int i = lane; i < a.V1; i += 64) sh[i] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < IS_MAXR; ++r)
    if (key[r] > 0) atomicAdd(&sh[key[r]], 1);
  __syncthreads();
  for (int i = lane; i < a.V1; i += 64) a.hist[(size_t)c * a.V1 + i] = sh[i];
}

// per item: exclusive prefix of its counts over the chunks (in place) and its total -> base[item].  A workgroup takes 64 items: the
// [256 chunks][64 items] tile goes through LDS (256-byte rows in, 256-byte rows out), a wave scans an item's 256 counts in one go
// (four per lane + a 64-lane scan).  (One thread per item walking the 256 chunks was 32 dependent round trips: 13.7 us.)
__global__ __launch_bounds__(256) void k_isort_scan_chunks(ItemSortArgs a) {
  __shared__ int tile[IS_NCH][65];
  __shared__ int stot[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, k0 = blockIdx.x * 64;
  const int k = k0 + lane;
  for (int c0 = w; c0 < IS_NCH; c0 += 64) {            // sixteen rows in flight per wave
    int h[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) h[u] = k < a.V1 ? a.hist[(size_t)(c0 + 4 * u) * a.V1 + k] : 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) tile[c0 + 4 * u][lane] = h[u];
  }
  __syncthreads();
  for (int it = w * 16; it < w * 16 + 16; ++it) {      // item k0 + it: lane l owns chunks 4 l .. 4 l + 3
    int h[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = tile[4 * lane + j][it]; s += h[j]; }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    int run = inc - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) { tile[4 * lane + j][it] = run; run += h[j]; }
    if (lane == 63) stot[it] = k0 + it < a.V1 ? inc : 0;
  }
  __syncthreads();
  if (w == 0) {      // the 64 totals: exclusive prefix inside the workgroup -> base[item], their sum -> bsum[workgroup] (k_isort_place adds the
                     // prefix over the workgroups: no scan kernel in between)
    const int t = stot[lane];
    int inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (k < a.V1) a.base[k] = inc - t;
    if (lane == 63) a.bsum[blockIdx.x] = inc;
  }
  for (int c = w; c < IS_NCH; c += 4)
    if (k < a.V1) a.hist[(size_t)c * a.V1 + k] = tile[c][lane];
}

__global__ __launch_bounds__(64) void k_isort_place(ItemSortArgs a) {
  extern __shared__ int sh[];
  const int lane = threadIdx.x, c = blockIdx.x, N = a.nsrc * a.T;
  const int CH = (N + IS_NCH - 1) / IS_NCH, e0 = c * CH, e1 = min(N, e0 + CH);
  int key[IS_MAXR];
  is_load_keys(a, e0, e1, lane, key);
  // the cursor of every item for this chunk = base[item] (prefix inside its 64-item group) + the prefix of bsum over the groups + this
  // chunk's prefix over the chunks.  Every wave scans the <= 250 group sums itself (four per lane + a 64-lane scan).
  static_assert(IS_MAXV1 <= 64 * 64 * 4, "four groups per lane");
  const int ngrp = (a.V1 + 63) / 64;
  int gs[4], gsum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { gs[j] = 4 * lane + j < ngrp ? a.bsum[4 * lane + j] : 0; gsum += gs[j]; }
  int ginc = gsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(ginc, o, 64); if (lane >= o) ginc += t; }
  int* const goff = sh + a.V1;                          // [256] exclusive prefix of the groups
  {
    int run = ginc - gsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) { goff[4 * lane + j] = run; run += gs[j]; }
  }
  if (c == 0 && lane == 63) a.base[a.V1] = ginc;        // number of entries (k_item_segsum's bound)
  __syncthreads();
  for (int i0 = lane; i0 < a.V1; i0 += 64 * 8) {
    int bs[8], hs[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 64 * u, ii = i < a.V1 ? i : 0;
      bs[u] = a.base[ii]; hs[u] = a.hist[(size_t)c * a.V1 + ii];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 64 * u < a.V1) sh[i0 + 64 * u] = bs[u] + hs[u] + goff[(i0 + 64 * u) >> 6];
  }
  __syncthreads();
  const int nbits = 32 - __builtin_clz((unsigned)(a.V1 > 1 ? a.V1 - 1 : 1));
  const int T = a.T;
#pragma unroll
  for (int r = 0; r < IS_MAXR; ++r) {
    if (e0 + r * 64 >= e1) break;                  // wave-uniform
    const int k = key[r], e = e0 + r * 64 + lane;
    // the lanes with the same key, from one ballot per key BIT (lanes that agree with this lane on every bit); rank = those below this lane
    unsigned long long mine = ~0ull;
    for (int b = 0; b < nbits; ++b) {
      const unsigned long long m = __ballot((k >> b) & 1);
      mine &= ((k >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(mine & ((1ull << lane) - 1ull)), cnt = __popcll(mine);
    if (k > 0) {
      const int pos = sh[k] + rank;
      a.perm[pos] = e;
      a.pitem[pos] = k;
      if (a.prow) {
        const int src = (e >= T) + (e >= 2 * T) + (e >= 3 * T), t = e - src * T;
        const int kind = src == 0 ? a.kind[0] : (src == 1 ? a.kind[1] : (src == 2 ? a.kind[2] : a.kind[3]));
        const float* rows = src == 0 ? a.rows[0] : (src == 1 ? a.rows[1] : (src == 2 ? a.rows[2] : a.rows[3]));
        const float* coef = src == 0 ? a.coef[0] : (src == 1 ? a.coef[1] : (src == 2 ? a.coef[2] : a.coef[3]));
        a.prow[pos] = (uint64_t)(rows + (size_t)t * 64);
        a.pcoef[pos] = kind == 1 ? (uint64_t)(coef + t) : (uint64_t)ig_zero_row;
        a.pmeta[pos] = ((uint32_t)(t + a.row_offset) * 64u) | ((uint32_t)src << 1) | (kind == 1 ? 1u : 0u);
      }
    }
    __syncthreads();                              // one wave: orders the cursor reads above before the updates below
    if (k > 0 && rank == 0) sh[k] += cnt;
    __syncthreads();
  }
}

// ---- segmented sums --------------------------------------------------------------------------------------------------------------------
// value of sorted entry e for feature `lane`:
//   kind 0 (an id array of an embedding layer): rows[t] * emb_scale * keep / (1 - p)   (the forward's dropout decision of that element)
//   kind 1 (pos / neg of the logits)          : rows[t] * coef[t]
struct ItemSegArgs {
  const int* pitem; const int* total;      // sorted items ; *total = base[V1]
  const uint64_t* prow; const uint64_t* pcoef; const uint32_t* pmeta;      // the gather plan (k_isort_place)
  int nblk;
  uint32_t site[4];
  uint32_t src_mask;                 // bit s: this pass sums source s (the others count as zero rows)
  const uint32_t* seed; uint32_t thr; float dscale; float emb_scale;
  float* dE;                         // [V1][64]: dE[item] = sum, or += sum when `rmw` (one owner per item and pass)
  int rmw;                           // 0: plain stores -- dE must not hold anything worth keeping (a read-modify-write is a dependent round
                                     // trip per segment)
  float* carry;                      // [nblk][2][64]: head / tail pieces of the segments that cross workgroups
  int* cflag;                        // [nblk][4]: head item (-1 none), tail item (-1 none), whole (1: the workgroup is one piece, in slot 0), pad
};

ADT_DEVICE_INLINE void ig_emit(const ItemSegArgs& a, int item, int q, const float4& v) {
  float4* p = reinterpret_cast<float4*>(a.dE + (size_t)item * 64 + 4 * q);
  if (a.rmw) { const float4 o = *p; *p = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w); }
  else *p = v;
}

// SIXTEEN lanes per entry (a float4 of the row each), so a wave walks four sub-ranges of the sorted list at once and everything per entry --
// source, row address, scale factor, dropout word -- is ordinary per-lane vector arithmetic (one hash word per lane: its four features
// share it).  With one entry per wave-iteration (lane = feature) that work was scalar code every lane waited for: ~150 wave instructions
// per entry, 55 us for the 205k entries of the flagship step; with per-lane descriptors + v_readlane 26 us; this form: see DESIGN.md.
// A sub-range is IG_PER_GROUP consecutive sorted entries; a segment inside a sub-range is written by its group, pieces that cross
// sub-ranges are joined by wave 0 in sub-range order, pieces that cross workgroups go to the carry area (k_item_carry).
ADT_DEVICE_INLINE void item_segsum_body(const ItemSegArgs& a, const int b) {
  __shared__ __attribute__((aligned(16))) float sp[IG_SUBS][2][64];          // per sub-range: left-open piece, right-open piece
  __shared__ int sflag[IG_SUBS][4];                                          // left item (-1), right item (-1), through (1: one piece, in slot 0)
  typedef const f32x4 __attribute__((address_space(1))) * gf4;
  typedef const float __attribute__((address_space(1))) * gf;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, q = lane & 15;
  const int N = *a.total;
  uint32_t key[4] = {0u, 0u, 0u, 0u};
  if (a.thr) {
    const uint32_t sd = *a.seed;
#pragma unroll
    for (int s = 0; s < 4; ++s) key[s] = adt_site_key(sd, a.site[s]);
  }
  const int sub = w * 4 + g;
  const int r0 = b * IG_PER_BLOCK + sub * IG_PER_GROUP, r1 = min(N, r0 + IG_PER_GROUP);
  int litem = -1, ritem = -1, through = 0;
  if (r0 < r1) {
    const int prev_item = r0 > 0 ? a.pitem[r0 - 1] : -1;
    const int next_item = r1 < N ? a.pitem[r1] : -1;
    int iv[IG_PER_GROUP];
    uint64_t px[IG_PER_GROUP], pc[IG_PER_GROUP];
    uint32_t pm[IG_PER_GROUP];
#pragma unroll
    for (int i = 0; i < IG_PER_GROUP; ++i) {        // the sub-range's plan (the sixteen lanes of a group read the same words)
      const int r = r0 + i < r1 ? r0 + i : r0;
      iv[i] = r0 + i < r1 ? a.pitem[r] : -1;
      px[i] = a.prow[r]; pc[i] = a.pcoef[r]; pm[i] = a.pmeta[r];
    }
    int cur = -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool open_left = false;
#pragma unroll
    for (int i0 = 0; i0 < IG_PER_GROUP; i0 += IG_U) {
      f32x4 x[IG_U];
      float c[IG_U];
#pragma unroll
      for (int u = 0; u < IG_U; ++u) {                // IG_U rows in flight per group
        const bool on = iv[i0 + u] >= 0 && ((a.src_mask >> ((pm[i0 + u] >> 1) & 3u)) & 1u) != 0;
        x[u] = *(on ? (gf4)(px[i0 + u] + 16u * (uint32_t)q) : (gf4)(ig_zero_row + 4 * q));
        c[u] = *(on ? (gf)pc[i0 + u] : (gf)ig_zero_row);
      }
#pragma unroll
      for (int u = 0; u < IG_U; ++u) {
        const int item = iv[i0 + u];
        if (item < 0) continue;
        const uint32_t meta = pm[i0 + u];
        float4 v;
        if (meta & 1u) {
          v = make_float4(x[u][0] * c[u], x[u][1] * c[u], x[u][2] * c[u], x[u][3] * c[u]);
        } else {
          v = make_float4(x[u][0] * a.emb_scale, x[u][1] * a.emb_scale, x[u][2] * a.emb_scale, x[u][3] * a.emb_scale);
          if (a.thr) {
            const uint32_t src = (meta >> 1) & 3u;
            const uint32_t kw = src == 0 ? key[0] : (src == 1 ? key[1] : (src == 2 ? key[2] : key[3]));
            const uint32_t bits = adt_keep4(kw, (meta & ~63u) + 4u * (uint32_t)q, a.thr);
            v.x = (bits & 1u) ? v.x * a.dscale : 0.f; v.y = (bits & 2u) ? v.y * a.dscale : 0.f;
            v.z = (bits & 4u) ? v.z * a.dscale : 0.f; v.w = (bits & 8u) ? v.w * a.dscale : 0.f;
          }
        }
        if (item != cur) {
          if (cur >= 0) {                           // the segment of `cur` ends inside this sub-range
            if (open_left) { *reinterpret_cast<float4*>(&sp[sub][0][4 * q]) = acc; litem = cur; }
            else ig_emit(a, cur, q, acc);
          }
          open_left = (cur < 0) && (item == prev_item);      // the first segment may continue one of the previous sub-range
          cur = item;
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    if (cur >= 0) {
      const bool open_right = cur == next_item;
      if (open_left) { *reinterpret_cast<float4*>(&sp[sub][0][4 * q]) = acc; litem = cur; through = open_right ? 1 : 0; }
      else if (open_right) { *reinterpret_cast<float4*>(&sp[sub][1][4 * q]) = acc; ritem = cur; }
      else ig_emit(a, cur, q, acc);
    }
  }
  if (q == 0) { sflag[sub][0] = litem; sflag[sub][1] = ritem; sflag[sub][2] = through; }
  __syncthreads();
  if (w != 0) return;
  // wave 0 (lane = feature) joins the pieces in sub-range order.  chain: the running piece ; chain_ext: it started in an earlier workgroup
  int chain = -1;
  bool chain_ext = false;
  float cacc = 0.f;
  int head_item = -1, whole = 0;
  static_assert(IG_SUBS == 64, "lane k keeps the flags of sub-range k");
  const int fl = sflag[lane][0], fr = sflag[lane][1], ft = sflag[lane][2];      // (read per iteration they were three dependent LDS round trips)
  for (int k = 0; k < IG_SUBS; ++k) {
    const int li = __builtin_amdgcn_readlane(fl, k), ri = __builtin_amdgcn_readlane(fr, k), th = __builtin_amdgcn_readlane(ft, k);
    if (li < 0 && ri < 0) continue;
    if (li >= 0) {
      if (chain != li) { chain = li; chain_ext = true; cacc = 0.f; }      // only at the workgroup's first piece: continues an earlier workgroup
      cacc += sp[k][0][lane];
      if (!th) {                                 // the piece ends inside sub-range k
        if (chain_ext) { a.carry[((size_t)b * 2 + 0) * 64 + lane] = cacc; head_item = chain; }
        else if (a.rmw) a.dE[(size_t)chain * 64 + lane] += cacc;
        else a.dE[(size_t)chain * 64 + lane] = cacc;
        chain = -1;
      }
    }
    if (ri >= 0) { chain = ri; chain_ext = false; cacc = sp[k][1][lane]; }
  }
  int tail_item = -1;
  if (chain >= 0) {                               // open towards the next workgroup
    if (chain_ext) { a.carry[((size_t)b * 2 + 0) * 64 + lane] = cacc; head_item = chain; whole = 1; }      // the whole workgroup is one piece
    else { a.carry[((size_t)b * 2 + 1) * 64 + lane] = cacc; tail_item = chain; }
  }
  if (lane == 0) { a.cflag[b * 4 + 0] = head_item; a.cflag[b * 4 + 1] = tail_item; a.cflag[b * 4 + 2] = whole; }
}

// one wave per workgroup b of k_item_segsum whose tail piece starts a chain: tail(b) + whole(b+1 ..) + head(first workgroup that is not whole).
// The chain's length comes from one 64-lane look at the flags of the next workgroups (ballots); its rows are then read sixteen at a time.
__global__ __launch_bounds__(64) void k_item_carry(ItemSegArgs a) {
  const int lane = threadIdx.x, b = blockIdx.x;
  const int item = a.cflag[b * 4 + 1];
  if (item < 0) return;
  float acc = a.carry[((size_t)b * 2 + 1) * 64 + lane];
  for (int c0 = b + 1; c0 < a.nblk; c0 += 64) {
    const int c = c0 + lane;
    const bool in = c < a.nblk;
    const bool cont = in && a.cflag[c * 4 + 0] == item;
    const bool wh = cont && a.cflag[c * 4 + 2] != 0;
    const unsigned long long mc = __ballot(cont), mw = __ballot(wh);
    // the chain covers lanes 0 .. m-1: every lane below the first that does not continue, up to and including the first that is not whole
    const unsigned long long stop = ~mc | (mc & ~mw);
    const int first_stop = stop ? __ffsll((long long)stop) - 1 : 64;
    const int m = first_stop < 64 && ((mc >> first_stop) & 1ull) ? first_stop + 1 : first_stop;
    for (int j0 = 0; j0 < m; j0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = j0 + u < m ? a.carry[((size_t)(c0 + j0 + u) * 2 + 0) * 64 + lane] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
    if (m < 64) break;
  }
  if (a.rmw) a.dE[(size_t)item * 64 + lane] += acc;
  else a.dE[(size_t)item * 64 + lane] = acc;
}

// ---- positional table: dP[l][f] = sum_b [ids[b, l] != 0] * keep / (1 - p) * dX[b, l][f], b ascending, for up to two embedding layers -------
struct PosSumArgs {
  const int* ids[2]; const float* dX[2]; uint32_t site[2]; int nsrc;
  int B, L; const uint32_t* seed; uint32_t thr; float dscale; uint32_t row_offset;
  float* dP;                         // [L][64], += (one owner per position)
};
constexpr int PS_WAVES = 16;         // waves per position: (source, eighth of the batch) each -- the sums are chains of dependent round trips
ADT_DEVICE_INLINE void posemb_sum_body(const PosSumArgs& a, const int l) {
  __shared__ float sw[PS_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t sd = a.thr ? *a.seed : 0u;
  float acc = 0.f;
  const int s = w >> 3, part = w & 7;
  const int per = (a.B + 7) / 8, b0 = part * per, b1 = min(a.B, b0 + per);
  if (s < a.nsrc) {
    const uint32_t key = a.thr ? adt_site_key(sd, a.site[s]) : 0u;
    for (int bb0 = b0; bb0 < b1; bb0 += 64) {      // the ids of up to 64 rows in one load, then the rows sixteen at a time
      const int idv = bb0 + lane < b1 ? a.ids[s][(size_t)(bb0 + lane) * a.L + l] : 0;
      const int n = min(64, b1 - bb0);
      for (int j0 = 0; j0 < n; j0 += 16) {
        int id[16];
        float g[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + u < n ? j0 + u : n - 1;
          id[u] = j0 + u < n ? __builtin_amdgcn_readlane(idv, j) : 0;
          g[u] = a.dX[s][((size_t)(bb0 + j) * a.L + l) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (id[u] == 0) continue;
          float v = g[u];
          if (a.thr) v = adt_keep(key, (uint32_t)((bb0 + j0 + u) * a.L + l + a.row_offset) * 64u + (uint32_t)lane, a.thr) ? v * a.dscale : 0.f;
          acc += v;
        }
      }
    }
  }
  sw[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < PS_WAVES; ++k) t += sw[k][lane];      // fixed order: source 0's eighths, then source 1's
    a.dP[(size_t)l * 64 + lane] += t;
  }
}

static_assert(PS_WAVES == IG_WAVES, "one launch runs both bodies");
__global__ __launch_bounds__(IG_WAVES * 64) void k_item_segsum(ItemSegArgs a) { item_segsum_body(a, blockIdx.x); }
__global__ __launch_bounds__(PS_WAVES * 64) void k_posemb_sum(PosSumArgs a) { posemb_sum_body(a, blockIdx.x); }
// the two in one launch (they are independent, and each alone leaves most of the chip idle): workgroups [0, nblk) the item segments, then one
// per position
__global__ __launch_bounds__(IG_WAVES * 64) void k_item_segsum_posemb(ItemSegArgs a, PosSumArgs p) {
  if ((int)blockIdx.x < a.nblk) item_segsum_body(a, blockIdx.x);
  else posemb_sum_body(p, (int)blockIdx.x - a.nblk);
}

}  // namespace adt

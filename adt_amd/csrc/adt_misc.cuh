// Embedding gather / scatter, head classifier, pos/neg logits + BCE, reconstruction (MSE) seeds,
// clip + Adam.  HBM-bound integer/elementwise work: coalesced float4 accesses, no MFMA.
#pragma once
#include "adt_common.cuh"
#include "adt_wave.cuh"

namespace adt {

// ---------------------------------------------------------------------------------------------
// sasrec/model.py:34-41 / :53-59 : x = dropout(E[ids] * sqrt(d) + P[l]) * (ids != 0)
struct EmbedArgs {
  const int* ids;            // T = B*L
  const float* E; const float* P;
  int T, L, d;
  float scale;               // sqrt(d)
  DropCfg drop; uint32_t row_offset;
  float* X;                  // fwd out (T x d)
  const float* dX;           // bwd in
  float* dE; float* dP;      // bwd out (atomics)
};

__global__ __launch_bounds__(256) void k_embed_fwd(EmbedArgs a) {
  const int V = a.d / 4;
  const uint32_t key = drop_key(a.drop);
  const size_t n = (size_t)a.T * V;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / V), c4 = (int)(i % V) * 4;
    const int id = a.ids[row];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (id != 0) {
      const float4 e = *reinterpret_cast<const float4*>(a.E + (size_t)id * a.d + c4);
      const float4 p = *reinterpret_cast<const float4*>(a.P + (size_t)(row % a.L) * a.d + c4);
      v[0] = e.x * a.scale + p.x; v[1] = e.y * a.scale + p.y; v[2] = e.z * a.scale + p.z; v[3] = e.w * a.scale + p.w;
      if (a.drop.thr) {
        const uint32_t base = (uint32_t)(row + a.row_offset) * (uint32_t)a.d + (uint32_t)c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
      }
    }
    *reinterpret_cast<float4*>(a.X + (size_t)row * a.d + c4) = *reinterpret_cast<float4*>(v);
  }
}

// Row scatter-add into the item table: dE[ids[row]] += G[row] * rowscale[row] * keep/(1-p) * scale.
// One wave per row so that every atomic wave-instruction covers 256 contiguous bytes of one table row (the
// shape that runs at the full float-atomic rate, MI355X_MICROARCH.md "Global float atomics"); rows with
// id == 0 (padding_idx, sasrec/model.py:18) contribute nothing.
struct ScatterArgs {
  const int* ids; const float* G; int ldg; const float* rowscale;
  int T, d; float scale;
  DropCfg drop; uint32_t row_offset;
  float* dE;
  int nrep; size_t rep_stride;   // nrep > 1: dE is a set of replicas (rep_stride floats apart); wave -> replica wave % nrep.
                                 // Popular items receive thousands of adds per step; same-address float atomics
                                 // serialise, replicas divide that chain by nrep (k_replica_reduce sums them afterwards)
};

__global__ __launch_bounds__(256) void k_item_scatter(ScatterArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t key = drop_key(a.drop);
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float* dE = a.dE + (a.nrep > 1 ? (size_t)(wave % a.nrep) * a.rep_stride : 0);
  for (int row = wave; row < a.T; row += nwaves) {
    const int id = a.ids[row];
    if (id == 0) continue;
    const float rs = (a.rowscale ? a.rowscale[row] : 1.0f) * a.scale;
    if (rs == 0.f) continue;
    for (int c = lane; c < a.d; c += 64) {
      float v = a.G[(size_t)row * a.ldg + c] * rs;
      if (a.drop.thr) v = adt_keep(key, (uint32_t)(row + a.row_offset) * (uint32_t)a.d + (uint32_t)c, a.drop.thr) ? v * a.drop.scale : 0.f;
      atomicAdd(dE + (size_t)id * a.d + c, v);
    }
  }
}

// dst[i] += sum_r rep[r * stride + i]  (float4 granularity; n multiple of 4)
// sum of the nrep replica values of one float4 slot.  Eight loads are in flight at a time: with the plain `for (r < nrep) s += rep[r]` loop
// (runtime trip count, not unrolled) every load was its own round trip -- 16 serial round trips made the 20 MB fold of a training step an
// 11.6 us kernel (profiles/r03_kernel_stats.csv).
__device__ __forceinline__ float4 replica_sum4(const float* rep, size_t stride, int nrep, float4 s) {
  int r = 0;
  for (; r + 8 <= nrep; r += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(rep + (size_t)(r + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; r < nrep; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(rep + (size_t)r * stride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  return s;
}
__global__ __launch_bounds__(256) void k_replica_reduce(float* dst, const float* rep, size_t n, int nrep, size_t stride) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
    *reinterpret_cast<float4*>(dst + i) = replica_sum4(rep + i, stride, nrep, *reinterpret_cast<const float4*>(dst + i));
  }
}

// two replica folds in one launch (the item table and the other parameters at the end of a backward phase): blocks [0, g0) take job 0
struct RepReduce2Args { float* dst[2]; const float* rep[2]; size_t n[2]; int nrep[2]; size_t stride[2]; int g0; };
__global__ __launch_bounds__(256) void k_replica_reduce2(RepReduce2Args a) {
  const int j = (int)blockIdx.x >= a.g0 ? 1 : 0;
  const int bid = j ? blockIdx.x - a.g0 : blockIdx.x, nblk = j ? gridDim.x - a.g0 : a.g0;
  float* dst = a.dst[j];
  const float* rep = a.rep[j];
  const size_t n = a.n[j], stride = a.stride[j];
  const int nrep = a.nrep[j];
  for (size_t i = ((size_t)bid * 256 + threadIdx.x) * 4; i < n; i += (size_t)nblk * 1024) {
    *reinterpret_cast<float4*>(dst + i) = replica_sum4(rep + i, stride, nrep, *reinterpret_cast<const float4*>(dst + i));
  }
}

// dP[l] += sum_b dX[b, l] * keep/(1-p) * (ids != 0).  grid: (blocks over L*d/4 columns, slices over the
// batch); each thread sums its batch slice for one (l, 4 columns) in registers, then one atomic per column.
__global__ __launch_bounds__(256) void k_posemb_bwd(EmbedArgs a) {
  const int V = a.d / 4;
  const uint32_t key = drop_key(a.drop);
  const int B = a.T / a.L;
  const int lc = blockIdx.x * 256 + threadIdx.x;   // (l, c4)
  if (lc >= a.L * V) return;
  const int l = lc / V, c4 = (lc % V) * 4;
  const int bper = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = b0; b < b1; ++b) {
    const int row = b * a.L + l;
    if (a.ids[row] == 0) continue;
    float v[4];
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(a.dX + (size_t)row * a.d + c4);
    if (a.drop.thr) {
      const uint32_t base = (uint32_t)(row + a.row_offset) * (uint32_t)a.d + (uint32_t)c4;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(a.dP + (size_t)l * a.d + c4 + j, acc[j]);
}

// ---------------------------------------------------------------------------------------------
// Independence head classifier: SparseInputLinear(hd -> H) + log_softmax over the H outputs
// (sasrec/modules.py:648-649,679-703).  One thread per (token, head).  rec is written in the REFERENCE's
// row order: row l*B + b holds token (b, l) (the .view() at sasrec/modules.py:518 reinterprets (L,B,E)).
struct HeadClsArgs {
  const float* O; int ldo;    // T x d attention output (pre out_proj)
  const float* Ws; const float* bs;   // H x hd, H
  int B, L, H, hd;
  float* rec;                 // (L*B) x H x H log-probabilities
  const float* drec;          // bwd: same layout
  float* dO; int lddo;        // bwd: dO += dz Ws
  float* dWs; float* dbs;     // bwd: atomics
};

constexpr int MAXH = 8;

__global__ __launch_bounds__(256) void k_headcls_fwd(HeadClsArgs a) {
  const int n = a.B * a.L * a.H;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int tok = i / a.H, h = i % a.H;
    const int b = tok / a.L, l = tok % a.L;
    const float* o = a.O + (size_t)tok * a.ldo + h * a.hd;
    float z[MAXH];
#pragma unroll
    for (int cc = 0; cc < MAXH; ++cc) z[cc] = (cc < a.H) ? a.bs[cc] : -INFINITY;
    for (int j = 0; j < a.hd; j += 4) {
      const float4 ov = *reinterpret_cast<const float4*>(o + j);
#pragma unroll
      for (int cc = 0; cc < MAXH; ++cc)
        if (cc < a.H) {
          const float4 wv = *reinterpret_cast<const float4*>(a.Ws + cc * a.hd + j);
          z[cc] += ov.x * wv.x + ov.y * wv.y + ov.z * wv.z + ov.w * wv.w;
        }
    }
    float m = z[0];
#pragma unroll
    for (int cc = 1; cc < MAXH; ++cc) m = fmaxf(m, z[cc]);
    float s = 0.f;
#pragma unroll
    for (int cc = 0; cc < MAXH; ++cc) s += (cc < a.H) ? expf(z[cc] - m) : 0.f;
    const float lz = m + logf(s);
    float* dst = a.rec + ((size_t)(l * a.B + b) * a.H + h) * a.H;
#pragma unroll
    for (int cc = 0; cc < MAXH; ++cc)
      if (cc < a.H) dst[cc] = z[cc] - lz;
  }
}

// dz = drec - softmax(z) * sum(drec) with softmax = exp(rec);  dO[h*hd + j] += sum_c dz[c] Ws[c][j];
// dWs[c][j] += dz[c] o[j]; dbs[c] += dz[c].  One wave per token, lane = column of o (coalesced 256-B rows);
// each lane keeps its H partial sums of dWs[:, j] in registers across tokens, reduced through LDS at the end.
__global__ __launch_bounds__(256) void k_headcls_bwd(HeadClsArgs a) {
  extern __shared__ float sacc[];   // H*hd + H
  const int nW = a.H * a.hd, d = a.H * a.hd;
  for (int i = threadIdx.x; i < nW + a.H; i += 256) sacc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int T = a.B * a.L;
  for (int c0 = 0; c0 < d; c0 += 64) {     // column chunk (d = 64: one pass)
    const int col = c0 + lane;
    const bool live = col < d;
    const int h = live ? col / a.hd : 0, j = live ? col % a.hd : 0;
    float wreg[MAXH], acc[MAXH], accb[MAXH];
#pragma unroll
    for (int cc = 0; cc < MAXH; ++cc) {
      wreg[cc] = (live && cc < a.H) ? a.Ws[cc * a.hd + j] : 0.f;
      acc[cc] = 0.f;
      accb[cc] = 0.f;
    }
    // TU tokens per iteration: all of their loads are issued before the first dependent use (the dO read-modify-write would
    // otherwise serialise one memory latency per token)
    constexpr int TU = 4;
    for (int tok0 = wave * TU; tok0 < T; tok0 += nwaves * TU) {
      float dzr[TU][MAXH], rc[TU][MAXH], ovv[TU], dold[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int tok = tok0 + u;
        const bool ok = live && tok < T;
        const int b = ok ? tok / a.L : 0, l = ok ? tok % a.L : 0;
        const size_t ro = ((size_t)(l * a.B + b) * a.H + h) * a.H;
#pragma unroll
        for (int cc = 0; cc < MAXH; ++cc) {
          dzr[u][cc] = (ok && cc < a.H) ? a.drec[ro + cc] : 0.f;
          rc[u][cc] = (ok && cc < a.H) ? a.rec[ro + cc] : -INFINITY;
        }
        ovv[u] = ok ? a.O[(size_t)tok * a.ldo + col] : 0.f;
        dold[u] = ok ? a.dO[(size_t)tok * a.lddo + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int tok = tok0 + u;
        float sd = 0.f;
#pragma unroll
        for (int cc = 0; cc < MAXH; ++cc) sd += dzr[u][cc];
        float dov = 0.f;
#pragma unroll
        for (int cc = 0; cc < MAXH; ++cc)
          if (cc < a.H) {
            const float dz = dzr[u][cc] - expf(rc[u][cc]) * sd;
            dov += dz * wreg[cc];
            acc[cc] += dz * ovv[u];
            if (j == 0) accb[cc] += dz;
          }
        if (live && tok < T) a.dO[(size_t)tok * a.lddo + col] = dold[u] + dov;
      }
    }
    if (live) {
#pragma unroll
      for (int cc = 0; cc < MAXH; ++cc)
        if (cc < a.H) {
          atomicAdd(&sacc[cc * a.hd + j], acc[cc]);
          if (j == 0) atomicAdd(&sacc[nW + cc], accb[cc]);
        }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nW; i += 256) atomicAdd(a.dWs + i, sacc[i]);
  for (int i = threadIdx.x; i < a.H; i += 256) atomicAdd(a.dbs + i, sacc[nW + i]);
}

// ---------------------------------------------------------------------------------------------
// sasrec/model.py:72-76: pos/neg logits = sum_d f * E[pos|neg]; 16 lanes per token.
struct LogitsArgs {
  const float* F; int ldf;     // T x d (post last_layernorm)
  const float* E;
  const int* pos; const int* neg;
  int T, d;
  float* pos_logits; float* neg_logits;   // T
  const float* dpos; const float* dneg;   // bwd: T
  float* dF; int lddf;                    // bwd out (overwritten)
  float* dE;                              // bwd atomics
};

__global__ __launch_bounds__(256) void k_logits_fwd(LogitsArgs a) {
  const int sub = threadIdx.x & 15;
  for (int row = blockIdx.x * 16 + (threadIdx.x >> 4); row < a.T; row += gridDim.x * 16) {
    const int ip = a.pos[row], in = a.neg[row];
    float sp = 0.f, sn = 0.f;
    for (int c4 = 4 * sub; c4 < a.d; c4 += 64) {
      const float4 f = *reinterpret_cast<const float4*>(a.F + (size_t)row * a.ldf + c4);
      const float4 p = *reinterpret_cast<const float4*>(a.E + (size_t)ip * a.d + c4);
      const float4 q = *reinterpret_cast<const float4*>(a.E + (size_t)in * a.d + c4);
      sp += f.x * p.x + f.y * p.y + f.z * p.z + f.w * p.w;
      sn += f.x * q.x + f.y * q.y + f.z * q.z + f.w * q.w;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sn += __shfl_xor(sn, o, 64); }
    if (sub == 0) { a.pos_logits[row] = sp; a.neg_logits[row] = sn; }
  }
}

__global__ __launch_bounds__(256) void k_logits_bwd(LogitsArgs a) {
  // dF = dpos * E[pos] + dneg * E[neg]; the item-table side goes through k_item_scatter
  const int sub = threadIdx.x & 15;
  for (int row = blockIdx.x * 16 + (threadIdx.x >> 4); row < a.T; row += gridDim.x * 16) {
    const int ip = a.pos[row], in = a.neg[row];
    const float gp = a.dpos[row], gn = a.dneg[row];
    for (int c4 = 4 * sub; c4 < a.d; c4 += 64) {
      const float4 p = *reinterpret_cast<const float4*>(a.E + (size_t)ip * a.d + c4);
      const float4 q = *reinterpret_cast<const float4*>(a.E + (size_t)in * a.d + c4);
      float4 df;
      df.x = gp * p.x + gn * q.x; df.y = gp * p.y + gn * q.y; df.z = gp * p.z + gn * q.z; df.w = gp * p.w + gn * q.w;
      *reinterpret_cast<float4*>(a.dF + (size_t)row * a.lddf + c4) = df;
    }
  }
}

// d log_feats and the item-table rows of the positive / negative items in ONE pass over the tokens (k_logits_bwd + two k_item_scatter
// launches re-read F and the ids three times): dF = dpos E[pos] + dneg E[neg]; rep[pos] += dpos F; rep[neg] += dneg F.  One wave per
// token row (every atomic wave-instruction covers 256 contiguous bytes), replica = wave % nrep as in k_item_scatter; same arithmetic.
struct LogitsScatterArgs {
  const float* F; int ldf; const float* E; const int* pos; const int* neg; const float* dpos; const float* dneg;
  int T, d; float* dF; int lddf; float* rep; int nrep; size_t rep_stride;
};
__global__ __launch_bounds__(256) void k_logits_bwd_scatter(LogitsScatterArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float* dE = a.rep + (a.nrep > 1 ? (size_t)(wave % a.nrep) * a.rep_stride : 0);
  for (int row = wave; row < a.T; row += nwaves) {
    const int ip = a.pos[row], in = a.neg[row];
    const float gp = a.dpos[row], gn = a.dneg[row];
    for (int c = lane; c < a.d; c += 64) {
      const float p = a.E[(size_t)ip * a.d + c], q = a.E[(size_t)in * a.d + c], f = a.F[(size_t)row * a.ldf + c];
      a.dF[(size_t)row * a.lddf + c] = gp * p + gn * q;
      if (ip != 0 && gp != 0.f) atomicAdd(dE + (size_t)ip * a.d + c, f * gp);
      if (in != 0 && gn != 0.f) atomicAdd(dE + (size_t)in * a.d + c, f * gn);
    }
  }
}

// The same pass with the logits and the BCE seed formed HERE (d = 64): pos / neg logits = f . E[pos | neg] (sasrec/model.py:72-76), BCE-with-logits
// over the positions with pos != 0 and its derivatives (sasrec/main.py:151-153; bce_body above), then dF and the item rows as in
// k_logits_bwd_scatter.  The rows E[pos], E[neg] and f are gathered by this kernel anyway, so the forward needs no logits kernel and the
// loss assembly no BCE pass.  One wave per token row, lane = feature; the two dot products are wave sums.
struct LogitsBceArgs {
  const float* F; const float* E; const int* pos; const int* neg; const float* norms; int T;
  float* pos_logits; float* neg_logits; float* dpos; float* dneg; float* loss;      // loss: 2 x 64 sub-slots (pos term, neg term)
  float* dF; float* rep; int nrep; size_t rep_stride;
  int neg_only;              // the item rows of the NEGATIVE ids only: the positive ones are added by k_embed_bwd3 with the embedding rows they share
};
ADT_DEVICE_INLINE void logits_bce_body(const LogitsBceArgs& a, const int bid, const int nblk) {
  const int lane = threadIdx.x & 63;
  const int wave = bid * 4 + (threadIdx.x >> 6), nwaves = nblk * 4;
  float* dE = a.rep ? a.rep + (a.nrep > 1 ? (size_t)(wave % a.nrep) * a.rep_stride : 0) : nullptr;
  const float inv = 1.0f / a.norms[0];
  float lp = 0.f, ln = 0.f;
  // a wave owns R <= 64 CONSECUTIVE rows and keeps their four per-row scalars in lanes 0 .. R-1: one coalesced store per array at the end.
  // (One 4-byte store per row and array from waves on eight XCDs made every 64-byte line of those arrays sixteen partial writes: 68 us
  // against 23 for the same pass without them.)
  const int R = (a.T + nwaves - 1) / nwaves, r0 = wave * R, r1 = min(a.T, r0 + R);
  float keep_sp = 0.f, keep_sn = 0.f, keep_gp = 0.f, keep_gn = 0.f;
  // EIGHT rows in flight per wave, their ids from one coalesced load per 64 rows (lane = row) + v_readlane: with two rows and an id load per
  // iteration a wave's ~13 rows were 7 iterations of two dependent round trips each -- the kernel is latency, not bandwidth (20 us alone,
  // 33-46 beside anything else).
  constexpr int U = 8;
  int ipv = 0, inv_ids = 0;
  for (int row0 = r0; row0 < r1; row0 += U) {
    if (((row0 - r0) & 63) == 0) {              // wave-uniform: the ids of the next (up to) 64 rows
      const int rr = row0 + lane;
      ipv = rr < r1 ? a.pos[rr] : 0;
      inv_ids = rr < r1 ? a.neg[rr] : 0;
    }
    int ip[U], in[U];
    float p[U], q[U], f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (row0 + u - r0) & 63;      // (U divides 64: a batch never straddles two id loads)
      ip[u] = __builtin_amdgcn_readlane(ipv, k);
      in[u] = __builtin_amdgcn_readlane(inv_ids, k);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = row0 + u;
      p[u] = a.E[(size_t)ip[u] * 64 + lane];
      q[u] = a.E[(size_t)in[u] * 64 + lane];
      f[u] = row < r1 ? a.F[(size_t)row * 64 + lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = row0 + u;
      if (row >= r1) break;                     // wave-uniform
      const float sp = wave_sum_valu(f[u] * p[u]), sn = wave_sum_valu(f[u] * q[u]);
      float gp = 0.f, gn = 0.f;
      if (ip[u] != 0) {                         // BCEWithLogits: target 1 -> softplus(-x), target 0 -> softplus(x)   (bce_body above)
        const float ep = __expf(-fabsf(sp)), en = __expf(-fabsf(sn));
        lp += fmaxf(-sp, 0.f) + __logf(1.0f + ep);
        ln += fmaxf(sn, 0.f) + __logf(1.0f + en);
        const float sgp = sp >= 0.f ? 1.0f / (1.0f + ep) : ep / (1.0f + ep);      // sigmoid(x) from exp(-|x|)
        const float sgn = sn >= 0.f ? 1.0f / (1.0f + en) : en / (1.0f + en);
        gp = (sgp - 1.0f) * inv;
        gn = sgn * inv;
      }
      if (lane == ((row - r0) & 63)) { keep_sp = sp; keep_sn = sn; keep_gp = gp; keep_gn = gn; }
      if (((row - r0) & 63) == 63 || row == r1 - 1) {      // wave-uniform: flush the scalars of up to 64 rows
        const int base = r0 + ((row - r0) & ~63);
        if (base + lane <= row) {
          a.pos_logits[base + lane] = keep_sp; a.neg_logits[base + lane] = keep_sn; a.dpos[base + lane] = keep_gp; a.dneg[base + lane] = keep_gn;
        }
      }
      a.dF[(size_t)row * 64 + lane] = gp * p[u] + gn * q[u];
      if (a.rep == nullptr) continue;           // the item rows are summed elsewhere (adt_itemgrad.cuh: sorted segments)
      if (!a.neg_only && ip[u] != 0 && gp != 0.f) atomicAdd(dE + (size_t)ip[u] * 64 + lane, f[u] * gp);
      if (in[u] != 0 && gn != 0.f) atomicAdd(dE + (size_t)in[u] * 64 + lane, f[u] * gn);
    }
  }
  // one atomic per term and BLOCK: the 64 sub-slots of a term share four 64-byte lines, and float atomics to one line are processed one
  // after the other at the memory side (~20 ns each): 8,192 waves x 2 single-lane atomics were 43 of this kernel's 69 us
  __shared__ float sl[8];
  if (lane == 0) { sl[threadIdx.x >> 6] = lp; sl[4 + (threadIdx.x >> 6)] = ln; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(a.loss + (bid & 63), (sl[0] + sl[1] + sl[2] + sl[3]) * inv);
    atomicAdd(a.loss + 64 + (bid & 63), (sl[4] + sl[5] + sl[6] + sl[7]) * inv);
  }
}
__global__ __launch_bounds__(256) void k_logits_bce_scatter(LogitsBceArgs a) { logits_bce_body(a, blockIdx.x, gridDim.x); }

// Embedding backward, d = 64, one pass over dX (k_posemb_bwd + k_item_scatter read it twice): a wave owns position l and a slice of the
// batch; per row it adds dX * sqrt(d) * keep/(1-p) to the item replica row and keeps the positional sum in a register (one atomic per
// lane at the end).  sasrec/model.py:34-41 / :53-59 reversed.
struct EmbedBwdArgs {
  const int* ids; const float* dX; int T, L; float scale; DropCfg drop; uint32_t row_offset;
  float* dP; float* rep; int nrep; size_t rep_stride; int nslices;
};
__global__ __launch_bounds__(256) void k_embed_bwd64(EmbedBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= a.L * a.nslices) return;
  const uint32_t key = drop_key(a.drop);
  const int l = wave % a.L, s = wave / a.L, B = a.T / a.L;
  const int bper = (B + a.nslices - 1) / a.nslices, b0 = s * bper, b1 = min(B, b0 + bper);
  float* dE = a.rep + (a.nrep > 1 ? (size_t)(wave % a.nrep) * a.rep_stride : 0);
  float acc = 0.f;
  for (int b = b0; b < b1; b += 4) {                      // four rows in flight: ids, then the rows, then the atomics
    int id[4];
    float g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) id[u] = b + u < b1 ? a.ids[(b + u) * a.L + l] : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = id[u] != 0 ? a.dX[(size_t)((b + u) * a.L + l) * 64 + lane] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (id[u] == 0) continue;
      const int row = (b + u) * a.L + l;
      float vi = g[u] * a.scale, vp = g[u];
      if (a.drop.thr) {
        const bool keep = adt_keep(key, (uint32_t)(row + a.row_offset) * 64u + (uint32_t)lane, a.drop.thr);
        vi = keep ? vi * a.drop.scale : 0.f;
        vp = keep ? vp * a.drop.scale : 0.f;
      }
      atomicAdd(dE + (size_t)id[u] * 64 + lane, vi);
      acc += vp;
    }
  }
  atomicAdd(a.dP + (size_t)l * 64 + lane, acc);
}

// The three scatters into the item table whose ids are SHIFTS of one item list, in one pass (d = 64).  The reference's sampler builds
// seq = items[:-1], pos = items[1:] and the decoder input dec[:, 1:] = seq[:, :-1] (sasrec/utils.py:288-307), so for almost every token
//   row seq[b, l] receives (A) the encoder embedding gradient of (b, l), (B) the decoder embedding gradient of (b, l + 1) and
//   (C) the positive-logit row coef_pos[b, l - 1] * log_feats[b, l - 1].
// Float atomics run at one chip-wide byte rate (~1.3 TB/s: MI355X_MICROARCH.md, Global float atomics) and the four scatters of a step were
// 52 MB of them; here the wave that owns the anchor token (b, l) adds A + B + C with ONE atomic row-add when the ids agree -- checked per
// token, so arbitrary ids stay correct: a B or C whose id does not match its anchor is an "orphan" and is added on its own by the wave of
// its own (b, l).  Every contribution is added exactly once: B(b, l') joins anchor (b, l' - 1) iff l' >= 1 and dec[b, l'] == seq[b, l' - 1],
// else it is the orphan of (b, l'); C(b, l') joins anchor (b, l' + 1) iff l' + 1 < L and pos[b, l'] == seq[b, l' + 1], else orphan of (b, l').
// Positional table: the wave keeps the sums of position l (A and orphan B) and l + 1 (joined B) in registers, one atomic per lane each.
// (sasrec/model.py:34-41, :53-59, :72-76 reversed.)
static __device__ __attribute__((aligned(256))) const float eb_zero_row[64] = {};
struct EmbedBwd3Args {
  const int* seq; const int* dec; const int* pos;
  const float* dXs; const float* dXd; const float* F; const float* gp;      // d / d (encoder / decoder embedding output), log_feats, d / d pos_logits
  int T, L; float scale; DropCfg drop_s, drop_d; uint32_t row_offset;
  float* dP; float* rep; int nrep; size_t rep_stride; int nslices;
};
ADT_DEVICE_INLINE void embed_bwd3_body(const EmbedBwd3Args& a, const int bid) {
  typedef const float __attribute__((address_space(1))) * gf;
  const int lane = threadIdx.x & 63;
  const int wave = bid * 4 + (threadIdx.x >> 6);
  if (wave >= a.L * a.nslices) return;
  const uint32_t key_s = drop_key(a.drop_s), key_d = drop_key(a.drop_d);
  const int l = wave % a.L, s = wave / a.L, B = a.T / a.L, L = a.L;
  const int bper = (B + a.nslices - 1) / a.nslices, b0 = s * bper, b1 = min(B, b0 + bper);
  float* dE = a.rep + (a.nrep > 1 ? (size_t)(wave % a.nrep) * a.rep_stride : 0);
  float acc0 = 0.f, acc1 = 0.f;                 // positional sums of positions l and l + 1
  const bool has_prev = l > 0, has_next = l + 1 < L;
  constexpr int U = 2;                          // anchors in flight: seven ids, then up to five rows each
  for (int b = b0; b < b1; b += U) {
    int s0[U], sp[U], sn[U], d0[U], dn[U], p0[U], pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool on = b + u < b1;
      const int row = (b + u) * L + l;
      s0[u] = on ? a.seq[row] : 0;
      sp[u] = on && has_prev ? a.seq[row - 1] : 0;
      sn[u] = on && has_next ? a.seq[row + 1] : 0;
      d0[u] = on ? a.dec[row] : 0;
      dn[u] = on && has_next ? a.dec[row + 1] : 0;
      p0[u] = on ? a.pos[row] : 0;
      pp[u] = on && has_prev ? a.pos[row - 1] : 0;
    }
    bool fA[U], fBm[U], fBo[U], fCm[U], fCo[U];
    float ga[U], gbn[U], gb0[U], fprev[U], f0[U], cprev[U], c0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {               // every load unconditional (address select): all in flight together
      const size_t row = (size_t)((b + u) * L + l);
      fA[u] = s0[u] != 0;
      fBm[u] = dn[u] != 0 && dn[u] == s0[u];
      fBo[u] = d0[u] != 0 && (!has_prev || d0[u] != sp[u]);
      fCm[u] = has_prev && pp[u] != 0 && pp[u] == s0[u];
      fCo[u] = p0[u] != 0 && (!has_next || p0[u] != sn[u]);
      ga[u] = *((fA[u] ? (gf)(a.dXs + row * 64) : (gf)eb_zero_row) + lane);
      gbn[u] = *((fBm[u] ? (gf)(a.dXd + (row + 1) * 64) : (gf)eb_zero_row) + lane);
      gb0[u] = *((fBo[u] ? (gf)(a.dXd + row * 64) : (gf)eb_zero_row) + lane);
      fprev[u] = *((fCm[u] ? (gf)(a.F + (row - 1) * 64) : (gf)eb_zero_row) + lane);
      f0[u] = *((fCo[u] ? (gf)(a.F + row * 64) : (gf)eb_zero_row) + lane);
      cprev[u] = *(fCm[u] ? (gf)(a.gp + row - 1) : (gf)eb_zero_row);
      c0[u] = *(fCo[u] ? (gf)(a.gp + row) : (gf)eb_zero_row);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row = (uint32_t)((b + u) * L + l) + a.row_offset;
      float va = ga[u], vbn = gbn[u], vb0 = gb0[u];
      if (a.drop_s.thr) {
        va = adt_keep(key_s, row * 64u + (uint32_t)lane, a.drop_s.thr) ? va * a.drop_s.scale : 0.f;
        vbn = adt_keep(key_d, (row + 1u) * 64u + (uint32_t)lane, a.drop_d.thr) ? vbn * a.drop_d.scale : 0.f;
        vb0 = adt_keep(key_d, row * 64u + (uint32_t)lane, a.drop_d.thr) ? vb0 * a.drop_d.scale : 0.f;
      }
      acc0 += va + vb0;
      acc1 += vbn;
      if (fA[u]) atomicAdd(dE + (size_t)s0[u] * 64 + lane, (va + vbn) * a.scale + fprev[u] * cprev[u]);
      if (fBo[u]) atomicAdd(dE + (size_t)d0[u] * 64 + lane, vb0 * a.scale);
      if (fCo[u] && c0[u] != 0.f) atomicAdd(dE + (size_t)p0[u] * 64 + lane, f0[u] * c0[u]);
    }
  }
  atomicAdd(a.dP + (size_t)l * 64 + lane, acc0);
  if (has_next) atomicAdd(a.dP + (size_t)(l + 1) * 64 + lane, acc1);
}
__global__ __launch_bounds__(256) void k_embed_bwd3(EmbedBwd3Args a) { embed_bwd3_body(a, blockIdx.x); }

// ---------------------------------------------------------------------------------------------
// Loss seeds of sasrec/main.py:151-169.  `norms` is a device array {n_bce, n_mse, n_nll} holding the
// GLOBAL normalisers (data-parallel exactness).  loss slots: [0] bce_pos [1] bce_neg
// [2..2+nl) mse_i [2+nl..2+2nl) nll_l  (sums already divided by their normalisers).
struct BceArgs {
  const float* pos_logits; const float* neg_logits; const int* pos;
  int T;
  const float* norms;
  float* dpos; float* dneg;
  float* loss;
};

ADT_DEVICE_INLINE float block_sum(float v, float* sbuf) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = sbuf[0] + sbuf[1] + sbuf[2] + sbuf[3];
  __syncthreads();
  return r;  // valid on thread 0
}

ADT_DEVICE_INLINE void bce_body(const BceArgs& a, int bid, int nblk, float* sbuf) {
  const float inv = 1.0f / a.norms[0];
  float lp = 0.f, ln = 0.f;
  for (int i = bid * 256 + threadIdx.x; i < a.T; i += nblk * 256) {
    float gp = 0.f, gn = 0.f;
    if (a.pos[i] != 0) {
      const float xp = a.pos_logits[i], xn = a.neg_logits[i];
      // BCEWithLogits: target 1 -> softplus(-x), target 0 -> softplus(x)
      lp += fmaxf(-xp, 0.f) + log1pf(expf(-fabsf(xp)));
      ln += fmaxf(xn, 0.f) + log1pf(expf(-fabsf(xn)));
      gp = (1.0f / (1.0f + expf(-xp)) - 1.0f) * inv;
      gn = (1.0f / (1.0f + expf(-xn))) * inv;
    }
    a.dpos[i] = gp;
    a.dneg[i] = gn;
  }
  const float sp = block_sum(lp, sbuf);
  const float sn = block_sum(ln, sbuf);
  // 64 sub-slots per loss term: same-address float atomics serialise (~10 ns each), thousands of blocks would cost
  // tens of microseconds on one address
  if (threadIdx.x == 0) { atomicAdd(a.loss + (bid & 63), sp * inv); atomicAdd(a.loss + 64 + (bid & 63), sn * inv); }
}
__global__ __launch_bounds__(256) void k_bce(BceArgs a) {
  __shared__ float sbuf[4];
  bce_body(a, blockIdx.x, gridDim.x, sbuf);
}

// g = 2*lambda/n_mse * (A - Bm);  GA += g (or = g), GB = -g;  loss += sum (A-B)^2 / n_mse
struct MseArgs {
  const float* A; const float* Bm;
  size_t n;                  // elements (multiple of 4)
  float lambda;
  const float* norms;
  float* GA; int accA;       // accA: GA += g
  float* GB;
  float* loss;               // one slot
  float* coef_out;           // optional: 2 lambda / n_mse for the consumers that form the seed themselves (GA / GB == nullptr)
};

ADT_DEVICE_INLINE void mse_body(const MseArgs& a, int bid, int nblk, float* sbuf) {
  const float inv = 1.0f / a.norms[1];
  const float coef = 2.0f * a.lambda * inv;
  if (a.coef_out && bid == 0 && threadIdx.x == 0) *a.coef_out = coef;
  if (!a.GA && !a.GB) return;      // neither seed is materialised: the consumer of the pair also adds its loss term (SeqBwdArgs::seed_loss)
  float acc = 0.f;
  for (size_t i = ((size_t)bid * 256 + threadIdx.x) * 4; i < a.n; i += (size_t)nblk * 1024) {
    const float4 x = *reinterpret_cast<const float4*>(a.A + i);
    const float4 y = *reinterpret_cast<const float4*>(a.Bm + i);
    float4 dlt = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
    acc += dlt.x * dlt.x + dlt.y * dlt.y + dlt.z * dlt.z + dlt.w * dlt.w;
    float4 g = make_float4(coef * dlt.x, coef * dlt.y, coef * dlt.z, coef * dlt.w);
    // GA / GB == nullptr: that seed is not materialised -- its consumer (k_seqtt_attn_pre_bwd: SeqBwdArgs::seed_other) forms it from the two rows
    if (a.GB) *reinterpret_cast<float4*>(a.GB + i) = make_float4(-g.x, -g.y, -g.z, -g.w);
    if (a.GA) {
      if (a.accA) {
        const float4 o = *reinterpret_cast<const float4*>(a.GA + i);
        g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
      }
      *reinterpret_cast<float4*>(a.GA + i) = g;
    }
  }
  const float s = block_sum(acc, sbuf);
  if (threadIdx.x == 0) atomicAdd(a.loss + (bid & 63), s * inv);
}
__global__ __launch_bounds__(256) void k_mse_seed(MseArgs a) {
  __shared__ float sbuf[4];
  mse_body(a, blockIdx.x, gridDim.x, sbuf);
}

// drec[n][h][c] = -(lambda2 / n_nll) * [h == c];  loss += -sum_n,h rec[n][h][h] / n_nll
struct NllArgs {
  const float* rec; int n_rows; int H;
  float lambda2;
  const float* norms;
  float* drec;
  float* loss;
};

ADT_DEVICE_INLINE void nll_body(const NllArgs& a, int bid, int nblk, float* sbuf) {
  const float inv = 1.0f / a.norms[2];
  const int HH = a.H * a.H;
  const int n = a.n_rows * HH;
  float acc = 0.f;
  for (int i = bid * 256 + threadIdx.x; i < n; i += nblk * 256) {
    const int e = i % HH;
    const bool diag = (e / a.H) == (e % a.H);
    if (diag) acc -= a.rec[i];
    a.drec[i] = diag ? -a.lambda2 * inv : 0.f;
  }
  const float s = block_sum(acc, sbuf);
  if (threadIdx.x == 0) atomicAdd(a.loss + (bid & 63), s * inv);
}
__global__ __launch_bounds__(256) void k_nll_seed(NllArgs a) {
  __shared__ float sbuf[4];
  nll_body(a, blockIdx.x, gridDim.x, sbuf);
}

// All loss seeds of a step (sasrec/main.py:151-169) in one launch: the BCE seed, up to four reconstruction seeds and up to four independence
// seeds are independent streaming passes; as five launches of 5-11 us each they cost 39 us at the flagship shape.  Blocks [0, gb) run the
// BCE body, the next nmse * gm the MSE bodies, the last nnll * gn the NLL bodies (same arithmetic, same per-task block counts as the
// stand-alone kernels).
// Prefetch of the NEXT step's id batch (optional, the last `gp` workgroups of k_loss_seeds): the step's first kernel otherwise reads its batch
// from the pinned host ring over PCIe (~16 us for the flagship's 819 KB) before anything else of the step can start.  If the producer has
// already published the next batch (*produced > state[0], the count of batches fetched so far), these workgroups copy its slot into a
// device staging buffer while the loss assembly streams, mark it staged (state[2] = batch index + 1) and release the slot (*consumed);
// k_step_begin of the next step then copies the batch HBM -> HBM.  Not published yet: nothing happens, the next step reads the ring itself.
// state: [0] batches fetched, [1] k_step_begin's ticket, [2] staged batch + 1, [3] the prefetch ticket, [4] *produced as k_step_begin saw it,
// [5] statistics: batches k_step_begin took from the staging buffer.
struct RingPrefetchArgs {
  const int32_t* ring; size_t slot_ints; int nslots; size_t n_ints; uint32_t* state; uint32_t* consumed; int32_t* staging;
  int part, nparts;      // this launch copies 16-byte words [part, part + 1) * n / nparts of the slot; the LAST part marks the batch staged and releases the
                         // slot (the parts run in stream order).  0, 0 = the whole slot.  819 KB over PCIe took 34-39 us inside the 24 us loss launch:
                         // half there, half inside the 20 us embedding scatter is hidden in both.
};
ADT_DEVICE_INLINE void ring_prefetch_body(const RingPrefetchArgs& a, int bid, int nblk) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const uint32_t k = a.state[0];                      // index of the next batch (this step's k_step_begin has counted its own)
  if (a.state[4] <= k) return;                        // state[4]: the producer's count as k_step_begin of this step saw it -- the same answer in every workgroup
  const int32_t* slot = a.ring + (size_t)(k % (uint32_t)a.nslots) * a.slot_ints;
  const size_t nall = a.n_ints / 4, np = a.nparts > 1 ? (size_t)a.nparts : 1;
  const size_t lo = nall * (size_t)(a.nparts > 1 ? a.part : 0) / np, n16 = nall * (size_t)((a.nparts > 1 ? a.part : 0) + 1) / np;
  constexpr int RU = 8;
  for (size_t i0 = lo + (size_t)bid * 256 + threadIdx.x; i0 < n16; i0 += (size_t)RU * nblk * 256) {
    v4i v[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const size_t i = i0 + (size_t)u * nblk * 256;
      v[u] = i < n16 ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(slot) + i) : v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const size_t i = i0 + (size_t)u * nblk * 256;
      if (i < n16) reinterpret_cast<v4i*>(a.staging)[i] = v[u];
    }
  }
  if (a.nparts > 1 && a.part != a.nparts - 1) return;      // (workgroup-uniform) not the last part: nothing to mark
  __syncthreads();
  if (threadIdx.x == 0) {
    // no fences: the staging words and state[2] are read by the NEXT launch (a kernel boundary orders them), and the host only learns that
    // the slot has been READ -- a system-scope release here would write back every dirty L2 line of the running step first
    if (atomicAdd(a.state + 3, 1u) == (uint32_t)nblk - 1u) {
      a.state[3] = 0u;
      a.state[2] = k + 1u;
      if (a.consumed) __hip_atomic_store(a.consumed, k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// k_embed_bwd3 with the second half of the ring prefetch as its first gp workgroups
__global__ __launch_bounds__(256) void k_embed_bwd3_prefetch(EmbedBwd3Args a, RingPrefetchArgs pf, int gp) {
  if ((int)blockIdx.x < gp) { ring_prefetch_body(pf, blockIdx.x, gp); return; }
  embed_bwd3_body(a, (int)blockIdx.x - gp);
}

struct LossSeedsArgs {
  BceArgs bce; MseArgs mse[4]; NllArgs nll[4];
  int nmse, nnll, gb, gm, gn, gp;
  LogitsBceArgs lb; int gl;      // the first gl workgroups: the logits / BCE / item-row pass of the deferred path (k_logits_bce_scatter's body)
  RingPrefetchArgs pf;
};
__global__ __launch_bounds__(256) void k_loss_seeds(LossSeedsArgs a) {
  __shared__ float sbuf[4];
  int b = blockIdx.x;
  if (b < a.gp) { ring_prefetch_body(a.pf, b, a.gp); return; }   // the PCIe read first: it needs the whole launch to hide behind (as the last workgroups
  b -= a.gp;                                                     // it started late and stretched the launch 24 -> 39 us)
  if (b < a.gl) { logits_bce_body(a.lb, b, a.gl); return; }      // then the longest job
  b -= a.gl;
  if (b < a.gb) { bce_body(a.bce, b, a.gb, sbuf); return; }
  b -= a.gb;
  if (b < a.nmse * a.gm) { const int k = b / a.gm; mse_body(a.mse[k], b - k * a.gm, a.gm, sbuf); return; }
  b -= a.nmse * a.gm;
  if (b < a.nnll * a.gn) { const int k = b / a.gn; nll_body(a.nll[k], b - k * a.gn, a.gn, sbuf); return; }
}

// ---------------------------------------------------------------------------------------------
// sasrec/main.py:170-173: + wd * ||item_emb||_F (un-squared), clip_grad_norm_(clip), Adam(b1, b2).
struct OptArgs {
  float* P; float* G; float* M; float* Vv;
  size_t n;          // number of trainable floats (flat prefix of the parameter buffer)
  size_t nE;         // item table size (flat offset 0)
  float wd, clip, lr, b1, b2, eps;
  float* scal;       // device scalars: [0] ||E||^2  [1] ||g||^2  [2] step (float)  [3] wd loss term ; [64..128) partial
                     // sums of ||E||^2, [128..192) partial sums of ||g||^2 (64 sub-slots each: no same-address contention)
  float grad_scale;  // multiply grads first (1/world for averaged all-reduce; normally 1)
  float l2;          // Adam's coupled weight_decay (bert4rec/trainer.py:41): g += l2 * p AFTER clipping; 0 = off
  int fold_only;                 // k_fold_parts_gradnorm in front of a gradient all-reduce: the step count and the optimizer scalars are left to the kernels behind it
  float* gn_part; int gn_n;      // optional: ||g||^2 as gn_n per-block partials (stored by k_fold_parts_gradnorm, summed in order by k_adam) instead of
                                 // the 64 atomically added sub-slots scal[128..192): the norm, and with it the clip factor, has the same bits in every run
};

__global__ __launch_bounds__(256) void k_sumsq(const float* x, size_t n, float* out) {
  __shared__ float sbuf[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i] * x[i];
  const float s = block_sum(acc, sbuf);
  if (threadIdx.x == 0) atomicAdd(out + (blockIdx.x & 63), s);
}

// Everything a training step does before its forward, in one launch (the trainer used to issue five: seed += inc, normaliser copy, gradient
// fill, loss-slot fill, optimizer-scalar fill -- and a sixth, k_sumsq, later): G = 0; loss slots = 0; scal[128..192) = 0;
// scal[64 + b] = block b's share of ||E||^2 (stored, not added: no zero needed; the parameters do not change before adt_clip_adam_pre);
// norms_dst = norms_src; *seed += inc.  grid >= 66 blocks.
struct StepBeginArgs {
  uint32_t* seed; uint32_t inc; float* norms_dst; const float* norms_src; float* loss; int nloss; float* scal; float* G; size_t n;
  const float* E; size_t nE;
  // optional id ring (adt_sasrec_step_begin_ring): slot (state[0] % nslots) of `ring` -- nslots blocks of slot_ints int32 in pinned HOST
  // memory (read over PCIe by this kernel: no copy engine, no cross-queue wait between the copy and the step) or in HBM -- is copied to
  // ids_dst (n_ints int32, a multiple of 4; its last four words are the loss normalisers, which then replace norms_src).  The last block
  // to finish bumps state[0] and publishes the new count to *consumed (pinned host word the producer polls before it refills a slot).
  const int32_t* ring; size_t slot_ints; int nslots; int32_t* ids_dst; size_t n_ints; uint32_t* state; uint32_t* consumed;
  const int32_t* staging;            // optional: device copy of a prefetched batch, valid when state[2] == state[0] + 1 (ring_prefetch_body)
  const uint32_t* produced;          // optional: the producer's published batch count (pinned host word), sampled into state[4]
  // optional extras of the model-level step (adt_sasrec_step_begin*): a second range to zero (the parameter-gradient replicas the backward
  // chains flush into) and the bf16 weight images of the step (pk.n blocks, packed by the LAST pk.n workgroups of the grid: k_pack_wimg's
  // work without its launch)
  float* Z; size_t nz;
  PackArgs pk;
};
constexpr int SB_RING_BLOCKS = 32;      // workgroups that fetch the id batch: enough bytes in flight for a PCIe read (25 x 16 B per thread at the
                                        // flagship batch, eight at a time), few enough that their completion ticket is a short chain
__global__ __launch_bounds__(256) void k_step_begin(StepBeginArgs a) {
  __shared__ float sbuf[4];
  __shared__ __attribute__((aligned(16))) __bf16 spk[4 * WPACK_IMG];
  typedef int v4i __attribute__((ext_vector_type(4)));
  const unsigned nblk = gridDim.x - (unsigned)a.pk.n;      // the workgroups of the step's own work ; the rest pack weight images, one per block
  if (blockIdx.x >= nblk) { pack_wimg_block_lds(a.pk, (int)(blockIdx.x - nblk), spk); return; }
  const unsigned nring = a.ring ? (nblk < (unsigned)SB_RING_BLOCKS ? nblk : (unsigned)SB_RING_BLOCKS) : 0u;
  const bool ring_blk = blockIdx.x < nring;
  const int32_t* slot = nullptr;
  constexpr int RU = 8;
  v4i idv[RU];
  size_t n16 = 0;
  if (a.ring) {
    const uint32_t k = a.state[0];
    slot = (a.staging && a.state[2] == k + 1u) ? a.staging : a.ring + (size_t)(k % (uint32_t)a.nslots) * a.slot_ints;      // prefetched by the previous step?
  }
  if (ring_blk) {      // requests first: the PCIe round trips run under the zero-fill and the ||E||^2 sums below
    n16 = a.n_ints / 4;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x + (size_t)u * nring * 256;
      idv[u] = i < n16 ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(slot) + i) : v4i{0, 0, 0, 0};
    }
  }
  const size_t n4 = a.n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)nblk * 256)
    reinterpret_cast<float4*>(a.G)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && threadIdx.x < (int)(a.n - n4 * 4)) a.G[n4 * 4 + threadIdx.x] = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.nz / 4; i += (size_t)nblk * 256)      // nz: a multiple of 4
    reinterpret_cast<float4*>(a.Z)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned eb = blockIdx.x - nring;       // the small jobs go to the workgroups behind the ring readers (nblk >= nring + 66)
  if (blockIdx.x >= nring && eb < 64) {
    // 16-byte loads, four in flight (a scalar loop with a runtime trip count was 13 serial round trips at the ml-1m table: ~10 us)
    float acc = 0.f;
    const size_t nE4 = a.nE / 4, step = (size_t)64 * 256;
    size_t i = (size_t)eb * 256 + threadIdx.x;
    for (; i + 3 * step < nE4; i += 4 * step) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(a.E)[i + u * step];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
    }
    for (; i < nE4; i += step) {
      const float4 v = reinterpret_cast<const float4*>(a.E)[i];
      acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (eb == 0 && threadIdx.x < (int)(a.nE - nE4 * 4)) { const float t = a.E[nE4 * 4 + threadIdx.x]; acc += t * t; }
    const float s = block_sum(acc, sbuf);
    if (threadIdx.x == 0) a.scal[64 + eb] = s;
  } else if (eb == 64) {
    if (threadIdx.x < 64) a.scal[128 + threadIdx.x] = 0.f;
    for (int i = threadIdx.x; i < a.nloss; i += 256) a.loss[i] = 0.f;
  } else if (eb == 65) {
    const float* nsrc = slot ? reinterpret_cast<const float*>(slot + a.n_ints - 4) : a.norms_src;
    if (threadIdx.x < 4 && a.norms_dst) a.norms_dst[threadIdx.x] = nsrc[threadIdx.x];
    if (threadIdx.x == 4 && a.seed) *a.seed += a.inc;
    if (threadIdx.x == 5 && a.ring && a.produced) a.state[4] = __hip_atomic_load(a.produced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (ring_blk) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x + (size_t)u * nring * 256;
      if (i < n16) reinterpret_cast<v4i*>(a.ids_dst)[i] = idv[u];
    }
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x + (size_t)RU * nring * 256; i0 < n16; i0 += (size_t)RU * nring * 256) {      // larger batches
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const size_t i = i0 + (size_t)u * nring * 256;
        idv[u] = i < n16 ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(slot) + i) : v4i{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const size_t i = i0 + (size_t)u * nring * 256;
        if (i < n16) reinterpret_cast<v4i*>(a.ids_dst)[i] = idv[u];
      }
    }
    // every word this block reads has arrived in registers (it was stored): the block is done with the slot.  The ticket only counts
    // completed READS -- nothing another agent reads is published here, so no release fence (a __threadfence() here wrote back the L2 lines
    // this block's zero-fill had dirtied, ~6 us) -- and it is taken by the nring reading blocks only (256 tickets on one word were a
    // chain of 256 same-line atomics, ~5 us).
    __syncthreads();
    if (threadIdx.x == 0) {
      if (atomicAdd(a.state + 1, 1u) == nring - 1) {      // the last reader: every reader has read state[0] and its part of the slot
        a.state[1] = 0u;
        const uint32_t c = a.state[0] + 1u;
        if (slot == a.staging && a.staging) a.state[5] += 1u;      // statistics: batches taken from the staging buffer
        a.state[0] = c;
        // relaxed: the host only learns that the slot has been read.  (As a system-scope RELEASE this one store wrote back the ~11 MB of
        // zero-fill the kernel had just left dirty in L2: k_step_begin took 24-50 us on a host ring against 9.5 us on a device ring.)
        if (a.consumed) __hip_atomic_store(a.consumed, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

ADT_DEVICE_INLINE float sum64(const float* part, float* sbuf) {
  // every thread returns the sum of 64 partial slots
  float v = threadIdx.x < 64 ? part[threadIdx.x] : 0.f;
  v = wave_sum(v);
  if (threadIdx.x == 0) sbuf[0] = v;
  __syncthreads();
  const float r = sbuf[0];
  __syncthreads();
  return r;
}

// Both optimizer kernels are a few dependent memory round trips long (1.3 MB of parameters at the flagship shape): the operands of a
// thread's first element are requested BEFORE the reduction of the 64 partial sums, whose round trip they then share.
__global__ __launch_bounds__(256) void k_wd_gradnorm(OptArgs a) {
  __shared__ float sbuf[4];
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool first = i < a.n;
  float g0 = 0.f, p0 = 0.f;
  if (first) { g0 = a.G[i]; if (i < a.nE) p0 = a.P[i]; }
  const float nrm = sqrtf(sum64(a.scal + 64, sbuf));
  const float coef = (a.wd != 0.f && nrm > 0.f) ? a.wd / nrm : 0.f;
  float acc = 0.f;
  if (first) {
    float g = g0 * a.grad_scale;
    if (i < a.nE) g += coef * p0;
    a.G[i] = g;
    acc += g * g;
    i += stride;
  }
  for (; i < a.n; i += stride) {
    float g = a.G[i] * a.grad_scale;
    if (i < a.nE) g += coef * a.P[i];
    a.G[i] = g;
    acc += g * g;
  }
  const float s = block_sum(acc, sbuf);
  if (threadIdx.x == 0) atomicAdd(a.scal + 128 + (blockIdx.x & 63), s);
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.scal[0] = nrm * nrm; a.scal[3] = a.wd * nrm; a.scal[2] += 1.0f; }
}

__global__ __launch_bounds__(256) void k_adam(OptArgs a) {
  __shared__ float sbuf[4];
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool first = i < a.n;
  float g0 = 0.f, p0 = 0.f, m0 = 0.f, v0 = 0.f;
  if (first) { g0 = a.G[i]; p0 = a.P[i]; m0 = a.M[i]; v0 = a.Vv[i]; }
  const float t = a.scal[2];
  float gn2;
  if (a.gn_part) {      // ordered: thread t sums partials t, t + 256, ... ; then the block's fixed reduction tree
    float part = 0.f;
    for (int i = threadIdx.x; i < a.gn_n; i += 256) part += a.gn_part[i];
    const float bs = block_sum(part, sbuf);
    if (threadIdx.x == 0) sbuf[0] = bs;
    __syncthreads();
    gn2 = sbuf[0];
    __syncthreads();
  } else {
    gn2 = sum64(a.scal + 128, sbuf);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[1] = gn2;
  const float tn = sqrtf(gn2);
  const float coef = fminf(1.0f, a.clip / (tn + 1e-6f));
  const float bc1 = 1.0f - powf(a.b1, t), bc2 = 1.0f - powf(a.b2, t);
  const float step = a.lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  if (first) {
    const float g = g0 * coef + a.l2 * p0;
    const float m = a.b1 * m0 + (1.0f - a.b1) * g;
    const float v = a.b2 * v0 + (1.0f - a.b2) * g * g;
    a.M[i] = m;
    a.Vv[i] = v;
    a.P[i] = p0 - step * m / (sqrtf(v) * rs2 + a.eps);
    i += stride;
  }
  for (; i < a.n; i += stride) {
    const float g = a.G[i] * coef + a.l2 * a.P[i];
    const float m = a.b1 * a.M[i] + (1.0f - a.b1) * g;
    const float v = a.b2 * a.Vv[i] + (1.0f - a.b2) * g * g;
    a.M[i] = m;
    a.Vv[i] = v;
    a.P[i] -= step * m / (sqrtf(v) * rs2 + a.eps);
  }
}

// The last fold of the backward (k_replica_reduce2) and k_wd_gradnorm in ONE pass, for a single-GPU step: every gradient element passes
// through the fold anyway (job 0: item table = G + its replicas, job 1: every other parameter), so the scaling, the weight-decay term of the
// item table and the partial sums of ||g||^2 are applied where the sum is formed.  Same per-element arithmetic as the two kernels in sequence.
__global__ __launch_bounds__(256) void k_fold_wd_gradnorm(RepReduce2Args a, OptArgs o) {
  __shared__ float sbuf[4];
  const float nrm = sqrtf(sum64(o.scal + 64, sbuf));
  const float coef = (o.wd != 0.f && nrm > 0.f) ? o.wd / nrm : 0.f;
  const int j = (int)blockIdx.x >= a.g0 ? 1 : 0;
  const int bid = j ? blockIdx.x - a.g0 : blockIdx.x, nblk = j ? gridDim.x - a.g0 : a.g0;
  float* dst = a.dst[j];
  const float* rep = a.rep[j];
  const float* Pj = o.P + (dst - o.G);           // the parameters under this job's gradient range
  const size_t n = a.n[j], stride = a.stride[j];
  const int nrep = a.nrep[j];
  float acc = 0.f;
  for (size_t i = ((size_t)bid * 256 + threadIdx.x) * 4; i < n; i += (size_t)nblk * 1024) {
    float4 g = replica_sum4(rep + i, stride, nrep, *reinterpret_cast<const float4*>(dst + i));
    g.x *= o.grad_scale; g.y *= o.grad_scale; g.z *= o.grad_scale; g.w *= o.grad_scale;
    if (j == 0) {                                // the item table is job 0 (flat offset 0, nE floats): + wd * E / ||E||_F
      const float4 p = *reinterpret_cast<const float4*>(Pj + i);
      g.x += coef * p.x; g.y += coef * p.y; g.z += coef * p.z; g.w += coef * p.w;
    }
    *reinterpret_cast<float4*>(dst + i) = g;
    acc += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
  }
  const float s = block_sum(acc, sbuf);
  if (threadIdx.x == 0) atomicAdd(o.scal + 128 + (blockIdx.x & 63), s);
  if (blockIdx.x == 0 && threadIdx.x == 0) { o.scal[0] = nrm * nrm; o.scal[3] = o.wd * nrm; o.scal[2] += 1.0f; }
}

// The same pass that ALSO sums the per-sequence partials of the 64 x 64 weight gradients (adt_seqbwd_tt.cuh: sb_dw_tiles), in workgroup
// order, into G -- the two k_dwpart_reduce launches of a step (one beside the encoder's chain kernels, which it slowed, one at the tail in
// front of a cross-queue join) and their atomics are gone, and every weight gradient is one ordered sum.  Job 1 (the non-item parameters)
// leaves the rows of the weight blocks alone (rowmask: one bit per 64 floats from `mask_base`); job 2 owns them:
// G[block] = G[block] + sum_wg part[wg][slot] (the replicas hold nothing there: the chain kernels wrote partials instead).
// Job 2: see its body.
constexpr int FP_MAXSLOTS = 64, FP_MASKWORDS = 160;      // 64 x 64 blocks per step ; 160 x 32 rows of 64 floats = 327,680 non-item floats
// Job 3: the bias / LayerNorm / head-classifier gradient sums the chain kernels STORED per workgroup (BwdChainArgs::vpart, SeqBwdArgs::vpart, the
// last LayerNorm's per-block sums) instead of adding them to replicas with float atomics: chunk i is 64 consecutive floats of G at off[i],
// = the sum over nwg[i] workgroups of 64 floats at vpart + src[i] + wg * stride[i], ascending.  One workgroup per chunk: wave q sums a quarter
// of the workgroups (lane = element, 256-byte rows), the four range sums are joined in order.
constexpr int FV_MAXCHUNKS = 80;
struct VecFoldArgs { const float* vpart; int n; int src[FV_MAXCHUNKS]; int nwg[FV_MAXCHUNKS]; int stride[FV_MAXCHUNKS]; int off[FV_MAXCHUNKS]; };
struct PartFoldArgs {
  const float* part; size_t stride; int nslots;
  int slot[FP_MAXSLOTS]; int off[FP_MAXSLOTS];      // slot inside a workgroup's partial area ; float offset of the block in G
  int nwg[FP_MAXSLOTS];                             // workgroups that wrote the slot (B, or B * S for kernels with S workgroups per sequence)
  uint32_t rowmask[FP_MASKWORDS]; int64_t mask_base;       // bit r: row r (64 floats, from G + mask_base) belongs to a weight block
};
__global__ __launch_bounds__(256) void k_fold_parts_gradnorm(RepReduce2Args a, OptArgs o, PartFoldArgs pf, VecFoldArgs vf, int g1, int g2) {
  __shared__ float sbuf[4];
  __shared__ float4 sq[256];
  const float nrm = sqrtf(sum64(o.scal + 64, sbuf));
  const float coef = (o.wd != 0.f && nrm > 0.f) ? o.wd / nrm : 0.f;
  float acc = 0.f;
  if ((int)blockIdx.x < a.g0 + g1) {
    const int j = (int)blockIdx.x >= a.g0 ? 1 : 0;
    const int bid = j ? blockIdx.x - a.g0 : blockIdx.x, nblk = j ? g1 : a.g0;
    float* dst = a.dst[j];
    const float* rep = a.rep[j];
    const float* Pj = o.P + (dst - o.G);           // the parameters under this job's gradient range
    const size_t n = a.n[j], stride = a.stride[j];
    const int nrep = a.nrep[j];
    const int64_t row0 = j ? ((dst - o.G) - pf.mask_base) / 64 : 0;
    for (size_t i = ((size_t)bid * 256 + threadIdx.x) * 4; i < n; i += (size_t)nblk * 1024) {
      if (j == 1) {
        const int64_t r = row0 + (int64_t)(i >> 6);
        if ((pf.rowmask[r >> 5] >> (r & 31)) & 1u) continue;      // a weight block's row: job 2
      }
      float4 g = replica_sum4(rep + i, stride, nrep, *reinterpret_cast<const float4*>(dst + i));
      g.x *= o.grad_scale; g.y *= o.grad_scale; g.z *= o.grad_scale; g.w *= o.grad_scale;
      if (j == 0) {                                // the item table is job 0 (flat offset 0, nE floats): + wd * E / ||E||_F
        const float4 p = *reinterpret_cast<const float4*>(Pj + i);
        g.x += coef * p.x; g.y += coef * p.y; g.z += coef * p.z; g.w += coef * p.w;
      }
      *reinterpret_cast<float4*>(dst + i) = g;
      acc += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
    }
  } else if ((int)blockIdx.x >= a.g0 + g1 + g2) {      // job 3
    const int ci = blockIdx.x - a.g0 - g1 - g2, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int nwg = vf.nwg[ci], per = (nwg + 3) / 4, w0 = q * per, w1 = min(nwg, w0 + per);
    const float* p = vf.vpart + vf.src[ci] + lane;
    const size_t st = (size_t)vf.stride[ci];
    float sum = 0.f;
    int wg = w0;
    for (; wg + 16 <= w1; wg += 16) {      // sixteen rows in flight (the last LayerNorm has 1,024 of them)
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(wg + u) * st];
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += v[u];
    }
    for (; wg < w1; ++wg) sum += p[(size_t)wg * st];
    float* sf = reinterpret_cast<float*>(sq);
    sf[threadIdx.x] = sum;
    __syncthreads();
    if (q == 0) {
      float* dst = o.G + vf.off[ci] + lane;
      const float g = (*dst + (((sf[lane] + sf[64 + lane]) + sf[128 + lane]) + sf[192 + lane])) * o.grad_scale;
      *dst = g;
      acc += g * g;
    }
  } else {
    // bf16 partials: element (tile, lane, r) of a slot at (tile * 64 + lane) * 4 + r.  A work item = 16 consecutive elements (32 bytes):
    // lanes l .. l + 3 of one tile, r = 0 .. 3 -> rows n = 16 nt + 4 g + r (four of them), columns k .. k + 3: four float4 of G.
    // 8 workgroups of this job per slot, 32 work items each, EIGHT threads per work item (an eighth of the workgroups' partials each,
    // ascending), joined in order.
    const int bid = blockIdx.x - a.g0 - g1;      // job 2
    const int js = bid >> 3, item = (bid & 7) * 32 + (threadIdx.x >> 3), q = threadIdx.x & 7;
    const int nwg = pf.nwg[js];
    const int per = (nwg + 7) / 8, w0 = q * per, w1 = min(nwg, w0 + per);
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const __bf16* p = reinterpret_cast<const __bf16*>(pf.part + (size_t)pf.slot[js] * 4096) + item * 16;
    const size_t strideb = pf.stride * 2;                  // bf16 elements between two workgroups' areas
    float sum[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] = 0.f;
    auto add = [&](const u4v& lo, const u4v& hi) {
      const unsigned wds[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) { sum[2 * i] += __uint_as_float(wds[i] << 16); sum[2 * i + 1] += __uint_as_float(wds[i] & 0xFFFF0000u); }
    };
    int wg = w0;
    for (; wg + 4 <= w1; wg += 4) {
      u4v lo[4], hi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const u4v* src = reinterpret_cast<const u4v*>(p + (size_t)(wg + u) * strideb);
        lo[u] = src[0]; hi[u] = src[1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(lo[u], hi[u]);
    }
    for (; wg < w1; ++wg) {
      const u4v* src = reinterpret_cast<const u4v*>(p + (size_t)wg * strideb);
      add(src[0], src[1]);
    }
    // join the eight range sums in order through LDS (sq: 256 float4 = 4 KB ; 16 floats per thread = 4 float4: four rounds)
    const int tile = item >> 4, lane0 = (item & 15) * 4;      // element e = (tile * 64 + lane0 + j) * 4 + r  <->  sum[4 j + r]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __syncthreads();
      sq[threadIdx.x] = make_float4(sum[r], sum[4 + r], sum[8 + r], sum[12 + r]);      // row r: lanes lane0 .. lane0 + 3 = columns k .. k + 3
      __syncthreads();
      if (q == 0) {
        float4 t = sq[threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float4 v = sq[threadIdx.x + k]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float* dst = o.G + pf.off[js] + (16 * (tile >> 2) + 4 * (lane0 >> 4) + r) * 64 + 16 * (tile & 3) + (lane0 & 15);
        float4 g = *reinterpret_cast<const float4*>(dst);
        g.x = (g.x + t.x) * o.grad_scale; g.y = (g.y + t.y) * o.grad_scale; g.z = (g.z + t.z) * o.grad_scale; g.w = (g.w + t.w) * o.grad_scale;
        *reinterpret_cast<float4*>(dst) = g;
        acc += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
      }
    }
  }
  const float s = block_sum(acc, sbuf);
  if (threadIdx.x == 0) {
    if (o.gn_part) o.gn_part[blockIdx.x] = s;      // summed in block order by k_adam
    else atomicAdd(o.scal + 128 + (blockIdx.x & 63), s);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && !o.fold_only) { o.scal[0] = nrm * nrm; o.scal[3] = o.wd * nrm; o.scal[2] += 1.0f; }
}

// ---------------------------------------------------------------------------------------------
// SASRecADT.predict (sasrec/model.py:89-96) + the rank of evaluate_loader (sasrec/utils.py:410):
// logits[b][c] = E[cand[b][c]] . f[b] ; rank[b] = #{c > 0 : logits[b][c] > logits[b][0]}.
struct ScoreArgs {
  const float* F; int ldf;   // B rows (already the last position), d wide
  const float* E;
  const int* cand;           // B x C item ids, or null: all items 0..C-1
  int B, C, d;
  float* logits;             // B x C
  int* rank;                 // B (may be null)
  const float* bias;         // optional per-item bias (bert4rec/model/bert.py:89 mask_bias)
};

__global__ __launch_bounds__(256) void k_score(ScoreArgs a) {
  // one wave per (b, candidate) pair group: 16 lanes per candidate
  const int sub = threadIdx.x & 15;
  const size_t n = (size_t)a.B * a.C;
  for (size_t i = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < n; i += (size_t)gridDim.x * 16) {
    const int b = (int)(i / a.C), cidx = (int)(i % a.C);
    const int item = a.cand ? a.cand[i] : cidx;
    float s = 0.f;
    for (int c4 = 4 * sub; c4 < a.d; c4 += 64) {
      const float4 f = *reinterpret_cast<const float4*>(a.F + (size_t)b * a.ldf + c4);
      const float4 e = *reinterpret_cast<const float4*>(a.E + (size_t)item * a.d + c4);
      s += f.x * e.x + f.y * e.y + f.z * e.z + f.w * e.w;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) a.logits[i] = s + (a.bias ? a.bias[item] : 0.f);
  }
}

__global__ __launch_bounds__(256) void k_rank(ScoreArgs a) {
  const int lane = threadIdx.x & 63;
  for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < a.B; b += gridDim.x * 4) {
    const float* row = a.logits + (size_t)b * a.C;
    const float s0 = row[0];
    int cnt = 0;
    for (int cidx = 1 + lane; cidx < a.C; cidx += 64) cnt += (row[cidx] > s0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) a.rank[b] = cnt;
  }
}

}  // namespace adt

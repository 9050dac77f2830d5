// Fused token-parallel forward chains.  One kernel, k_rowchain_fwd, interprets a short program of steps over
// LDS-resident 64-row x 64-column fp32 tiles, so that everything between two attention calls of the
// reference's EncoderLayer / DecoderLayer (sasrec/modules.py:644-655, :666-677) -- embedding gather,
// LayerNorm, the q/k/v projections, out_proj + residual + LayerNorm + the two-conv FFN with its dropouts,
// ReLU, residual and padding mask, the head classifier, the last LayerNorm and the pos/neg logits
// (sasrec/model.py:34-41,48,72-76) -- touches HBM once per tensor instead of once per ATen op.
//
// Geometry: 512 threads (8 waves), tile = 64 tokens; wave w owns output rows 16*(w>>1).. and columns
// 32*(w&1).. of every 64x64x64 product (two 16x16 MFMA accumulators).  3 tile buffers + 1 weight buffer
// (RS = 68 floats) = 70 KB of LDS, so two workgroups share a CU.  d == 64 only.
#pragma once
#include "adt_chain_args.h"
#include "adt_common.cuh"
#include "adt_misc.cuh"

namespace adt {

ADT_DEVICE_INLINE void ch_load(float* s, const float* g, int ld, int row0, int T) {
  for (int i = threadIdx.x; i < 64 * 16; i += CH_THREADS) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < T) v = *reinterpret_cast<const float4*>(g + (size_t)(row0 + r) * ld + c4);
    *reinterpret_cast<float4*>(s + r * CH_RS + c4) = v;
  }
}

ADT_DEVICE_INLINE void ch_store(float* g, int ld, const float* s, int row0, int T) {
  for (int i = threadIdx.x; i < 64 * 16; i += CH_THREADS) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    if (row0 + r < T) *reinterpret_cast<float4*>(g + (size_t)(row0 + r) * ld + c4) = *reinterpret_cast<const float4*>(s + r * CH_RS + c4);
  }
}

// LayerNorm of a 64x64 tile: 8 lanes per row, 8 columns per lane
ADT_DEVICE_INLINE void ch_ln(float* dst, const float* src, const float* gamma, const float* beta, float eps) {
  const int r = threadIdx.x >> 3, part = threadIdx.x & 7;
  float x[8];
  *reinterpret_cast<float4*>(x) = *reinterpret_cast<const float4*>(src + r * CH_RS + 8 * part);
  *reinterpret_cast<float4*>(x + 4) = *reinterpret_cast<const float4*>(src + r * CH_RS + 8 * part + 4);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += x[e];
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s * (1.0f / 64);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { x[e] -= mu; q += x[e] * x[e]; }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q * (1.0f / 64) + eps);
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = x[e] * rstd * gamma[8 * part + e] + beta[8 * part + e];
  *reinterpret_cast<float4*>(dst + r * CH_RS + 8 * part) = *reinterpret_cast<float4*>(y);
  *reinterpret_cast<float4*>(dst + r * CH_RS + 8 * part + 4) = *reinterpret_cast<float4*>(y + 4);
}

template <int PREC>
__global__ __launch_bounds__(CH_THREADS) void k_rowchain_fwd(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem + CH_NBUF * CH_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
  const int ntiles = (a.T + 63) / 64;
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * 64;
    for (int si = 0; si < CH_MAXSTEPS; ++si) {
      const ChainStep& st = a.steps[si];
      if (st.op == ST_END) break;
      float* dst = smem + st.dst * CH_TILE;
      const float* src = smem + st.src * CH_TILE;
      __syncthreads();   // previous step's LDS traffic (and the previous tile's) is complete
      if (st.op == ST_LOAD) {
        ch_load(dst, st.in_g, st.ld_in, row0, a.T);
      } else if (st.op == ST_GATHER) {
        // x = dropout(E[id] * sqrt(d) + P[l]) * (id != 0)           (sasrec/model.py:34-41)
        const uint32_t key = adt_site_key(seedv, st.site);
        for (int i = threadIdx.x; i < 64 * 16; i += CH_THREADS) {
          const int r = i >> 4, c4 = (i & 15) * 4;
          const int row = row0 + r;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          const int id = row < a.T ? a.ids[row] : 0;
          if (id != 0) {
            const float4 e = *reinterpret_cast<const float4*>(a.E + (size_t)id * 64 + c4);
            const float4 p = *reinterpret_cast<const float4*>(a.P + (size_t)(row % a.L) * 64 + c4);
            v[0] = e.x * a.emb_scale + p.x; v[1] = e.y * a.emb_scale + p.y;
            v[2] = e.z * a.emb_scale + p.z; v[3] = e.w * a.emb_scale + p.w;
            if (a.drop.thr) {
              const uint32_t base = (uint32_t)(row + a.row_offset) * 64u + (uint32_t)c4;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
            }
          }
          *reinterpret_cast<float4*>(dst + r * CH_RS + c4) = *reinterpret_cast<float4*>(v);
        }
      } else if (st.op == ST_LN) {
        ch_ln(dst, src, st.W, st.b, a.ln_eps);
      } else if (st.op == ST_GEMM) {
        // stage the 64x64 weight block, then C = A W^T
        for (int i = threadIdx.x; i < 64 * 16; i += CH_THREADS) {
          const int r = i >> 4, c4 = (i & 15) * 4;
          *reinterpret_cast<float4*>(sW + r * CH_RS + c4) = *reinterpret_cast<const float4*>(st.W + r * 64 + c4);
        }
        __syncthreads();
        f32x4 acc[2] = {};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const Frag8 fa = frag_contig(src + (16 * wr + c) * CH_RS + kb * 32 + 8 * g);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const Frag8 fb = frag_contig(sW + (32 * wc + 16 * nt + c) * CH_RS + kb * 32 + 8 * g);
            acc[nt] = mma16<PREC>(acc[nt], fa, fb);
          }
        }
        if (st.dst == st.src) __syncthreads();   // in-place: every wave has read its A rows
        const uint32_t key = (st.flags & F_DROP) ? adt_site_key(seedv, st.site) : 0u;
        const float* addb = st.add_buf >= 0 ? smem + st.add_buf * CH_TILE : nullptr;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int col = 32 * wc + 16 * nt + c;
          const float bias = st.b ? st.b[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = 16 * wr + 4 * g + r, row = row0 + rl;
            float v = acc[nt][r] + bias;
            if ((st.flags & F_DROP) && a.drop.thr)
              v = adt_keep(key, (uint32_t)(row + a.row_offset) * 64u + (uint32_t)col, a.drop.thr) ? v * a.drop.scale : 0.f;
            if (st.flags & F_RELU) v = fmaxf(v, 0.f);
            if (addb) v += addb[rl * CH_RS + col];
            if (st.in_g && row < a.T) v += st.in_g[(size_t)row * st.ld_in + col];
            if ((st.flags & F_MASK) && row < a.T && a.ids[row] == 0) v = 0.f;
            dst[rl * CH_RS + col] = v;
          }
        }
      } else if (st.op == ST_CLS) {
        // head classifier on the src tile: one thread per (row, head); rec row order l*B + b
        const int hd = 64 / a.H;
        for (int i = threadIdx.x; i < 64 * a.H; i += CH_THREADS) {
          const int rl = i / a.H, h = i % a.H, row = row0 + rl;
          if (row >= a.T) continue;
          float z[MAXH];
#pragma unroll
          for (int cc = 0; cc < MAXH; ++cc) z[cc] = (cc < a.H) ? a.bs[cc] : -INFINITY;
          for (int j = 0; j < hd; ++j) {
            const float ov = src[rl * CH_RS + h * hd + j];
#pragma unroll
            for (int cc = 0; cc < MAXH; ++cc)
              if (cc < a.H) z[cc] += ov * a.Ws[cc * hd + j];
          }
          float m = z[0];
#pragma unroll
          for (int cc = 1; cc < MAXH; ++cc) m = fmaxf(m, z[cc]);
          float s = 0.f;
#pragma unroll
          for (int cc = 0; cc < MAXH; ++cc) s += (cc < a.H) ? expf(z[cc] - m) : 0.f;
          const float lz = m + logf(s);
          const int b = row / a.L, l = row % a.L;
          float* o = a.rec + ((size_t)(l * a.B + b) * a.H + h) * a.H;
#pragma unroll
          for (int cc = 0; cc < MAXH; ++cc)
            if (cc < a.H) o[cc] = z[cc] - lz;
        }
      } else if (st.op == ST_LOGITS) {
        // pos/neg logits of the src tile rows: 8 lanes per row                   (sasrec/model.py:72-76)
        const int rl = threadIdx.x >> 3, part = threadIdx.x & 7, row = row0 + rl;
        float sp = 0.f, sn = 0.f;
        if (row < a.T) {
          const int ip = a.pos[row], in = a.neg[row];
#pragma unroll
          for (int e = 0; e < 8; e += 4) {
            const float4 f = *reinterpret_cast<const float4*>(src + rl * CH_RS + 8 * part + e);
            const float4 p = *reinterpret_cast<const float4*>(a.E + (size_t)ip * 64 + 8 * part + e);
            const float4 q = *reinterpret_cast<const float4*>(a.E + (size_t)in * 64 + 8 * part + e);
            sp += f.x * p.x + f.y * p.y + f.z * p.z + f.w * p.w;
            sn += f.x * q.x + f.y * q.y + f.z * q.z + f.w * q.w;
          }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sn += __shfl_xor(sn, o, 64); }
        if (part == 0 && row < a.T) { a.pos_logits[row] = sp; a.neg_logits[row] = sn; }
      }
      if (st.out_g && (st.op == ST_GATHER || st.op == ST_LN || st.op == ST_GEMM)) {
        __syncthreads();
        ch_store(st.out_g, st.ld_out, dst, row0, a.T);
      }
    }
  }
}

}  // namespace adt

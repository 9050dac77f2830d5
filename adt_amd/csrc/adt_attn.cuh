// Fused scaled-dot-product attention, forward and backward, one workgroup per (sequence, head).
// Replaces sasrec/modules.py:21-64 (_scaled_dot_product_attention: q/sqrt(hd) k^T + mask -> softmax ->
// dropout -> @v) together with the head split/merge of multi_head_attention_forward (:457-468,:517), and
// torch.nn.MultiheadAttention's identical core for the decoder (:661-662).  The (B*H, L, L) probability
// tensor the reference materialises (81.9 MB per call at cfg-A) never leaves registers.
//
// Layout: q/k/v/o are token-major (B*L rows), head h in columns [h*HD, (h+1)*HD) with arbitrary leading
// dimension, so the packed qkv projection output is consumed in place.  The whole K and V (and, backward,
// Q and dO) of one (b, h) are resident in LDS as fp32 row images with row stride HD+4.
//
// Forward, per wave, per 16-query tile: S^T = K Q^T with the KEY on the accumulator rows, so that the
// softmax row statistics are reductions over registers + 2 cross-lane steps, and the normalised,
// dropped-out probabilities are already the A operand of P V (accumulator-as-operand, no LDS round trip).
#pragma once
#include "adt_common.cuh"

namespace adt {

struct AttnArgs {
  const float* Q; int ldq;
  const float* K; int ldk;
  const float* V; int ldv;
  float* O; int ldo;          // forward: output; backward: forward output (input)
  float* LSE;                 // (B*H*L) log-sum-exp of the scaled, masked scores
  int B, H, L;
  int causal;
  float scale;                // 1/sqrt(HD)
  DropCfg drop;               // idx = ((bh + bh_offset) * L + q) * L + key
  uint32_t bh_offset;
  const float* dO; int lddo;  // backward
  float* dQ; int lddq;
  float* dK; int lddk;
  float* dV; int lddv;
  uint32_t* mask;             // optional (B*H*L x 8 words): dropout keep bits written by the bf16 forward, read by its backward
  unsigned long long* stamps; // timing experiments only (ADT_SEQ_STAMPS): s_memtime per wave of workgroup 0
  int in_bf16;                // backward, adt_seqattn.cuh only: Q, K, V, O point at bf16 rows and ldq / ldk / ldv / ldo count bf16 elements
  int out_bf16;               // backward, adt_seqattn.cuh only: dQ, dK, dV are written as bf16 rows in the saved-row order (adt_tt.cuh: tt_store_bf16), lddq /
                              // lddk / lddv count bf16 elements -- for a consumer that only builds bf16 MFMA operands from them (k_seqtt_mid_bwd): the same
                              // values it would have rounded itself, half the bytes
};

template <int HD>
ADT_DEVICE_INLINE Frag8 frag_contig_hd(const float* row, int kb, int g) {
  // 8 contiguous floats at column kb*32 + 8g of an HD-wide row; zero beyond HD (HD = 16 case)
  if (kb * 32 + 8 * g < HD) return frag_contig(row + kb * 32 + 8 * g);
  Frag8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z.v[j] = 0.f;
  return z;
}

template <int HD, int RS, int NTH>
ADT_DEVICE_INLINE void stage_head(float* s, const float* g, int ld, int L, int LP, float mul) {
  constexpr int V = HD / 4;
  for (int i = threadIdx.x; i < LP * V; i += NTH) {
    const int r = i / V, c4 = i % V;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) {
      v = *reinterpret_cast<const float4*>(g + (size_t)r * ld + 4 * c4);
      v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    }
    *reinterpret_cast<float4*>(s + r * RS + 4 * c4) = v;
  }
}

template <int PREC, int HD, int MAXKT, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_fwd(AttnArgs a) {
  constexpr int RS = HD + 4, LP = MAXKT * 16, NT = HD / 16, KB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sK = smem;
  float* sV = smem + LP * RS;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  stage_head<HD, RS, NW * 64>(sK, a.K + row_b * a.ldk + h * HD, a.ldk, L, LP, 1.0f);
  stage_head<HD, RS, NW * 64>(sV, a.V + row_b * a.ldv + h * HD, a.ldv, L, LP, 1.0f);
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const int nqt = (L + 15) / 16;
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    // snake assignment, heaviest causal tile (largest qt) first: balances the triangular work over waves
    const int tix = rnd * NW + ((rnd & 1) ? NW - 1 - w : w);
    if (tix >= nqt) continue;
    const int qt = nqt - 1 - tix;
    const int q = qt * 16 + c;
    Frag8 fq[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) fq[kb].v[j] = 0.f;
      if (q < L && kb * 32 + 8 * g < HD) {
        fq[kb] = frag_contig(a.Q + (row_b + q) * a.ldq + h * HD + kb * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) fq[kb].v[j] *= a.scale;
      }
    }
    const int nkt = a.causal ? qt + 1 : nqt;
    f32x4 s[MAXKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const Frag8 fk = frag_contig_hd<HD>(sK + (kt * 16 + c) * RS, kb, g);
          s[kt] = mma16<PREC>(s[kt], fk, fq[kb]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const bool valid = key < L && (!a.causal || key <= q);
          s[kt][r] = valid ? s[kt][r] : -INFINITY;
          m = fmaxf(m, s[kt][r]);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(s[kt][r] - m);
          s[kt][r] = e;
          sum += e;
        }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (g == 0 && q < L) a.LSE[(size_t)bh * L + q] = m + __logf(sum);
    const uint32_t idx_q = ((uint32_t)(bh + a.bh_offset) * (uint32_t)L + (uint32_t)q) * (uint32_t)L;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = s[kt][r] * inv;
          if (a.drop.thr) {
            const uint32_t key = kt * 16 + 4 * g + r;
            p = adt_keep(key_rng, idx_q + key, a.drop.thr) ? p * a.drop.scale : 0.f;
          }
          s[kt][r] = p;
        }
      }
    }
    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < MAXKT / 2; ++kp) {
      if (2 * kp < nkt) {
        Frag8 fp;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fp.v[j] = s[2 * kp][j];
          fp.v[4 + j] = (2 * kp + 1 < nkt) ? s[2 * kp + 1][j] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const Frag8 fv = frag_strided(sV + (kp * 32) * RS + nt * 16 + c, RS, g);
          o[nt] = mma16<PREC>(o[nt], fp, fv);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + 4 * g + r;
        if (qq < L) a.O[(row_b + qq) * a.ldo + h * HD + nt * 16 + c] = o[nt][r];
      }
  }
}

// Backward: recompute P from Q, K and the saved LSE.  Pass A (wave owns a query tile) produces dQ; pass B
// (wave owns a key tile) produces dK and dV, so no gradient is summed across waves or workgroups and the
// result is bitwise reproducible.
template <int PREC, int HD, int MAXKT, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_bwd(AttnArgs a) {
  constexpr int RS = HD + 4, LP = MAXKT * 16, NT = HD / 16, KB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sQ = smem;               // pre-scaled by 1/sqrt(HD)
  float* sK = sQ + LP * RS;
  float* sV = sK + LP * RS;
  float* sdO = sV + LP * RS;
  float* sLse = sdO + LP * RS;    // +inf for padded queries -> P = 0
  float* sDelta = sLse + LP;      // rowsum(dO * O)
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  stage_head<HD, RS, NW * 64>(sQ, a.Q + row_b * a.ldq + h * HD, a.ldq, L, LP, a.scale);
  stage_head<HD, RS, NW * 64>(sK, a.K + row_b * a.ldk + h * HD, a.ldk, L, LP, 1.0f);
  stage_head<HD, RS, NW * 64>(sV, a.V + row_b * a.ldv + h * HD, a.ldv, L, LP, 1.0f);
  {
    constexpr int V4 = HD / 4;  // threads per row (4, 8 or 16): a row's threads are adjacent lanes
    for (int i = threadIdx.x; i < LP * V4; i += NW * 64) {
      const int r = i / V4, c4 = i % V4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      float part = 0.f;
      if (r < L) {
        v = *reinterpret_cast<const float4*>(a.dO + (row_b + r) * a.lddo + h * HD + 4 * c4);
        const float4 o = *reinterpret_cast<const float4*>(a.O + (row_b + r) * a.ldo + h * HD + 4 * c4);
        part = v.x * o.x + v.y * o.y + v.z * o.z + v.w * o.w;
      }
      *reinterpret_cast<float4*>(sdO + r * RS + 4 * c4) = v;
#pragma unroll
      for (int off = V4 / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      if (c4 == 0) {
        sDelta[r] = part;
        sLse[r] = (r < L) ? a.LSE[(size_t)bh * L + r] : INFINITY;
      }
    }
  }
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const uint32_t idx_bh = (uint32_t)(bh + a.bh_offset) * (uint32_t)L;
  const int nqt = (L + 15) / 16;

  // ---- pass A: dQ -------------------------------------------------------------------------
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int tix = rnd * NW + ((rnd & 1) ? NW - 1 - w : w);
    if (tix >= nqt) continue;
    const int qt = nqt - 1 - tix;   // heaviest first (snake order)
    const int q = qt * 16 + c;
    const float lse_q = sLse[q], delta_q = sDelta[q];
    Frag8 fq[KB], fdo[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fq[kb] = frag_contig_hd<HD>(sQ + q * RS, kb, g);
      fdo[kb] = frag_contig_hd<HD>(sdO + q * RS, kb, g);
    }
    const int nkt = a.causal ? qt + 1 : nqt;
    f32x4 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t idx_q = (idx_bh + (uint32_t)q) * (uint32_t)L;
#pragma unroll
    for (int kp = 0; kp < MAXKT / 2; ++kp) {
      if (2 * kp < nkt) {
        Frag8 fds;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kp + t;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if (kt < nkt) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              const Frag8 fk = frag_contig_hd<HD>(sK + (kt * 16 + c) * RS, kb, g);
              const Frag8 fv = frag_contig_hd<HD>(sV + (kt * 16 + c) * RS, kb, g);
              s = mma16<PREC>(s, fk, fq[kb]);
              dp = mma16<PREC>(dp, fv, fdo[kb]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            const bool valid = kt < nkt && key < L && (!a.causal || key <= q);
            const float p = valid ? __expf(s[r] - lse_q) : 0.f;
            float d = dp[r];
            if (a.drop.thr) d = adt_keep(key_rng, idx_q + (uint32_t)key, a.drop.thr) ? d * a.drop.scale : 0.f;
            fds.v[4 * t + r] = p * (d - delta_q);
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const Frag8 fk = frag_strided(sK + (kp * 32) * RS + nt * 16 + c, RS, g);
          dq[nt] = mma16<PREC>(dq[nt], fds, fk);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + 4 * g + r;
        if (qq < L) a.dQ[(row_b + qq) * a.lddq + h * HD + nt * 16 + c] = dq[nt][r] * a.scale;
      }
  }

  // ---- pass B: dK, dV ---------------------------------------------------------------------
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int kt = rnd * NW + ((rnd & 1) ? NW - 1 - w : w);   // key tile 0 is the heaviest under the causal mask
    if (kt >= nqt) continue;
    const int key = kt * 16 + c;
    Frag8 fk[KB], fv[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fk[kb] = frag_contig_hd<HD>(sK + key * RS, kb, g);
      fv[kb] = frag_contig_hd<HD>(sV + key * RS, kb, g);
    }
    f32x4 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      dk[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int qp0 = a.causal ? kt / 2 : 0;
#pragma unroll
    for (int qp = 0; qp < MAXKT / 2; ++qp) {
      if (qp >= qp0 && 2 * qp < nqt) {
        Frag8 fp, fds;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int qt = 2 * qp + t;
          const bool live = qt < nqt && (!a.causal || qt >= kt);
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if (live) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              const Frag8 fq = frag_contig_hd<HD>(sQ + (qt * 16 + c) * RS, kb, g);
              const Frag8 fdo = frag_contig_hd<HD>(sdO + (qt * 16 + c) * RS, kb, g);
              s = mma16<PREC>(s, fq, fk[kb]);
              dp = mma16<PREC>(dp, fdo, fv[kb]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = qt * 16 + 4 * g + r;   // < LP always (MAXKT even)
            const bool valid = live && key < L && (!a.causal || key <= qq);
            const float p = valid ? __expf(s[r] - sLse[qq]) : 0.f;   // sLse = +inf for qq >= L
            float ks = 1.0f;
            if (a.drop.thr)
              ks = adt_keep(key_rng, (idx_bh + (uint32_t)qq) * (uint32_t)L + (uint32_t)key, a.drop.thr) ? a.drop.scale : 0.f;
            fp.v[4 * t + r] = p * ks;
            fds.v[4 * t + r] = p * (dp[r] * ks - sDelta[qq]);
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const Frag8 fdo = frag_strided(sdO + (qp * 32) * RS + nt * 16 + c, RS, g);
          const Frag8 fq = frag_strided(sQ + (qp * 32) * RS + nt * 16 + c, RS, g);
          dv[nt] = mma16<PREC>(dv[nt], fp, fdo);
          dk[nt] = mma16<PREC>(dk[nt], fds, fq);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + 4 * g + r;
        if (kk < L) {
          a.dK[(row_b + kk) * a.lddk + h * HD + nt * 16 + c] = dk[nt][r];
          a.dV[(row_b + kk) * a.lddv + h * HD + nt * 16 + c] = dv[nt][r];
        }
      }
  }
}

}  // namespace adt

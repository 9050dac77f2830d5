// TEST INFRASTRUCTURE ONLY -- libadt_cpu.so: C++17 / OpenMP fp32 restatement of the SASRec-ADT training + ranking step (the CPU twins
// `adt_cpu_*` of the model-level entry points of include/adt_hip.h).  Used by tests/ (checked against the golden vectors
// recorded from the reference and against the numpy oracle), by __graft_entry__.smoke() and as bench.py's `cpu_baseline` (all host cores).
// The product path (adt_amd/) never loads it.
//
// Parity status: PINNED -- tests/test_cpu_restatement.py holds it to tests/golden/sasrec_*.npz (forward tensors, loss, every gradient,
// clip norm, post-Adam weights; recorded from /root/reference/sasrec by tools/gen_golden.py) and, with dropout on, to
// oracle/sasrec_oracle.py through the shared hash RNG (oracle/rng.py; identical masks).
//
// Every function cites the reference lines it restates (paths relative to /root/reference).  One OpenMP task = one user sequence:
// LayerNorm is per token, attention per sequence and every loss a sum over tokens over a GLOBAL normaliser, so sequences only meet in
// the gradient sum (thread-private accumulators, folded in a fixed order).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Cfg { int32_t item_num, maxlen, hidden, num_heads, num_layers; float dropout; int32_t prec; };      // = adt_sasrec_cfg (include/adt_hip.h)

constexpr float LN_EPS = 1e-8f;      // sasrec/modules.py:638,640,660 ; sasrec/model.py:28

// ---- dropout RNG: oracle/rng.py == adt_amd/csrc/adt_common.cuh ---------------------------------------------------------------------
inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
inline uint32_t site_key(uint32_t seed, uint32_t site) { return hash32(seed ^ (site * 0x9E3779B9u)); }
struct Drop { uint32_t key, thr; float scale; };
inline Drop make_drop(float p, uint32_t seed, uint32_t site) {
  int t8 = p > 0.f ? (int)((double)p * 256.0 + 0.5) : 0;
  if (t8 > 255) t8 = 255;
  Drop d;
  d.thr = (uint32_t)t8;
  d.key = site_key(seed, site);
  d.scale = t8 ? (float)(1.0 / (1.0 - (double)t8 / 256.0)) : 1.0f;
  return d;
}
inline bool keep(const Drop& d, uint32_t idx) { return ((hash32((idx >> 2) ^ d.key) >> (8u * (idx & 3u))) & 0xFFu) >= d.thr; }
enum { SITE_EMB_SEQ = 1, SITE_EMB_DEC = 2 };
inline uint32_t enc_site(int layer, int which) { return 16u + 8u * (uint32_t)layer + (uint32_t)which; }        // 0 attn 1 ffn1 2 ffn2
inline uint32_t dec_site(int layer, int which) { return 128u + 8u * (uint32_t)layer + (uint32_t)which; }       // 0 slf 1 enc 2 ffn1 3 ffn2

// ---- flat parameter layout: the reference's state_dict tensors in the slot order of adt_sasrec_param_layout, packed without padding ----
struct EncP { size_t ln1w, ln1b, inw, inb, ow, ob, ln2w, ln2b, c1w, c1b, c2w, c2b, sw, sb; };
struct DecP { size_t lnw, lnb, sinw, sinb, sow, sob, einw, einb, eow, eob, c1w, c1b, c2w, c2b, uw, ub; };
struct Layout {
  size_t item, pos, lastw, lastb, total;
  std::vector<EncP> enc;
  std::vector<DecP> dec;
};
Layout make_layout(const Cfg& c) {
  const size_t d = c.hidden, H = c.num_heads, hd = d / H;
  Layout lo;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += n; return r; };
  lo.item = take((size_t)(c.item_num + 1) * d);
  lo.pos = take((size_t)c.maxlen * d);
  lo.lastw = take(d); lo.lastb = take(d);
  for (int i = 0; i < c.num_layers; ++i) {
    EncP e;
    e.ln1w = take(d); e.ln1b = take(d); e.inw = take(3 * d * d); e.inb = take(3 * d); e.ow = take(d * d); e.ob = take(d);
    e.ln2w = take(d); e.ln2b = take(d); e.c1w = take(d * d); e.c1b = take(d); e.c2w = take(d * d); e.c2b = take(d);
    e.sw = take(H * hd); e.sb = take(H);
    lo.enc.push_back(e);
  }
  for (int i = 0; i < c.num_layers; ++i) {
    DecP e;
    e.lnw = take(d); e.lnb = take(d); e.sinw = take(3 * d * d); e.sinb = take(3 * d); e.sow = take(d * d); e.sob = take(d);
    e.einw = take(3 * d * d); e.einb = take(3 * d); e.eow = take(d * d); e.eob = take(d);
    e.c1w = take(d * d); e.c1b = take(d); e.c2w = take(d * d); e.c2b = take(d); e.uw = take(d); e.ub = take(d);
    lo.dec.push_back(e);
  }
  lo.total = o;
  return lo;
}

// ---- workspace: outputs first (same `what` codes as adt_sasrec_ws_offset), then per-sequence saved activations ---------------------
enum { WS_ENC_X = 0, WS_DEC_X, WS_REC, WS_POS_LOGITS, WS_NEG_LOGITS, WS_F, WS_G_ENC_X, WS_G_DEC_X, WS_G_REC, WS_G_POS, WS_G_NEG, WS_LOSS, WS_NORMS };
struct WS {
  size_t T, Td, rec, enc_x, dec_x, recs, posl, negl, f, g_enc_x, g_dec_x, g_rec, g_pos, g_neg, loss, norms, fhat, frstd, save, save_stride, total;
  // per sequence (floats, relative to save + b * save_stride): encoder layer i at enc_off(i), decoder layer i at dec_off(i)
  size_t Ld, HL, enc_sz, dec_sz;
};
WS make_ws(const Cfg& c, int B) {
  WS w;
  const size_t L = c.maxlen, d = c.hidden, H = c.num_heads, nl = c.num_layers;
  w.T = (size_t)B * L; w.Td = w.T * d; w.rec = w.T * H * H; w.Ld = L * d; w.HL = H * L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += n; return r; };
  w.enc_x = take((nl + 1) * w.Td); w.dec_x = take((nl + 1) * w.Td); w.recs = take(nl * w.rec); w.posl = take(w.T); w.negl = take(w.T); w.f = take(w.Td);
  w.g_enc_x = take((nl + 1) * w.Td); w.g_dec_x = take((nl + 1) * w.Td); w.g_rec = take(nl * w.rec); w.g_pos = take(w.T); w.g_neg = take(w.T);
  w.loss = take(2 + 2 * nl); w.norms = take(4);
  w.fhat = take(w.Td); w.frstd = take(w.T);
  w.enc_sz = 6 * w.Ld + w.HL;               // q k v o h u | lse
  w.dec_sz = 11 * w.Ld + 2 * w.HL;          // q1 k1 v1 o1 a1 q2 k2 v2 o2 a2 u | lse1 lse2
  w.save_stride = nl * (w.enc_sz + w.dec_sz);
  w.save = take((size_t)B * w.save_stride);
  w.total = o;
  return w;
}

// ---- per-sequence primitives on row-major [L][n] fp32 blocks -------------------------------------------------------------------------
// torch.nn.LayerNorm over d (biased variance, eps inside the sqrt); xhat / rstd optional outputs
void ln_fwd(const float* x, const float* w, const float* b, int L, int d, float* y, float* xhat, float* rstd) {
  for (int l = 0; l < L; ++l) {
    const float* xr = x + (size_t)l * d;
    float mu = 0.f;
    for (int c = 0; c < d; ++c) mu += xr[c];
    mu /= (float)d;
    float var = 0.f;
    for (int c = 0; c < d; ++c) { const float t = xr[c] - mu; var += t * t; }
    var /= (float)d;
    const float rs = 1.0f / sqrtf(var + LN_EPS);
    if (rstd) rstd[l] = rs;
    for (int c = 0; c < d; ++c) {
      const float xh = (xr[c] - mu) * rs;
      if (xhat) xhat[(size_t)l * d + c] = xh;
      y[(size_t)l * d + c] = xh * w[c] + b[c];
    }
  }
}
// dx (+)= LN'(dy); dw += sum dy*xhat; db += sum dy.  xhat / rstd recomputed from x.
void ln_bwd(const float* dy, const float* x, const float* w, int L, int d, float* dx, bool acc, float* dw, float* db) {
  std::vector<float> xh(d), dxh(d);
  for (int l = 0; l < L; ++l) {
    const float* xr = x + (size_t)l * d;
    const float* g = dy + (size_t)l * d;
    float mu = 0.f;
    for (int c = 0; c < d; ++c) mu += xr[c];
    mu /= (float)d;
    float var = 0.f;
    for (int c = 0; c < d; ++c) { const float t = xr[c] - mu; var += t * t; }
    var /= (float)d;
    const float rs = 1.0f / sqrtf(var + LN_EPS);
    float m1 = 0.f, m2 = 0.f;
    for (int c = 0; c < d; ++c) {
      xh[c] = (xr[c] - mu) * rs;
      dxh[c] = g[c] * w[c];
      m1 += dxh[c]; m2 += dxh[c] * xh[c];
      dw[c] += g[c] * xh[c]; db[c] += g[c];
    }
    m1 /= (float)d; m2 /= (float)d;
    float* o = dx + (size_t)l * d;
    for (int c = 0; c < d; ++c) { const float v = rs * (dxh[c] - m1 - xh[c] * m2); o[c] = acc ? o[c] + v : v; }
  }
}
// y[L][N] = x[L][K] W[N][K]^T + b      (nn.Linear / Conv1d k=1 / a row block of in_proj_weight)
void lin_fwd(const float* x, const float* W, const float* b, int L, int K, int N, float* y) {
  for (int l = 0; l < L; ++l) {
    const float* xr = x + (size_t)l * K;
    float* yr = y + (size_t)l * N;
    for (int n = 0; n < N; ++n) {
      const float* wr = W + (size_t)n * K;
      float acc = 0.f;
#pragma omp simd reduction(+ : acc)
      for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
      yr[n] = acc + b[n];
    }
  }
}
// dx[L][K] (+)= dy W ; dW[N][K] += dy^T x ; db[N] += colsum(dy)
void lin_bwd(const float* dy, const float* x, const float* W, int L, int K, int N, float* dx, bool acc, float* dW, float* db) {
  for (int l = 0; l < L; ++l) {
    const float* g = dy + (size_t)l * N;
    const float* xr = x + (size_t)l * K;
    float* o = dx ? dx + (size_t)l * K : nullptr;
    if (o && !acc) for (int k = 0; k < K; ++k) o[k] = 0.f;
    for (int n = 0; n < N; ++n) {
      const float gn = g[n];
      const float* wr = W + (size_t)n * K;
      float* dwr = dW + (size_t)n * K;
      if (o) {
#pragma omp simd
        for (int k = 0; k < K; ++k) o[k] += gn * wr[k];
      }
#pragma omp simd
      for (int k = 0; k < K; ++k) dwr[k] += gn * xr[k];
      db[n] += gn;
    }
  }
}

// _scaled_dot_product_attention (sasrec/modules.py:21-64) with the head split / merge of multi_head_attention_forward (:457-468, :517):
// o = dropout(softmax(q / sqrt(hd) k^T + causal)) v, head h in columns [h hd, (h+1) hd).  lse[h][i] saved for the backward.
void attn_fwd(const float* q, const float* k, const float* v, int L, int d, int H, const Drop& dr, uint32_t bg, float* o, float* lse, float* srow) {
  const int hd = d / H;
  const float scale = 1.0f / sqrtf((float)hd);
  for (int h = 0; h < H; ++h)
    for (int i = 0; i < L; ++i) {
      const float* qi = q + (size_t)i * d + h * hd;
      float mx = -INFINITY;
      for (int j = 0; j <= i; ++j) {
        const float* kj = k + (size_t)j * d + h * hd;
        float acc = 0.f;
#pragma omp simd reduction(+ : acc)
        for (int c = 0; c < hd; ++c) acc += (qi[c] * scale) * kj[c];
        srow[j] = acc;
        mx = fmaxf(mx, acc);
      }
      float den = 0.f;
      for (int j = 0; j <= i; ++j) { srow[j] = expf(srow[j] - mx); den += srow[j]; }
      lse[(size_t)h * L + i] = mx + logf(den);
      float* oi = o + (size_t)i * d + h * hd;
      for (int c = 0; c < hd; ++c) oi[c] = 0.f;
      const uint32_t base = ((bg * (uint32_t)H + (uint32_t)h) * (uint32_t)L + (uint32_t)i) * (uint32_t)L;
      for (int j = 0; j <= i; ++j) {
        float p = srow[j] / den;
        if (dr.thr) p = keep(dr, base + (uint32_t)j) ? p * dr.scale : 0.f;
        if (p == 0.f) continue;
        const float* vj = v + (size_t)j * d + h * hd;
#pragma omp simd
        for (int c = 0; c < hd; ++c) oi[c] += p * vj[c];
      }
    }
}
// autograd of the above; probabilities recomputed from q, k and lse.  dq / dk / dv are OVERWRITTEN.
void attn_bwd(const float* dO, const float* q, const float* k, const float* v, const float* lse, int L, int d, int H, const Drop& dr, uint32_t bg,
              float* dq, float* dk, float* dv, float* prow, float* drow) {
  const int hd = d / H;
  const float scale = 1.0f / sqrtf((float)hd);
  memset(dq, 0, sizeof(float) * (size_t)L * d);
  memset(dk, 0, sizeof(float) * (size_t)L * d);
  memset(dv, 0, sizeof(float) * (size_t)L * d);
  for (int h = 0; h < H; ++h)
    for (int i = 0; i < L; ++i) {
      const float* qi = q + (size_t)i * d + h * hd;
      const float* gi = dO + (size_t)i * d + h * hd;
      const float ls = lse[(size_t)h * L + i];
      const uint32_t base = ((bg * (uint32_t)H + (uint32_t)h) * (uint32_t)L + (uint32_t)i) * (uint32_t)L;
      float delta = 0.f;
      for (int j = 0; j <= i; ++j) {
        const float* kj = k + (size_t)j * d + h * hd;
        const float* vj = v + (size_t)j * d + h * hd;
        float s = 0.f, dp = 0.f;
#pragma omp simd reduction(+ : s, dp)
        for (int c = 0; c < hd; ++c) { s += (qi[c] * scale) * kj[c]; dp += gi[c] * vj[c]; }
        const float p = expf(s - ls);
        float kp = 1.f;
        if (dr.thr) kp = keep(dr, base + (uint32_t)j) ? dr.scale : 0.f;
        const float pd = p * kp;                  // dropped probability (forward's P~)
        float* dvj = dv + (size_t)j * d + h * hd;
        if (pd != 0.f) {
#pragma omp simd
          for (int c = 0; c < hd; ++c) dvj[c] += pd * gi[c];
        }
        const float dpr = dp * kp;                // gradient w.r.t. the un-dropped probability
        prow[j] = p; drow[j] = dpr;
        delta += dpr * p;
      }
      float* dqi = dq + (size_t)i * d + h * hd;
      for (int j = 0; j <= i; ++j) {
        const float ds = prow[j] * (drow[j] - delta);
        if (ds == 0.f) continue;
        const float* kj = k + (size_t)j * d + h * hd;
        float* dkj = dk + (size_t)j * d + h * hd;
#pragma omp simd
        for (int c = 0; c < hd; ++c) { dqi[c] += ds * scale * kj[c]; dkj[c] += ds * scale * qi[c]; }
      }
    }
}

// PointWiseFeedForward without the residual (sasrec/modules.py:629): f = dropout2(conv2(relu(dropout1(conv1(x))))); u = relu(...) saved
void ffn_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, int L, int d, const Drop& d1, const Drop& d2, uint32_t rowg,
             float* u, float* f) {
  lin_fwd(x, W1, b1, L, d, d, u);
  for (int l = 0; l < L; ++l)
    for (int c = 0; c < d; ++c) {
      float t = u[(size_t)l * d + c];
      if (d1.thr) t = keep(d1, (rowg + (uint32_t)l) * (uint32_t)d + (uint32_t)c) ? t * d1.scale : 0.f;
      u[(size_t)l * d + c] = t > 0.f ? t : 0.f;
    }
  lin_fwd(u, W2, b2, L, d, d, f);
  if (d2.thr)
    for (int l = 0; l < L; ++l)
      for (int c = 0; c < d; ++c) {
        float& t = f[(size_t)l * d + c];
        t = keep(d2, (rowg + (uint32_t)l) * (uint32_t)d + (uint32_t)c) ? t * d2.scale : 0.f;
      }
}
// df: gradient of f (consumed / overwritten); dx (+)= gradient of the FFN input
void ffn_bwd(float* df, const float* x, const float* u, const float* W1, const float* W2, int L, int d, const Drop& d1, const Drop& d2, uint32_t rowg,
             float* dx, bool acc, float* dW1, float* db1, float* dW2, float* db2, float* du) {
  if (d2.thr)
    for (int l = 0; l < L; ++l)
      for (int c = 0; c < d; ++c) {
        float& t = df[(size_t)l * d + c];
        t = keep(d2, (rowg + (uint32_t)l) * (uint32_t)d + (uint32_t)c) ? t * d2.scale : 0.f;
      }
  lin_bwd(df, u, W2, L, d, d, du, false, dW2, db2);
  for (int l = 0; l < L; ++l)
    for (int c = 0; c < d; ++c) {
      float t = u[(size_t)l * d + c] > 0.f ? du[(size_t)l * d + c] : 0.f;
      if (d1.thr) t = keep(d1, (rowg + (uint32_t)l) * (uint32_t)d + (uint32_t)c) ? t * d1.scale : 0.f;
      du[(size_t)l * d + c] = t;
    }
  lin_bwd(du, x, W1, L, d, d, dx, acc, dW1, db1);
}

// x = dropout(E[ids] sqrt(d) + P[0..L-1]) * (ids != 0)                      sasrec/model.py:34-41 (and decode(): 53-59)
void embed_fwd(const int32_t* ids, const float* E, const float* P, int L, int d, const Drop& dr, uint32_t rowg, float* x) {
  const float sc = sqrtf((float)d);
  for (int l = 0; l < L; ++l) {
    const int id = ids[l];
    float* xr = x + (size_t)l * d;
    if (id == 0) { for (int c = 0; c < d; ++c) xr[c] = 0.f; continue; }
    for (int c = 0; c < d; ++c) {
      float t = E[(size_t)id * d + c] * sc + P[(size_t)l * d + c];
      if (dr.thr) t = keep(dr, (rowg + (uint32_t)l) * (uint32_t)d + (uint32_t)c) ? t * dr.scale : 0.f;
      xr[c] = t;
    }
  }
}
void embed_bwd(const int32_t* ids, const float* dx, int L, int d, const Drop& dr, uint32_t rowg, float* dE, float* dP) {
  const float sc = sqrtf((float)d);
  for (int l = 0; l < L; ++l) {
    const int id = ids[l];
    if (id == 0) continue;
    for (int c = 0; c < d; ++c) {
      float g = dx[(size_t)l * d + c];
      if (dr.thr) g = keep(dr, (rowg + (uint32_t)l) * (uint32_t)d + (uint32_t)c) ? g * dr.scale : 0.f;
      dP[(size_t)l * d + c] += g;
      dE[(size_t)id * d + c] += g * sc;
    }
  }
}

struct Scratch {      // per thread
  std::vector<float> a, b, c, e, f, g, row1, row2;
  void size(int L, int d) {
    const size_t n = (size_t)L * d;
    for (auto* v : {&a, &b, &c, &e, &f, &g}) v->assign(n, 0.f);
    row1.assign(L, 0.f); row2.assign(L, 0.f);
  }
};

// ---- one sequence, forward ------------------------------------------------------------------------------------------------------------
// EncoderLayer.forward (sasrec/modules.py:644-655): Q = LN1(x); q from Q, k / v from raw x; h = Q + out_proj(attn); h2 = LN2(h);
// y = (h2 + FFN(h2)) * mask; rec = log_softmax(SparseInputLinear(o))
void enc_layer_fwd(const Cfg& c, const float* P, const EncP& e, int layer, const int32_t* ids, const float* x, float* y, float* rec, int B, int b, float* sv,
                   const WS& w, float p, uint32_t seed, uint32_t bg, Scratch& s) {
  const int L = c.maxlen, d = c.hidden, H = c.num_heads, hd = d / H;
  float *q = sv, *k = sv + w.Ld, *v = sv + 2 * w.Ld, *o = sv + 3 * w.Ld, *h = sv + 4 * w.Ld, *u = sv + 5 * w.Ld, *lse = sv + 6 * w.Ld;
  float* Q = s.a.data();
  ln_fwd(x, P + e.ln1w, P + e.ln1b, L, d, Q, nullptr, nullptr);
  lin_fwd(Q, P + e.inw, P + e.inb, L, d, d, q);                                    // _in_projection_packed, "k is v" branch (:123-130)
  lin_fwd(x, P + e.inw + (size_t)d * d, P + e.inb + d, L, d, d, k);
  lin_fwd(x, P + e.inw + (size_t)2 * d * d, P + e.inb + 2 * d, L, d, d, v);
  attn_fwd(q, k, v, L, d, H, make_drop(p, seed, enc_site(layer, 0)), bg, o, lse, s.row1.data());
  if (rec) {                                                                       // :648-649, 679-703; stored in the reference's row order l*B + b (:518)
    for (int l = 0; l < L; ++l)
      for (int hh = 0; hh < H; ++hh) {
        float z[16], mx = -INFINITY;
        for (int cc = 0; cc < H; ++cc) {
          float acc = P[e.sb + cc];
          for (int j = 0; j < hd; ++j) acc += o[(size_t)l * d + hh * hd + j] * P[e.sw + (size_t)cc * hd + j];
          z[cc] = acc; mx = fmaxf(mx, acc);
        }
        float se = 0.f;
        for (int cc = 0; cc < H; ++cc) se += expf(z[cc] - mx);
        const float lz = mx + logf(se);
        for (int cc = 0; cc < H; ++cc) rec[(((size_t)l * B + b) * H + hh) * H + cc] = z[cc] - lz;
      }
  }
  float* a = s.b.data();
  lin_fwd(o, P + e.ow, P + e.ob, L, d, d, a);
  for (size_t i = 0; i < w.Ld; ++i) h[i] = Q[i] + a[i];                            // the residual adds LN1(x) (:650-651)
  float* h2 = s.c.data();
  ln_fwd(h, P + e.ln2w, P + e.ln2b, L, d, h2, nullptr, nullptr);
  float* f = s.e.data();
  ffn_fwd(h2, P + e.c1w, P + e.c1b, P + e.c2w, P + e.c2b, L, d, make_drop(p, seed, enc_site(layer, 1)), make_drop(p, seed, enc_site(layer, 2)),
          bg * (uint32_t)L, u, f);
  for (int l = 0; l < L; ++l)
    for (int cc = 0; cc < d; ++cc) y[(size_t)l * d + cc] = ids[l] ? h2[(size_t)l * d + cc] + f[(size_t)l * d + cc] : 0.f;
}

// DecoderLayer.forward (sasrec/modules.py:666-677): D = LN(x); a1 = MHA_slf(D, D, D) (no residual); a2 = MHA_enc(a1, enc, enc) (causal);
// y = (D + a2 + FFN(a2)) * mask.  torch.nn.MultiheadAttention = packed in-projection + SDPA + out_proj.
void dec_layer_fwd(const Cfg& c, const float* P, const DecP& e, int layer, const int32_t* ids, const float* x, const float* enc, float* y, float* sv, const WS& w,
                   float p, uint32_t seed, uint32_t bg, Scratch& s) {
  const int L = c.maxlen, d = c.hidden, H = c.num_heads;
  float *q1 = sv, *k1 = sv + w.Ld, *v1 = sv + 2 * w.Ld, *o1 = sv + 3 * w.Ld, *a1 = sv + 4 * w.Ld, *q2 = sv + 5 * w.Ld, *k2 = sv + 6 * w.Ld, *v2 = sv + 7 * w.Ld,
        *o2 = sv + 8 * w.Ld, *a2 = sv + 9 * w.Ld, *u = sv + 10 * w.Ld, *lse1 = sv + 11 * w.Ld, *lse2 = lse1 + w.HL;
  float* D = s.a.data();
  ln_fwd(x, P + e.lnw, P + e.lnb, L, d, D, nullptr, nullptr);
  lin_fwd(D, P + e.sinw, P + e.sinb, L, d, d, q1);
  lin_fwd(D, P + e.sinw + (size_t)d * d, P + e.sinb + d, L, d, d, k1);
  lin_fwd(D, P + e.sinw + (size_t)2 * d * d, P + e.sinb + 2 * d, L, d, d, v1);
  attn_fwd(q1, k1, v1, L, d, H, make_drop(p, seed, dec_site(layer, 0)), bg, o1, lse1, s.row1.data());
  lin_fwd(o1, P + e.sow, P + e.sob, L, d, d, a1);
  lin_fwd(a1, P + e.einw, P + e.einb, L, d, d, q2);
  lin_fwd(enc, P + e.einw + (size_t)d * d, P + e.einb + d, L, d, d, k2);
  lin_fwd(enc, P + e.einw + (size_t)2 * d * d, P + e.einb + 2 * d, L, d, d, v2);
  attn_fwd(q2, k2, v2, L, d, H, make_drop(p, seed, dec_site(layer, 1)), bg, o2, lse2, s.row1.data());
  lin_fwd(o2, P + e.eow, P + e.eob, L, d, d, a2);
  float* f = s.b.data();
  ffn_fwd(a2, P + e.c1w, P + e.c1b, P + e.c2w, P + e.c2b, L, d, make_drop(p, seed, dec_site(layer, 2)), make_drop(p, seed, dec_site(layer, 3)), bg * (uint32_t)L, u,
          f);
  for (int l = 0; l < L; ++l)
    for (int cc = 0; cc < d; ++cc) y[(size_t)l * d + cc] = ids[l] ? D[(size_t)l * d + cc] + a2[(size_t)l * d + cc] + f[(size_t)l * d + cc] : 0.f;
}

// ---- one sequence, backward -------------------------------------------------------------------------------------------------------------
// dy: gradient of the layer output (consumed); dx (+)= gradient of the layer input; drec: gradient of rec (reference row order) or null
void enc_layer_bwd(const Cfg& c, const float* P, float* G, const EncP& e, int layer, const int32_t* ids, const float* x, float* dy, const float* drec, int B, int b,
                   float* dx, const float* sv, const WS& w, float p, uint32_t seed, uint32_t bg, Scratch& s) {
  const int L = c.maxlen, d = c.hidden, H = c.num_heads, hd = d / H;
  const float *q = sv, *k = sv + w.Ld, *v = sv + 2 * w.Ld, *o = sv + 3 * w.Ld, *h = sv + 4 * w.Ld, *u = sv + 5 * w.Ld, *lse = sv + 6 * w.Ld;
  for (int l = 0; l < L; ++l)
    if (!ids[l]) for (int cc = 0; cc < d; ++cc) dy[(size_t)l * d + cc] = 0.f;
  float* h2 = s.a.data();
  ln_fwd(h, P + e.ln2w, P + e.ln2b, L, d, h2, nullptr, nullptr);
  float* dh2 = s.b.data();                     // y = h2 + f: dh2 = g + FFN'(g)
  memcpy(dh2, dy, sizeof(float) * w.Ld);
  ffn_bwd(dy, h2, u, P + e.c1w, P + e.c2w, L, d, make_drop(p, seed, enc_site(layer, 1)), make_drop(p, seed, enc_site(layer, 2)), bg * (uint32_t)L, dh2, true,
          G + e.c1w, G + e.c1b, G + e.c2w, G + e.c2b, s.c.data());
  float* dh = s.c.data();
  ln_bwd(dh2, h, P + e.ln2w, L, d, dh, false, G + e.ln2w, G + e.ln2b);
  float* dO = s.e.data();                      // h = Q + out_proj(o)
  lin_bwd(dh, o, P + e.ow, L, d, d, dO, false, G + e.ow, G + e.ob);
  if (drec) {                                  // rec = log_softmax(z): dz = drec - softmax(z) sum(drec)
    for (int l = 0; l < L; ++l)
      for (int hh = 0; hh < H; ++hh) {
        float z[16], mx = -INFINITY, sd = 0.f;
        const float* dr = drec + (((size_t)l * B + b) * H + hh) * H;
        for (int cc = 0; cc < H; ++cc) {
          float acc = P[e.sb + cc];
          for (int j = 0; j < hd; ++j) acc += o[(size_t)l * d + hh * hd + j] * P[e.sw + (size_t)cc * hd + j];
          z[cc] = acc; mx = fmaxf(mx, acc); sd += dr[cc];
        }
        float se = 0.f;
        for (int cc = 0; cc < H; ++cc) se += expf(z[cc] - mx);
        for (int cc = 0; cc < H; ++cc) {
          const float dz = dr[cc] - expf(z[cc] - mx) / se * sd;
          G[e.sb + cc] += dz;
          for (int j = 0; j < hd; ++j) {
            G[e.sw + (size_t)cc * hd + j] += dz * o[(size_t)l * d + hh * hd + j];
            dO[(size_t)l * d + hh * hd + j] += dz * P[e.sw + (size_t)cc * hd + j];
          }
        }
      }
  }
  float *dq = s.f.data(), *dk = s.g.data(), *dv = s.b.data();      // dh2 (s.b) is dead by now
  attn_bwd(dO, q, k, v, lse, L, d, H, make_drop(p, seed, enc_site(layer, 0)), bg, dq, dk, dv, s.row1.data(), s.row2.data());
  float* Q = s.a.data();
  ln_fwd(x, P + e.ln1w, P + e.ln1b, L, d, Q, nullptr, nullptr);
  float* dQ = dh;                              // residual: dQ = dh + dq Wq
  lin_bwd(dq, Q, P + e.inw, L, d, d, dQ, true, G + e.inw, G + e.inb);
  lin_bwd(dk, x, P + e.inw + (size_t)d * d, L, d, d, dx, true, G + e.inw + (size_t)d * d, G + e.inb + d);
  lin_bwd(dv, x, P + e.inw + (size_t)2 * d * d, L, d, d, dx, true, G + e.inw + (size_t)2 * d * d, G + e.inb + 2 * d);
  ln_bwd(dQ, x, P + e.ln1w, L, d, dx, true, G + e.ln1w, G + e.ln1b);
}

void dec_layer_bwd(const Cfg& c, const float* P, float* G, const DecP& e, int layer, const int32_t* ids, const float* x, const float* enc, float* dy, float* dx,
                   float* denc, const float* sv, const WS& w, float p, uint32_t seed, uint32_t bg, Scratch& s) {
  const int L = c.maxlen, d = c.hidden, H = c.num_heads;
  const float *q1 = sv, *k1 = sv + w.Ld, *v1 = sv + 2 * w.Ld, *o1 = sv + 3 * w.Ld, *a1 = sv + 4 * w.Ld, *q2 = sv + 5 * w.Ld, *k2 = sv + 6 * w.Ld, *v2 = sv + 7 * w.Ld,
              *o2 = sv + 8 * w.Ld, *a2 = sv + 9 * w.Ld, *u = sv + 10 * w.Ld, *lse1 = sv + 11 * w.Ld, *lse2 = lse1 + w.HL;
  for (int l = 0; l < L; ++l)
    if (!ids[l]) for (int cc = 0; cc < d; ++cc) dy[(size_t)l * d + cc] = 0.f;
  float* dD = s.a.data();                      // y = D + a2 + f
  memcpy(dD, dy, sizeof(float) * w.Ld);
  float* da2 = s.b.data();
  memcpy(da2, dy, sizeof(float) * w.Ld);
  ffn_bwd(dy, a2, u, P + e.c1w, P + e.c2w, L, d, make_drop(p, seed, dec_site(layer, 2)), make_drop(p, seed, dec_site(layer, 3)), bg * (uint32_t)L, da2, true,
          G + e.c1w, G + e.c1b, G + e.c2w, G + e.c2b, s.c.data());
  float* dO2 = s.c.data();
  lin_bwd(da2, o2, P + e.eow, L, d, d, dO2, false, G + e.eow, G + e.eob);
  float *dq = s.e.data(), *dk = s.f.data(), *dv = s.g.data();
  attn_bwd(dO2, q2, k2, v2, lse2, L, d, H, make_drop(p, seed, dec_site(layer, 1)), bg, dq, dk, dv, s.row1.data(), s.row2.data());
  float* da1 = s.b.data();
  lin_bwd(dq, a1, P + e.einw, L, d, d, da1, false, G + e.einw, G + e.einb);
  lin_bwd(dk, enc, P + e.einw + (size_t)d * d, L, d, d, denc, true, G + e.einw + (size_t)d * d, G + e.einb + d);
  lin_bwd(dv, enc, P + e.einw + (size_t)2 * d * d, L, d, d, denc, true, G + e.einw + (size_t)2 * d * d, G + e.einb + 2 * d);
  float* dO1 = s.c.data();
  lin_bwd(da1, o1, P + e.sow, L, d, d, dO1, false, G + e.sow, G + e.sob);
  attn_bwd(dO1, q1, k1, v1, lse1, L, d, H, make_drop(p, seed, dec_site(layer, 0)), bg, dq, dk, dv, s.row1.data(), s.row2.data());
  float* D = s.b.data();
  ln_fwd(x, P + e.lnw, P + e.lnb, L, d, D, nullptr, nullptr);
  lin_bwd(dq, D, P + e.sinw, L, d, d, dD, true, G + e.sinw, G + e.sinb);
  lin_bwd(dk, D, P + e.sinw + (size_t)d * d, L, d, d, dD, true, G + e.sinw + (size_t)d * d, G + e.sinb + d);
  lin_bwd(dv, D, P + e.sinw + (size_t)2 * d * d, L, d, d, dD, true, G + e.sinw + (size_t)2 * d * d, G + e.sinb + 2 * d);
  ln_bwd(dD, x, P + e.lnw, L, d, dx, false, G + e.lnw, G + e.lnb);
}

int g_threads = 0;
inline int nthreads() {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}
inline float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

}  // namespace

extern "C" {

int adt_cpu_version(void) { return 1; }
void adt_cpu_set_threads(int n) { g_threads = n; }
int adt_cpu_threads(void) { return nthreads(); }

// offsets[4 + 30 * num_layers]: flat float offset of every state_dict tensor, in the slot order of adt_sasrec_param_layout; returns the total
int64_t adt_cpu_sasrec_param_layout(const Cfg* c, int64_t* offsets) {
  const Layout lo = make_layout(*c);
  if (offsets) {
    int k = 0;
    offsets[k++] = lo.item; offsets[k++] = lo.pos; offsets[k++] = lo.lastw; offsets[k++] = lo.lastb;
    for (const EncP& e : lo.enc)
      for (size_t v : {e.ln1w, e.ln1b, e.inw, e.inb, e.ow, e.ob, e.ln2w, e.ln2b, e.c1w, e.c1b, e.c2w, e.c2b, e.sw, e.sb}) offsets[k++] = (int64_t)v;
    for (const DecP& e : lo.dec)
      for (size_t v : {e.lnw, e.lnb, e.sinw, e.sinb, e.sow, e.sob, e.einw, e.einb, e.eow, e.eob, e.c1w, e.c1b, e.c2w, e.c2b, e.uw, e.ub}) offsets[k++] = (int64_t)v;
  }
  return (int64_t)lo.total;
}
int64_t adt_cpu_sasrec_workspace_floats(const Cfg* c, int B) { return (int64_t)make_ws(*c, B).total; }
int64_t adt_cpu_sasrec_ws_offset(const Cfg* c, int B, int what, int layer) {
  const WS w = make_ws(*c, B);
  switch (what) {
    case WS_ENC_X: return (int64_t)(w.enc_x + (size_t)layer * w.Td);
    case WS_DEC_X: return (int64_t)(w.dec_x + (size_t)layer * w.Td);
    case WS_REC: return (int64_t)(w.recs + (size_t)layer * w.rec);
    case WS_POS_LOGITS: return (int64_t)w.posl;
    case WS_NEG_LOGITS: return (int64_t)w.negl;
    case WS_F: return (int64_t)w.f;
    case WS_G_ENC_X: return (int64_t)(w.g_enc_x + (size_t)layer * w.Td);
    case WS_G_DEC_X: return (int64_t)(w.g_dec_x + (size_t)layer * w.Td);
    case WS_G_REC: return (int64_t)(w.g_rec + (size_t)layer * w.rec);
    case WS_G_POS: return (int64_t)w.g_pos;
    case WS_G_NEG: return (int64_t)w.g_neg;
    case WS_LOSS: return (int64_t)w.loss;
    case WS_NORMS: return (int64_t)w.norms;
  }
  return -1;
}

// SASRecADT.forward (sasrec/model.py:67-81): ENC_X[i] = input of encoder layer i (ENC_X[nl] = encoder output), F = last_layernorm(ENC_X[nl]),
// DEC_X[0] = decoder embedding, DEC_X[i+1] = output of decoder layer i, REC[i] in the reference's row order, POS / NEG logits.
// enc_only != 0: log2feats only (predict).  `stream` is ignored (signature twin of adt_sasrec_forward).
int adt_cpu_sasrec_forward(const Cfg* c, const float* P, float* ws, const int32_t* seq, const int32_t* dec, const int32_t* pos, const int32_t* neg, int B,
                           int training, const uint32_t* seed, uint32_t b_offset, void* stream) {
  (void)stream;
  const Cfg& cf = *c;
  const Layout lo = make_layout(cf);
  const WS w = make_ws(cf, B);
  const int L = cf.maxlen, d = cf.hidden, nl = cf.num_layers, H = cf.num_heads;
  if (d % H || H > 16) return -1;
  const float p = training ? cf.dropout : 0.f;
  const uint32_t sd = seed ? *seed : 0u;
  const bool enc_only = dec == nullptr;
#pragma omp parallel num_threads(nthreads())
  {
    Scratch s;
    s.size(L, d);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      const uint32_t bg = (uint32_t)b + b_offset;
      const size_t r0 = (size_t)b * L;
      float* sv = ws + w.save + (size_t)b * w.save_stride;
      embed_fwd(seq + r0, P + lo.item, P + lo.pos, L, d, make_drop(p, sd, SITE_EMB_SEQ), bg * (uint32_t)L, ws + w.enc_x + r0 * d);
      for (int i = 0; i < nl; ++i)
        enc_layer_fwd(cf, P, lo.enc[i], i, seq + r0, ws + w.enc_x + (size_t)i * w.Td + r0 * d, ws + w.enc_x + (size_t)(i + 1) * w.Td + r0 * d,
                      H > 1 ? ws + w.recs + (size_t)i * w.rec : nullptr, B, b, sv + (size_t)i * w.enc_sz, w, p, sd, bg, s);
      float* f = ws + w.f + r0 * d;
      ln_fwd(ws + w.enc_x + (size_t)nl * w.Td + r0 * d, P + lo.lastw, P + lo.lastb, L, d, f, nullptr, nullptr);                 // model.py:48
      if (enc_only) continue;
      embed_fwd(dec + r0, P + lo.item, P + lo.pos, L, d, make_drop(p, sd, SITE_EMB_DEC), bg * (uint32_t)L, ws + w.dec_x + r0 * d);
      for (int i = 0; i < nl; ++i)
        dec_layer_fwd(cf, P, lo.dec[i], i, dec + r0, ws + w.dec_x + (size_t)i * w.Td + r0 * d, f, ws + w.dec_x + (size_t)(i + 1) * w.Td + r0 * d,
                      sv + (size_t)nl * w.enc_sz + (size_t)i * w.dec_sz, w, p, sd, bg, s);
      for (int l = 0; l < L; ++l) {                                                                                             // model.py:72-76
        const float* fr = f + (size_t)l * d;
        const float* pe = P + lo.item + (size_t)pos[r0 + l] * d;
        const float* ne = P + lo.item + (size_t)neg[r0 + l] * d;
        float ap = 0.f, an = 0.f;
        for (int cc = 0; cc < d; ++cc) { ap += fr[cc] * pe[cc]; an += fr[cc] * ne[cc]; }
        ws[w.posl + r0 + l] = ap; ws[w.negl + r0 + l] = an;
      }
    }
  }
  return 0;
}

// Loss assembly of sasrec/main.py:146-169 and its seeds.  norms = (n_bce, n_mse, n_nll) of the GLOBAL batch must be in NORMS.
// LOSS slots: [0] BCE pos, [1] BCE neg, [2 + i] MSE_i, [2 + nl + l] NLL_l (un-weighted terms).  The NLL weight is lambdas2[nl-1] (stale index, :169).
int adt_cpu_sasrec_loss_seed(const Cfg* c, float* ws, const int32_t* pos, int B, const float* lambdas1, const float* lambdas2, void* stream) {
  (void)stream;
  const Cfg& cf = *c;
  const WS w = make_ws(cf, B);
  const int nl = cf.num_layers, H = cf.num_heads;
  const float n_bce = ws[w.norms], n_mse = ws[w.norms + 1], n_nll = ws[w.norms + 2];
  double bp = 0.0, bn = 0.0;
#pragma omp parallel for reduction(+ : bp, bn) num_threads(nthreads())
  for (int64_t t = 0; t < (int64_t)w.T; ++t) {
    const float m = pos[t] != 0 ? 1.f : 0.f;
    const float pl = ws[w.posl + t], ng = ws[w.negl + t];
    bp += softplus(-pl) * m; bn += softplus(ng) * m;                                   // BCEWithLogits, targets 1 / 0 over pos != 0 (:151-153)
    ws[w.g_pos + t] = (1.0f / (1.0f + expf(-pl)) - 1.0f) * m / n_bce;
    ws[w.g_neg + t] = (1.0f / (1.0f + expf(-ng))) * m / n_bce;
  }
  ws[w.loss] = (float)(bp / n_bce); ws[w.loss + 1] = (float)(bn / n_bce);
  for (int i = 0; i < nl; ++i) {                                                        // lambda1[i] MSE(enc_in[i], dec_out_rev[i]) over ALL elements (:155-158)
    const float* A = ws + w.enc_x + (size_t)i * w.Td;
    const float* Bm = ws + w.dec_x + (size_t)(nl - i) * w.Td;
    float* GA = ws + w.g_enc_x + (size_t)i * w.Td;
    float* GB = ws + w.g_dec_x + (size_t)(nl - i) * w.Td;
    const float cg = 2.0f * lambdas1[i] / n_mse;
    double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) num_threads(nthreads())
    for (int64_t t = 0; t < (int64_t)w.Td; ++t) {
      const float df = A[t] - Bm[t];
      acc += (double)df * df;
      GA[t] = cg * df; GB[t] = -cg * df;
    }
    ws[w.loss + 2 + i] = (float)(acc / n_mse);
  }
  memset(ws + w.g_enc_x + (size_t)nl * w.Td, 0, sizeof(float) * w.Td);
  memset(ws + w.g_dec_x, 0, sizeof(float) * w.Td);
  for (int l = 0; l < nl; ++l) {                                                        // -mean_{token,h} rec[token,h,h] (:160-169)
    double acc = 0.0;
    if (H > 1) {
      const float* R = ws + w.recs + (size_t)l * w.rec;
      float* GR = ws + w.g_rec + (size_t)l * w.rec;
      const float cg = -lambdas2[nl - 1] / n_nll;
      for (size_t r = 0; r < w.T * H; ++r) {
        const int h = (int)(r % H);
        for (int cc = 0; cc < H; ++cc) GR[r * H + cc] = cc == h ? cg : 0.f;
        acc += -(double)R[r * H + h];
      }
    }
    ws[w.loss + 2 + nl + l] = (float)(acc / n_nll);
  }
  return 0;
}

// Reverse pass: consumes the G_* buffers, ACCUMULATES into grads (flat layout of params).  phase is ignored (always everything).
int adt_cpu_sasrec_backward(const Cfg* c, const float* P, float* G, float* ws, const int32_t* seq, const int32_t* dec, const int32_t* pos, const int32_t* neg,
                            int B, int training, const uint32_t* seed, uint32_t b_offset, int phase, void* stream) {
  (void)stream; (void)phase;
  const Cfg& cf = *c;
  const Layout lo = make_layout(cf);
  const WS w = make_ws(cf, B);
  const int L = cf.maxlen, d = cf.hidden, nl = cf.num_layers, H = cf.num_heads;
  const float p = training ? cf.dropout : 0.f;
  const uint32_t sd = seed ? *seed : 0u;
  const int nt = std::min(nthreads(), std::max(B, 1));
  float* priv = static_cast<float*>(malloc(sizeof(float) * (size_t)nt * lo.total));      // zeroed by its owner thread (first touch)
  if (!priv) return -2;
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    float* Gt = priv + (size_t)omp_get_thread_num() * lo.total;
#else
    float* Gt = priv;
#endif
    memset(Gt, 0, sizeof(float) * lo.total);
#pragma omp barrier
    Scratch s;
    s.size(L, d);
    std::vector<float> df((size_t)L * d), dx((size_t)L * d), dyb((size_t)L * d);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      const uint32_t bg = (uint32_t)b + b_offset;
      const size_t r0 = (size_t)b * L;
      const float* sv = ws + w.save + (size_t)b * w.save_stride;
      const float* f = ws + w.f + r0 * d;
      // logits: df = dpos E[pos] + dneg E[neg]; dE[pos] += dpos f; dE[neg] += dneg f
      for (int l = 0; l < L; ++l) {
        const float gp = ws[w.g_pos + r0 + l], gn = ws[w.g_neg + r0 + l];
        const int ip = pos[r0 + l], in = neg[r0 + l];
        for (int cc = 0; cc < d; ++cc) {
          df[(size_t)l * d + cc] = gp * P[lo.item + (size_t)ip * d + cc] + gn * P[lo.item + (size_t)in * d + cc];
          Gt[lo.item + (size_t)ip * d + cc] += gp * f[(size_t)l * d + cc];
          Gt[lo.item + (size_t)in * d + cc] += gn * f[(size_t)l * d + cc];
        }
      }
      // decoder stack, last layer first; G_DEC_X[i+1] is the seed on the output of layer i
      for (int i = nl - 1; i >= 0; --i) {
        float* gy = ws + w.g_dec_x + (size_t)(i + 1) * w.Td + r0 * d;
        if (i < nl - 1) for (size_t t = 0; t < w.Ld; ++t) gy[t] += dx[t];
        dec_layer_bwd(cf, P, Gt, lo.dec[i], i, dec + r0, ws + w.dec_x + (size_t)i * w.Td + r0 * d, f, gy, dx.data(), df.data(),
                      sv + (size_t)nl * w.enc_sz + (size_t)i * w.dec_sz, w, p, sd, bg, s);
      }
      embed_bwd(dec + r0, dx.data(), L, d, make_drop(p, sd, SITE_EMB_DEC), bg * (uint32_t)L, Gt + lo.item, Gt + lo.pos);
      // last LayerNorm, then the encoder stack; G_ENC_X[i] is the seed on the INPUT of layer i (the reconstruction term)
      ln_bwd(df.data(), ws + w.enc_x + (size_t)nl * w.Td + r0 * d, P + lo.lastw, L, d, dyb.data(), false, Gt + lo.lastw, Gt + lo.lastb);
      for (int i = nl - 1; i >= 0; --i) {
        float* gx = ws + w.g_enc_x + (size_t)i * w.Td + r0 * d;           // dx accumulates on top of the seed
        enc_layer_bwd(cf, P, Gt, lo.enc[i], i, seq + r0, ws + w.enc_x + (size_t)i * w.Td + r0 * d, dyb.data(),
                      H > 1 ? ws + w.g_rec + (size_t)i * w.rec : nullptr, B, b, gx, sv + (size_t)i * w.enc_sz, w, p, sd, bg, s);
        memcpy(dyb.data(), gx, sizeof(float) * w.Ld);
      }
      embed_bwd(seq + r0, dyb.data(), L, d, make_drop(p, sd, SITE_EMB_SEQ), bg * (uint32_t)L, Gt + lo.item, Gt + lo.pos);
    }
    // fold the private accumulators in thread order (deterministic for a given thread count)
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)lo.total; ++i) {
      float acc = 0.f;
      for (int t = 0; t < nt; ++t) acc += priv[(size_t)t * lo.total + i];
      G[i] += acc;
    }
  }
  free(priv);
  return 0;
}

// sasrec/main.py:170-173: G[0..nE) += wd * E / ||E||_F (un-squared Frobenius norm of the item table in the loss), clip_grad_norm_(clip),
// Adam(lr, (b1, b2), eps).  scal (host floats): [0] ||E||^2, [1] ||g||^2, [2] step count (incremented), [3] wd * ||E||.  `skip`: nskip
// (offset, count) pairs of parameters torch leaves at grad None (no moment update, no step): pos_ffn_layernorm, and the head classifier at H = 1.
int adt_cpu_clip_adam(float* P, float* G, float* M, float* V, int64_t n, int64_t nE, float wd, float clip, float lr, float b1, float b2, float eps,
                      float grad_scale, float* scal, void* stream) {
  (void)stream;
  double e2 = 0.0;
#pragma omp parallel for reduction(+ : e2) num_threads(nthreads())
  for (int64_t i = 0; i < nE; ++i) e2 += (double)P[i] * P[i];
  const float nrm = (float)sqrt(e2);
  const float coef = (wd != 0.f && nrm > 0.f) ? wd / nrm : 0.f;
  double g2 = 0.0;
#pragma omp parallel for reduction(+ : g2) num_threads(nthreads())
  for (int64_t i = 0; i < n; ++i) {
    float g = G[i] * grad_scale;
    if (i < nE) g += coef * P[i];
    G[i] = g;
    g2 += (double)g * g;
  }
  scal[0] = (float)e2; scal[1] = (float)g2; scal[2] += 1.0f; scal[3] = wd * nrm;
  const float tn = (float)sqrt(g2);
  const float cc = fminf(1.0f, clip / (tn + 1e-6f));
  const float t = scal[2];
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
#pragma omp parallel for num_threads(nthreads())
  for (int64_t i = 0; i < n; ++i) {
    const float g = G[i] * cc;
    if (g == 0.f && M[i] == 0.f && V[i] == 0.f) continue;      // never-touched parameters (torch: grad None) stay as they are
    M[i] = b1 * M[i] + (1.0f - b1) * g;
    V[i] = b2 * V[i] + (1.0f - b2) * g * g;
    P[i] -= (lr / bc1) * M[i] / (sqrtf(V[i]) / sqrtf(bc2) + eps);
  }
  return 0;
}

// SASRecADT.predict (sasrec/model.py:83-97): encoder only, last position, scores against the candidates (B, C) or the whole table (cand NULL, C = V + 1)
int adt_cpu_sasrec_predict(const Cfg* c, const float* P, float* ws, const int32_t* seq, const int32_t* cand, int B, int C, float* logits, void* stream) {
  const int rc = adt_cpu_sasrec_forward(c, P, ws, seq, nullptr, nullptr, nullptr, B, 0, nullptr, 0, stream);
  if (rc) return rc;
  const Layout lo = make_layout(*c);
  const WS w = make_ws(*c, B);
  const int L = c->maxlen, d = c->hidden;
#pragma omp parallel for num_threads(nthreads())
  for (int b = 0; b < B; ++b) {
    const float* f = ws + w.f + ((size_t)b * L + (L - 1)) * d;
    for (int j = 0; j < C; ++j) {
      const int id = cand ? cand[(size_t)b * C + j] : j;
      float acc = 0.f;
      for (int cc = 0; cc < d; ++cc) acc += f[cc] * P[lo.item + (size_t)id * d + cc];
      logits[(size_t)b * C + j] = acc;
    }
  }
  return 0;
}

}  // extern "C"

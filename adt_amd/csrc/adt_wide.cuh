// Row / elementwise kernels shared by the BERT4Rec-ADT and STOSA-ADT paths: summed embeddings, dropout+activation,
// masked-row gather/scatter, softmax cross-entropy over all items.  HBM-bound: float4 accesses, wave reductions.
#pragma once
#include "adt_common.cuh"
#include "adt_gemm.cuh"

namespace adt {

// X[row] = E[ids[row]] * scale + P[row % L] (+ S0): bert4rec/model/modules.py:42-46 (word + position + sentence 0),
// stosa/models.py:183-210 (item + position).  Unlike sasrec's embedding no padding mask is applied here.
struct EmbedSumArgs {
  const int* ids; const float* E; const float* P; const float* S0;
  float scale; int T, L, d;
  float* X;
};

__global__ __launch_bounds__(256) void k_embed_sum_fwd(EmbedSumArgs a) {
  const int V = a.d / 4;
  const size_t n = (size_t)a.T * V;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / V), c4 = (int)(i % V) * 4;
    const float4 e = *reinterpret_cast<const float4*>(a.E + (size_t)a.ids[row] * a.d + c4);
    const float4 p = *reinterpret_cast<const float4*>(a.P + (size_t)(row % a.L) * a.d + c4);
    float4 v = make_float4(e.x * a.scale + p.x, e.y * a.scale + p.y, e.z * a.scale + p.z, e.w * a.scale + p.w);
    if (a.S0) {
      const float4 s = *reinterpret_cast<const float4*>(a.S0 + c4);
      v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
    }
    *reinterpret_cast<float4*>(a.X + (size_t)row * a.d + c4) = v;
  }
}

// Y = act(dropout(X)) / dX (+)= dY * act'(dropout(X)) * keep / (1 - p); element index idx_offset + i
struct DropActArgs {
  const float* X; const float* dY; float* Y; float* dX;
  size_t n; DropCfg drop; uint32_t idx_offset; int act; int accumulate;
};

template <bool BWD>
__global__ __launch_bounds__(256) void k_dropact(DropActArgs a) {
  const uint32_t key = drop_key(a.drop);
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (size_t)gridDim.x * 1024) {
    float x[4], o[4];
    *reinterpret_cast<float4*>(x) = *reinterpret_cast<const float4*>(a.X + i);
    float dy[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BWD) {
      *reinterpret_cast<float4*>(dy) = *reinterpret_cast<const float4*>(a.dY + i);
      if (a.accumulate) *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(a.dX + i);
      else o[0] = o[1] = o[2] = o[3] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ks = 1.0f;
      if (a.drop.thr) ks = adt_keep(key, a.idx_offset + (uint32_t)(i + j), a.drop.thr) ? a.drop.scale : 0.f;
      const float u = x[j] * ks;
      if constexpr (BWD) o[j] += dy[j] * act_grad(a.act, u) * ks;
      else o[j] = act_apply(a.act, u);
    }
    if constexpr (BWD) *reinterpret_cast<float4*>(a.dX + i) = *reinterpret_cast<float4*>(o);
    else *reinterpret_cast<float4*>(a.Y + i) = *reinterpret_cast<float4*>(o);
  }
}

// out[i] = F[rows[i]] (gather) / dF[rows[i]] (+)= G[i] (scatter; rows are distinct)
struct RowsArgs {
  const float* src; int lds; float* dst; int ldd; const int* rows; int M, d; int scatter; int accumulate;
  const int* m_dev;   // optional DEVICE row count: M = min(M, *m_dev)
};

__global__ __launch_bounds__(256) void k_rows(RowsArgs a) {
  const int V = a.d / 4;
  const int M = (a.m_dev && *a.m_dev < a.M) ? *a.m_dev : a.M;
  const size_t n = (size_t)M * V;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / V), c4 = (int)(i % V) * 4;
    const int r = a.rows[m];
    const size_t so = a.scatter ? (size_t)m * a.lds + c4 : (size_t)r * a.lds + c4;
    const size_t dof = a.scatter ? (size_t)r * a.ldd + c4 : (size_t)m * a.ldd + c4;
    float4 v = *reinterpret_cast<const float4*>(a.src + so);
    if (a.accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(a.dst + dof);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(a.dst + dof) = v;
  }
}

// nn.CrossEntropyLoss(ignore_index=0) over all items (bert4rec/trainer.py:45,113-115): one workgroup per row of the
// (M x V) logits; loss slot += w * (lse - z[label]); logits are overwritten with w * (softmax - onehot), w = *inv_count
// (1 / number of non-ignored labels of the GLOBAL batch); rows with label 0 get zero gradient.
struct CeArgs {
  float* logits; int ld; const int* labels; int M, V; const float* inv_count; float* loss; const int* m_dev;
};

ADT_DEVICE_INLINE float block_max256(float v, float* sbuf) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sbuf[0], sbuf[1]), fmaxf(sbuf[2], sbuf[3]));
  __syncthreads();
  return r;
}
ADT_DEVICE_INLINE float block_sum256(float v, float* sbuf) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sbuf[0] + sbuf[1]) + (sbuf[2] + sbuf[3]);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void k_ce_rows(CeArgs a) {
  __shared__ float sbuf[4];
  const float w = *a.inv_count;
  const int M = (a.m_dev && *a.m_dev < a.M) ? *a.m_dev : a.M;
  for (int m = blockIdx.x; m < M; m += gridDim.x) {
    float* z = a.logits + (size_t)m * a.ld;
    const int label = a.labels[m];
    if (label == 0) {
      for (int j = threadIdx.x; j < a.V; j += 256) z[j] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < a.V; j += 256) mx = fmaxf(mx, z[j]);
    mx = block_max256(mx, sbuf);
    float s = 0.f;
    for (int j = threadIdx.x; j < a.V; j += 256) s += expf(z[j] - mx);
    s = block_sum256(s, sbuf);
    const float lse = mx + logf(s);
    if (threadIdx.x == 0) atomicAdd(a.loss + (m & 63), w * (lse - z[label]));
    __syncthreads();   // z[label] read before it is overwritten
    const float inv = 1.0f / s;
    for (int j = threadIdx.x; j < a.V; j += 256) z[j] = w * (expf(z[j] - mx) * inv - (j == label ? 1.0f : 0.f));
  }
}

// Same arithmetic with the row held in LDS: one global read and one global write per element instead of three reads and a
// write (V = 26,844 at the ml-20m shape: 107 KB per row, 1.3 GB per step instead of 2.6 GB).  1,024 threads per row, 16-byte
// accesses; rows whose V * 4 bytes do not fit fall back to k_ce_rows.
constexpr int CE_NTH = 1024;
__global__ __launch_bounds__(CE_NTH) void k_ce_rows_lds(CeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ce_row[];     // [V rounded up to 4] + 16 reduction slots
  float* red = ce_row + (a.V + 3) / 4 * 4;
  const float w = *a.inv_count;
  const int M = (a.m_dev && *a.m_dev < a.M) ? *a.m_dev : a.M;
  const int V4 = a.V / 4, tid = threadIdx.x, wv = tid >> 6;
  for (int m = blockIdx.x; m < M; m += gridDim.x) {
    float* z = a.logits + (size_t)m * a.ld;
    const int label = a.labels[m];
    if (label == 0) {
      for (int j = tid; j < V4; j += CE_NTH) reinterpret_cast<float4*>(z)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = V4 * 4 + tid; j < a.V; j += CE_NTH) z[j] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int j = tid; j < V4; j += CE_NTH) {
      const float4 v = reinterpret_cast<const float4*>(z)[j];
      reinterpret_cast<float4*>(ce_row)[j] = v;
      mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int j = V4 * 4 + tid; j < a.V; j += CE_NTH) { ce_row[j] = z[j]; mx = fmaxf(mx, z[j]); }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < CE_NTH / 64; ++i) mx = fmaxf(mx, red[i]);
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < V4; j += CE_NTH) {
      float4 v = reinterpret_cast<float4*>(ce_row)[j];
      v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);
      reinterpret_cast<float4*>(ce_row)[j] = v;          // keep exp(z - max): the gradient needs it again
      s += (v.x + v.y) + (v.z + v.w);
    }
    for (int j = V4 * 4 + tid; j < a.V; j += CE_NTH) { const float e = expf(ce_row[j] - mx); ce_row[j] = e; s += e; }
    const float zl = z[label];                            // still the logit: nothing has been written back yet
    s = wave_sum(s);
    if ((tid & 63) == 0) red[wv] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int i = 0; i < CE_NTH / 64; ++i) s += red[i];
    if (tid == 0) atomicAdd(a.loss + (m & 63), w * (mx + logf(s) - zl));
    const float inv = w / s;
    for (int j = tid; j < V4; j += CE_NTH) {
      float4 v = reinterpret_cast<float4*>(ce_row)[j];
      v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
      const int lj = label - 4 * j;
      if (lj == 0) v.x -= w; else if (lj == 1) v.y -= w; else if (lj == 2) v.z -= w; else if (lj == 3) v.w -= w;
      reinterpret_cast<float4*>(z)[j] = v;
    }
    for (int j = V4 * 4 + tid; j < a.V; j += CE_NTH) z[j] = ce_row[j] * inv - (j == label ? w : 0.f);
    __syncthreads();     // the row buffer and the reduction slots are reused by the next row
  }
}

// dst = (beta ? dst : 0) + alpha * src * (ids == NULL || ids[i / d] != 0): candidate mixing of the supernet
// (sasrec/super_modules.py:42-49: sum_k w_k * layer_k(x)) and masked residual gradients.
struct AxpyArgs {
  float* dst; const float* src; float alpha; int beta; size_t n; const int* ids; int d;
};

__global__ __launch_bounds__(256) void k_axpy(AxpyArgs a) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (size_t)gridDim.x * 1024) {
    float4 v = *reinterpret_cast<const float4*>(a.src + i);
    const float m = (a.ids && a.ids[i / a.d] == 0) ? 0.f : a.alpha;
    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
    if (a.beta) {
      const float4 o = *reinterpret_cast<const float4*>(a.dst + i);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(a.dst + i) = v;
  }
}

// log_softmax over rows of H <= 8 values (the second log_softmax of SuperEncoder.forward, sasrec/super_modules.py:49) and
// its backward dX (+)= dY - softmax(X) * sum(dY), with softmax(X) = exp(Y).
struct LsmArgs {
  const float* X; float* Y; const float* dY; float* dX; size_t rows; int H; int accumulate;
};

template <bool BWD>
__global__ __launch_bounds__(256) void k_logsoftmax_rows(LsmArgs a) {
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < a.rows; r += (size_t)gridDim.x * 256) {
    float v[8];
    if constexpr (!BWD) {
      float m = -INFINITY;
      for (int j = 0; j < a.H; ++j) { v[j] = a.X[r * a.H + j]; m = fmaxf(m, v[j]); }
      float s = 0.f;
      for (int j = 0; j < a.H; ++j) s += expf(v[j] - m);
      const float lz = m + logf(s);
      for (int j = 0; j < a.H; ++j) a.Y[r * a.H + j] = v[j] - lz;
    } else {
      float sd = 0.f;
      for (int j = 0; j < a.H; ++j) { v[j] = a.dY[r * a.H + j]; sd += v[j]; }
      for (int j = 0; j < a.H; ++j) {
        const float g = v[j] - expf(a.Y[r * a.H + j]) * sd;
        a.dX[r * a.H + j] = (a.accumulate ? a.dX[r * a.H + j] : 0.f) + g;
      }
    }
  }
}

// ||g||^2 of a flat range into 64 partial slots, and Adam on a flat range with a caller-supplied step count: the supernet's
// optimizer state is per candidate layer (torch.optim.Adam skips parameters whose grad is None -- no moment decay, no
// step increment, no weight decay -- sasrec/evolution.py:109,314-316), while clip_grad_norm_ uses the global norm.
struct RangeOptArgs {
  float* P; float* G; float* M; float* V; size_t n;
  float l2, clip, lr, b1, b2, eps, step;
  const float* gn2_slots;   // 64 partial sums of ||g||^2 over ALL parameters
  float* out64;
  float wd_dec;             // torch.optim.AdamW's decoupled decay: p *= 1 - lr * wd_dec before the update (0 = off)
};

__global__ __launch_bounds__(256) void k_sumsq64(RangeOptArgs a) {
  __shared__ float sbuf[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) acc += a.G[i] * a.G[i];
  const float s = block_sum256(acc, sbuf);
  if (threadIdx.x == 0) atomicAdd(a.out64 + (blockIdx.x & 63), s);
}

__global__ __launch_bounds__(256) void k_adam_range(RangeOptArgs a) {
  __shared__ float sbuf[4];
  float v0 = threadIdx.x < 64 ? a.gn2_slots[threadIdx.x] : 0.f;
  const float gn2 = block_sum256(v0, sbuf);
  const float coef = fminf(1.0f, a.clip / (sqrtf(gn2) + 1e-6f));
  const float bc1 = 1.0f - powf(a.b1, a.step), bc2 = 1.0f - powf(a.b2, a.step);
  const float stepsz = a.lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
    const float g = a.G[i] * coef + a.l2 * a.P[i];
    const float m = a.b1 * a.M[i] + (1.0f - a.b1) * g;
    const float v = a.b2 * a.V[i] + (1.0f - a.b2) * g * g;
    a.M[i] = m;
    a.V[i] = v;
    a.P[i] = a.P[i] * (1.0f - a.lr * a.wd_dec) - stepsz * m / (sqrtf(v) * rs2 + a.eps);
  }
}

// ---- full-sort selection (stosa/trainer.py:598-612) -------------------------------------------------------------------------------
// One workgroup per user row: push the seen items (CSR) to 1e24, then pick the k smallest distances in ascending order.  Each
// thread keeps the minimum of its own strided slice in registers; a round is one block arg-min over the 256 slice minima, after
// which only the winning thread rescans its slice (N/256 elements, L2 resident).  Ties go to the smaller item id.  The row is
// consumed: seen entries become 1e24, selected entries +inf.
struct TopkArgs {
  float* dist; int ld; int B; int N;
  const int32_t* indptr; const int32_t* indices;
  int k; int32_t* out_idx; float* out_val;
};

__device__ __forceinline__ bool topk_less(float v, int i, float bv, int bi) { return v < bv || (v == bv && (unsigned)i < (unsigned)bi); }

__global__ __launch_bounds__(256) void k_topk_masked(TopkArgs a) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* row = a.dist + (size_t)b * a.ld;
  if (a.indptr) {
    for (int j = a.indptr[b] + tid; j < a.indptr[b + 1]; j += 256) {
      int it = a.indices[j];
      if (it >= 0 && it < a.N) row[it] = 1e24f;
    }
    __threadfence_block();
    __syncthreads();
  }
  const float INF = __builtin_inff();
  float bv = INF;
  int bi = 0x7fffffff;
  for (int i = tid; i < a.N; i += 256) {
    float v = row[i];
    if (v < INF && topk_less(v, i, bv, bi)) { bv = v; bi = i; }   // +inf / NaN entries are never selected
  }
  for (int r = 0; r < a.k; ++r) {
    float wv = bv;
    int wi = bi;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      float ov = __shfl_xor(wv, o);
      int oi = __shfl_xor(wi, o);
      if (topk_less(ov, oi, wv, wi)) { wv = ov; wi = oi; }
    }
    if ((tid & 63) == 0) { s_v[tid >> 6] = wv; s_i[tid >> 6] = wi; }
    __syncthreads();
    wv = s_v[0]; wi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (topk_less(s_v[w], s_i[w], wv, wi)) { wv = s_v[w]; wi = s_i[w]; }
    if (tid == 0) { a.out_idx[(size_t)b * a.k + r] = wi; if (a.out_val) a.out_val[(size_t)b * a.k + r] = wv; }
    if (wi != 0x7fffffff && (wi & 255) == tid) {   // my element won: retire it and refresh my slice minimum
      row[wi] = INF;
      bv = INF; bi = 0x7fffffff;
      for (int i = tid; i < a.N; i += 256) {
        float v = (i == wi) ? INF : row[i];
        if (v < INF && topk_less(v, i, bv, bi)) { bv = v; bi = i; }
      }
    }
    __syncthreads();
  }
}

// ---- the upstream gradient of a dense layer pulled back through its epilogue, materialised once ---------------------------------
// out[t][n] = dY[t][n] * (ids[t] != 0) * dropmask((t + row_offset) * N + n) * act'(U[t][n])  (GradSrc::at, adt_gemm.cuh).  For
// layers with an activation the backward GEMMs otherwise re-derive this per output-column tile (weight gradient) and need the
// saved pre-activation as a second register-resident operand (input gradient: twice the passes); one elementwise pass is cheaper.
struct GradSrcOutArgs { GradSrc G; float* out; int ldo; const int* t_dev; };

__global__ __launch_bounds__(256) void k_gradsrc(GradSrcOutArgs a) {
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const int n4 = G.N / 4;
  const size_t total = (size_t)G.T * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / n4), n = (int)(i % n4) * 4;
    *reinterpret_cast<float4*>(a.out + (size_t)t * a.ldo + n) = G.at(t, n);
  }
}

}  // namespace adt

// Model-level executor for SASRec-ADT: the sequence of stage launches that replaces SASRecADT.forward /
// predict (sasrec/model.py:32-97), Encoder/EncoderLayer/Decoder/DecoderLayer (sasrec/modules.py:635-757) and
// their autograd reverse pass, plus the loss seeds of sasrec/main.py:151-169.  Pure host code: it only
// enqueues kernels (via the per-stage C ABI) on the caller's stream -- and, in the one-phase backward, a few scatter / fold
// kernels on a side stream of its own that forks from and joins the caller's stream inside the call (side_stream below).
#include "adt_host.h"
#include <string.h>
#include <mutex>
#include "adt_bwdchain_args.h"
#include "adt_fwdchain_args.h"
#include "adt_seq_args.h"
#include "adt_seqbwd_args.h"

namespace {

constexpr float LN_EPS = 1e-8f;
constexpr int NREPP = 16;          // replicas of the layer-parameter gradients the backward chains flush into (one per XCD)
constexpr int LN_PART_BLOCKS = 1024;     // blocks of the last LayerNorm's backward when it stores per-block sums (adt_layernorm_bwd_parts); 256 blocks ran 22.8 us against 14.5
constexpr int NREP = 16;          // replicas of the item-table gradient (contention relief for popular items)   // sasrec/modules.py:638,640,660 ; sasrec/model.py:28

struct Layout {
  int64_t off[4 + 30 * 16];
  int64_t total;
  int nl;
  int64_t item() const { return off[0]; }
  int64_t posw() const { return off[1]; }
  int64_t lnl_w() const { return off[2]; }
  int64_t lnl_b() const { return off[3]; }
  int64_t enc(int i, int k) const { return off[4 + 14 * i + k]; }
  int64_t dec(int i, int k) const { return off[4 + 14 * nl + 16 * i + k]; }
};
enum { E_LN1W, E_LN1B, E_INW, E_INB, E_OW, E_OB, E_LN2W, E_LN2B, E_C1W, E_C1B, E_C2W, E_C2B, E_SW, E_SB };
enum { D_LNW, D_LNB, D_SINW, D_SINB, D_SOW, D_SOB, D_EINW, D_EINB, D_EOW, D_EOB, D_C1W, D_C1B, D_C2W, D_C2B, D_UW, D_UB };

int64_t up64(int64_t x) { return (x + 63) / 64 * 64; }

// workgroups per sequence of the per-sequence kernels: the largest power of two that keeps B * S within the CUs (ADT_SEQ_SPLIT overrides;
// 1 at B >= 129).  Small batches -- a data-parallel rank's share of a global batch -- otherwise leave most CUs without a workgroup.
int seq_split(int B) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("ADT_SEQ_SPLIT"); forced = e ? atoi(e) : 0; }
  if (forced > 0) return forced;
  int s = 1;
  while (s < 8 && B * s * 2 <= 256) s *= 2;
  return s;
}


bool make_layout(const adt_sasrec_cfg* c, Layout* lo) {
  if (c->num_layers < 1 || c->num_layers > 16) return false;
  const int64_t d = c->hidden, H = c->num_heads, hd = d / H, L = c->maxlen, V = c->item_num;
  lo->nl = c->num_layers;
  int64_t o = 0;
  int n = 0;
  auto put = [&](int64_t sz) { lo->off[n++] = o; o += up64(sz); };
  put((V + 1) * d); put(L * d); put(d); put(d);
  for (int i = 0; i < c->num_layers; ++i) {
    put(d); put(d); put(3 * d * d); put(3 * d); put(d * d); put(d); put(d); put(d);
    put(d * d); put(d); put(d * d); put(d); put(H * hd); put(H);
  }
  for (int i = 0; i < c->num_layers; ++i) {
    put(d); put(d);
    put(3 * d * d); put(3 * d); put(d * d); put(d);
    put(3 * d * d); put(3 * d); put(d * d); put(d);
    put(d * d); put(d); put(d * d); put(d); put(d); put(d);
  }
  lo->total = o;
  return true;
}

// workspace carve-up (offsets in floats)
struct WS {
  int64_t T, d, H, L, nl, B;
  int64_t enc_x, dec_x, f, posl, negl;                                  // (nl+1)*Td each for enc_x/dec_x
  int64_t e_qn, e_qkv, e_o, e_lse, e_h, e_h2, e_u, e_rec, e_mask, e_stride;     // per encoder layer
  int64_t d_dn, d_qkv, d_o1, d_lse1, d_a1, d_q2, d_kv2, d_o2, d_lse2, d_a2, d_u, d_mask1, d_mask2, d_stride;
  int64_t g_enc_x, g_dec_x, g_f, g_pos, g_neg, g_rec;                   // gradients
  int64_t s1, s2, s3, s4, s5;                                           // backward scratch: Td, Td, 3Td, 2Td, Td
  int64_t loss, norms, scal;
  int64_t rep, rep_stride;                                              // item-table gradient replicas
  int64_t prep, prep_stride;                                            // replicas of every other parameter gradient
  int64_t wpack;                                                        // pre-packed bf16 weight images (3 floats per parameter float)
  int64_t part, part_stride;                                            // per-sequence partials of the 64 x 64 weight gradients (bf16 mode, d = 64)
  int64_t isort, isort_n;                                               // scratch of the sorted item-table gradient (adt_item_sort), isort_n int32 (0: not used)
  int64_t vpart, vcall, lnpart, gnpart;                                 // per-workgroup bias / LayerNorm / classifier gradient sums: 5 * nl calls x vcall floats ;
                                                                        // the last LayerNorm's per-block sums ; per-block partials of ||g||^2
  int64_t total;
};

void make_ws(const adt_sasrec_cfg* c, int B, WS* w) {
  w->B = B; w->d = c->hidden; w->H = c->num_heads; w->L = c->maxlen; w->nl = c->num_layers;
  w->T = (int64_t)B * w->L;
  const int64_t Td = up64(w->T * w->d), T1 = up64(w->T), lse = up64((int64_t)B * w->H * w->L);
  const int64_t rec = up64(w->T * w->H * w->H);
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += n; return r; };
  w->enc_x = take((w->nl + 1) * Td); w->dec_x = take((w->nl + 1) * Td); w->f = take(Td);
  w->posl = take(T1); w->negl = take(T1);
  {
    const int64_t s = o;
    w->e_qn = take(Td); w->e_qkv = take(3 * Td); w->e_o = take(Td); w->e_lse = take(lse); w->e_h = take(Td);
    w->e_h2 = take(Td); w->e_u = take(Td); w->e_rec = take(rec); w->e_mask = take(8 * lse);
    w->e_stride = o - s;
    o = s + w->e_stride * w->nl;
  }
  {
    const int64_t s = o;
    w->d_dn = take(Td); w->d_qkv = take(3 * Td); w->d_o1 = take(Td); w->d_lse1 = take(lse); w->d_a1 = take(Td);
    w->d_q2 = take(Td); w->d_kv2 = take(2 * Td); w->d_o2 = take(Td); w->d_lse2 = take(lse); w->d_a2 = take(Td);
    w->d_u = take(Td); w->d_mask1 = take(8 * lse); w->d_mask2 = take(8 * lse);
    w->d_stride = o - s;
    o = s + w->d_stride * w->nl;
  }
  w->g_enc_x = take((w->nl + 1) * Td); w->g_dec_x = take((w->nl + 1) * Td); w->g_f = take(Td);
  w->g_pos = take(T1); w->g_neg = take(T1); w->g_rec = take(rec * w->nl);
  w->s1 = take(Td); w->s2 = take(Td); w->s3 = take(3 * Td); w->s4 = take(2 * Td); w->s5 = take(Td);
  w->loss = take(64 * (2 + 2 * 16)); w->norms = take(64); w->scal = take(192);
  w->rep_stride = up64((int64_t)(c->item_num + 1) * w->d);
  w->rep = take(NREP * w->rep_stride);
  {
    Layout lo;
    make_layout(c, &lo);
    w->prep_stride = up64(lo.total - lo.posw());
    w->prep = take(NREPP * w->prep_stride);
    w->wpack = take(3 * w->prep_stride);
  }
  w->part_stride = (c->prec == ADT_PREC_BF16 && c->hidden == 64) ? (int64_t)w->nl * 16 * 4096 : 0;
  w->part = take((int64_t)B * seq_split(B) * w->part_stride);      // one partial area per WORKGROUP (several per sequence at small batches)
  w->isort_n = (c->hidden == 64 && adt_item_sort_supported(c->item_num + 1)) ? adt_item_sort_work_ints(4, (int)w->T, c->item_num + 1) : 0;
  w->isort = take(up64(w->isort_n));
  w->vcall = (int64_t)B * seq_split(B) * 512;
  w->vpart = take(w->part_stride > 0 ? (5 * w->nl + 1) * w->vcall : 0);      // (+ 1: the last LayerNorm's sums from the first encoder kernel of the backward)
  w->lnpart = take(LN_PART_BLOCKS * 128);
  w->gnpart = take(4096);
  w->total = o;
}

#define CK(call) do { int _r = (call); if (_r) return _r; } while (0)

// Side stream of the backward pass.  The chain kernels are one workgroup per CU and serial by data dependence; the scatter / fold kernels
// beside them (item-table gradient of the logits, decoder input embedding gradient, the ordered sums of the weight-gradient partials) depend
// on little and are bound by memory-side atomics or HBM, so they run on a second stream under the chain kernels: fork = an event recorded on
// the caller's stream that the side stream waits for, join = the reverse.  Inside a stream capture both become edges of the graph.  The
// stream and its events are created on the first call that is not being captured (creating them is not a capturable operation); until then,
// and with ADT_SIDE_STREAM=0, everything stays on the caller's stream.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork_ev[4] = {nullptr, nullptr, nullptr, nullptr}, join_ev[4] = {nullptr, nullptr, nullptr, nullptr};      // [3]: the id sort
};
// ADT_SIDE_STREAM: bit 0 the logits' item rows, bit 1 the decoder's embedding gradient + partial sums, bit 2 the encoder's partial sums
// (default 7, 0 = everything on the caller's stream)
int side_sites() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_SIDE_STREAM"); on = e ? (atoi(e) & 7) : 7; }
  return on;
}
SideStream* side_stream(hipStream_t main) {
  static SideStream g[16];
  static std::mutex mu;                          // first use from two host threads at once (one trainer per thread)
  if (!side_sites()) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream* sd = &g[dev];
  std::lock_guard<std::mutex> lock(mu);
  if (sd->s) return sd;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  for (int i = 0; i < 4; ++i)
    if (hipEventCreateWithFlags(&sd->fork_ev[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sd->join_ev[i], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamDestroy(s);
      return nullptr;
    }
  sd->s = s;
  return sd;
}
// fork in two halves: side_mark records the point of the caller's stream the side work depends on, side_enter makes the side stream wait
// for it and returns the stream to enqueue the side work on (the caller's stream when there is no side stream).  The caller enqueues its
// OWN next kernel between the two: in a captured graph the branch whose node is created first stays on the queue of the parent node, and
// a chain kernel that hops queues pays the cross-queue signal (5-11 us, profiles/r03_side_stream_timeline.txt).
int side_mark(SideStream* sd, int k, void* main) {
  if (!sd) return 0;
  if (hipEventRecord(sd->fork_ev[k], (hipStream_t)main) != hipSuccess) return adt_set_error("backward: side-stream fork %d failed", k);
  return 0;
}
int side_enter(SideStream* sd, int k, void* main, void** out) {
  *out = main;
  if (!sd) return 0;
  if (hipStreamWaitEvent(sd->s, sd->fork_ev[k], 0) != hipSuccess) return adt_set_error("backward: side-stream fork %d failed", k);
  *out = sd->s;
  return 0;
}
int side_join(SideStream* sd, int k, void* main) {
  if (!sd) return 0;
  if (hipEventRecord(sd->join_ev[k], sd->s) != hipSuccess || hipStreamWaitEvent((hipStream_t)main, sd->join_ev[k], 0) != hipSuccess)
    return adt_set_error("backward: side-stream join %d failed", k);
  return 0;
}

// The item-table and positional-table gradients as sorted segmented sums (adt_itemgrad.cuh) instead of float-atomic scatters into replicas:
// deterministic, and no atomics-bound kernels beside the chain kernels.  ADT_ITEM_SORT=1 switches it on (default: the scatters).
bool item_det(const WS& w) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_ITEM_SORT"); on = (e && atoi(e) != 0) ? 1 : 0; }      // opt-in: +12 us per flagship step (0.637 against 0.625 ms: DESIGN.md "item-table gradient")
  return on && w.isort_n > 0;
}

// ADT_LNL_FUSED=0: the model's last LayerNorm is reversed by its own kernel (k_ln_bwd) instead of inside k_seqtt_post_bwd<ENC>
bool lnl_fuse_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_LNL_FUSED"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on != 0;
}

int check_cfg(const adt_sasrec_cfg* c) {
  if (c->hidden != 64) return adt_set_error("sasrec: hidden=%d unsupported in this build (64)", c->hidden);
  if (c->num_heads < 1 || c->hidden % c->num_heads) return adt_set_error("sasrec: bad num_heads");
  const int hd = c->hidden / c->num_heads;
  if (hd != 16 && hd != 32 && hd != 64) return adt_set_error("sasrec: head_dim=%d unsupported (16/32/64)", hd);
  if (c->maxlen > 224) return adt_set_error("sasrec: maxlen=%d > 224 unsupported", c->maxlen);
  if (c->num_layers < 1 || c->num_layers > 16) return adt_set_error("sasrec: num_layers");
  return 0;
}


adt::FwdChainArgs fwd_args(int T, int L, int B, int H, const int32_t* ids, float p, const uint32_t* seed, uint32_t row_offset) {
  adt::FwdChainArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T; a.L = L; a.B = B; a.H = H; a.ids = ids; a.drop = adt_make_drop(p, seed, 0); a.row_offset = row_offset; a.ln_eps = LN_EPS;
  return a;
}

adt::BwdChainArgs bwd_args(int T, int L, int B, const int32_t* ids, float p, const uint32_t* seed, uint32_t row_offset) {
  adt::BwdChainArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T; a.L = L; a.B = B; a.ids = ids; a.drop = adt_make_drop(p, seed, 0); a.row_offset = row_offset; a.ln_eps = LN_EPS;
  return a;
}

// bf16 mode: both LDS images of every 64 x 64 layer weight, written once per forward (the weights do not change until the
// optimizer step that follows the backward), so that every kernel's weight staging is a plain copy
int pack_offsets(const adt_sasrec_cfg* c, const Layout& lo, int* offs) {      // float offsets from lo.posw() of the 64 x 64 blocks ; 0 blocks outside bf16 mode
  if (c->prec != ADT_PREC_BF16) return 0;
  int n = 0;
  const int64_t base = lo.posw();
  const int dd = c->hidden * c->hidden;
  for (int i = 0; i < c->num_layers; ++i) {
    for (int j = 0; j < 3; ++j) offs[n++] = (int)(lo.enc(i, E_INW) + j * dd - base);
    offs[n++] = (int)(lo.enc(i, E_OW) - base); offs[n++] = (int)(lo.enc(i, E_C1W) - base); offs[n++] = (int)(lo.enc(i, E_C2W) - base);
    for (int j = 0; j < 3; ++j) offs[n++] = (int)(lo.dec(i, D_SINW) + j * dd - base);
    for (int j = 0; j < 3; ++j) offs[n++] = (int)(lo.dec(i, D_EINW) + j * dd - base);
    offs[n++] = (int)(lo.dec(i, D_SOW) - base); offs[n++] = (int)(lo.dec(i, D_EOW) - base);
    offs[n++] = (int)(lo.dec(i, D_C1W) - base); offs[n++] = (int)(lo.dec(i, D_C2W) - base);
  }
  return n;
}
int pack_weights(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, const float* P, float* ws, void* st) {
  int offs[256];
  const int n = pack_offsets(c, lo, offs);
  if (n == 0) return 0;
  return adt_pack_wimg(P + lo.posw(), ws + w.wpack, offs, n, st);
}

// slot of a 64 x 64 block in a workgroup's partial area: the order of pack_weights
enum { PS_E_IN = 0, PS_E_O = 3, PS_E_C1 = 4, PS_E_C2 = 5, PS_D_SIN = 6, PS_D_EIN = 9, PS_D_SO = 12, PS_D_EO = 13, PS_D_C1 = 14, PS_D_C2 = 15 };

// the partial slots of the encoder (enc = true) and / or decoder (dec = true) blocks of every layer and their offsets in G
int partial_slots(const adt_sasrec_cfg* c, const Layout& lo, bool enc, bool dec, int* slots, int* offs);
// workgroups that write partial slot `slot` of a step with B sequences: the token-chain kernels run S workgroups per sequence
// k_seqtt_attn_pre_bwd: one workgroup per sequence, two at small batches (ADT_ATTN_SPLIT=0: always one)
int attn_split(int B) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_ATTN_SPLIT"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on && seq_split(B) >= 2 ? 2 : 1;
}
// workgroups that wrote partial `slot`
int partial_nwg(int slot, int B) {
  const int k = slot % 16;
  const bool attn_block = (k >= PS_E_IN && k < PS_E_IN + 3) || (k >= PS_D_SIN && k < PS_D_SIN + 3);
  return attn_block ? B * attn_split(B) : B * seq_split(B);
}
// sums the partials of the encoder (enc = true) and / or decoder (dec = true) blocks of every layer into G
int reduce_partials(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, float* G, float* ws, bool enc, bool dec, void* st) {
  int slots[256], offs[256];
  const int n = partial_slots(c, lo, enc, dec, slots, offs);
  int nwg[256];
  for (int i = 0; i < n; ++i) nwg[i] = partial_nwg(slots[i], (int)w.B);
  return adt_dwpart_reduce_n(G, ws + w.part, (size_t)w.part_stride, (int)w.B, nwg, slots, offs, n, st);
}
int partial_slots(const adt_sasrec_cfg* c, const Layout& lo, bool enc, bool dec, int* slots, int* offs) {
  int n = 0;
  const int dd = c->hidden * c->hidden;
  for (int i = 0; i < c->num_layers; ++i) {
    if (enc) {
      for (int j = 0; j < 3; ++j) { slots[n] = 16 * i + PS_E_IN + j; offs[n++] = (int)(lo.enc(i, E_INW) + j * dd); }
      slots[n] = 16 * i + PS_E_O; offs[n++] = (int)lo.enc(i, E_OW);
      slots[n] = 16 * i + PS_E_C1; offs[n++] = (int)lo.enc(i, E_C1W);
      slots[n] = 16 * i + PS_E_C2; offs[n++] = (int)lo.enc(i, E_C2W);
    }
    if (dec) {
      for (int j = 0; j < 3; ++j) { slots[n] = 16 * i + PS_D_SIN + j; offs[n++] = (int)(lo.dec(i, D_SINW) + j * dd); }
      for (int j = 0; j < 3; ++j) { slots[n] = 16 * i + PS_D_EIN + j; offs[n++] = (int)(lo.dec(i, D_EINW) + j * dd); }
      slots[n] = 16 * i + PS_D_SO; offs[n++] = (int)lo.dec(i, D_SOW);
      slots[n] = 16 * i + PS_D_EO; offs[n++] = (int)lo.dec(i, D_EOW);
      slots[n] = 16 * i + PS_D_C1; offs[n++] = (int)lo.dec(i, D_C1W);
      slots[n] = 16 * i + PS_D_C2; offs[n++] = (int)lo.dec(i, D_C2W);
    }
  }
  return n;
}
// single-GPU step: the partials are summed by the optimizer's first kernel (adt_fold_parts_clip_adam), not by k_dwpart_reduce launches
bool fold_sums_partials(const adt_sasrec_cfg* c, const WS& w) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_FOLD_PARTS"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on && w.part_stride > 0 && c->num_layers <= 2 && adt_seq_partials(c->prec, c->maxlen, c->hidden, c->hidden / c->num_heads) != 0;
}

adt::SeqBwdArgs seq_bwd_args(int L, int B, int H, const int32_t* ids, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset, int hd) {
  adt::SeqBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.L = L; a.B = B; a.H = H; a.ids = ids; a.drop = adt_make_drop(p, seed, site); a.b_offset = b_offset; a.ln_eps = LN_EPS;
  a.scale = 1.0f / sqrtf((float)hd);
  return a;
}

adt::SeqFwdArgs seq_args(int L, int B, int H, const int32_t* ids, float p, const uint32_t* seed, uint32_t b_offset, int hd) {
  adt::SeqFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.nsplit = seq_split(B);
  a.L = L; a.B = B; a.H = H; a.ids = ids; a.drop = adt_make_drop(p, seed, 0); a.b_offset = b_offset; a.ln_eps = LN_EPS;
  a.scale = 1.0f / sqrtf((float)hd);
  return a;
}

// argument block of one encoder layer's fused forward (adt_seqfwd_tt.cuh / adt_seqfwd.cuh)
adt::SeqFwdArgs enc_layer_seq_args(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, const float* P, float* ws, const int32_t* seq, float p,
                                   const uint32_t* seed, uint32_t b_offset, int i, bool training_outputs, bool lean) {
  const int d = (int)w.d, H = (int)w.H, hd = d / H, L = (int)w.L, B = (int)w.B, prec = c->prec;
  const int64_t Td = up64(w.T * w.d);
  float* x = ws + w.enc_x + i * Td;
  float* y = ws + w.enc_x + (i + 1) * Td;
  float* base = ws + i * w.e_stride;
  adt::SeqFwdArgs a = seq_args(L, B, H, seq, p, seed, b_offset, hd);
  a.x = i == 0 ? nullptr : x; a.E = P + lo.item(); a.P = P + lo.posw(); a.emb_scale = sqrtf((float)d); a.site_emb = SITE_EMB_SEQ;
  a.site_attn = enc_site(i, 0); a.site1 = enc_site(i, 1); a.site2 = enc_site(i, 2);
  a.gamma = P + lo.enc(i, E_LN1W); a.beta = P + lo.enc(i, E_LN1B); a.Win = P + lo.enc(i, E_INW); a.bin = P + lo.enc(i, E_INB);
  a.Wo = P + lo.enc(i, E_OW); a.bo = P + lo.enc(i, E_OB); a.gamma2 = P + lo.enc(i, E_LN2W); a.beta2 = P + lo.enc(i, E_LN2B);
  a.W1 = P + lo.enc(i, E_C1W); a.b1 = P + lo.enc(i, E_C1B); a.W2 = P + lo.enc(i, E_C2W); a.b2 = P + lo.enc(i, E_C2B);
  a.x_out = x; a.xn = base + w.e_qn; a.qkv = base + w.e_qkv; a.o = base + w.e_o; a.lse = base + w.e_lse;
  a.mask = reinterpret_cast<uint32_t*>(base + w.e_mask); a.h = base + w.e_h; a.u = base + w.e_u; a.y = y;
  if (training_outputs && H > 1) { a.rec = base + w.e_rec; a.Ws = P + lo.enc(i, E_SW); a.bs = P + lo.enc(i, E_SB); }
  a.wp_base = P + lo.posw(); a.wp_img = prec == ADT_PREC_BF16 ? (const void*)(ws + w.wpack) : nullptr;
  if (lean) { a.xn = nullptr; a.qkv = nullptr; a.saved_bf16 = 1; }
  return a;
}

// encoder stack forward (gathers the input embedding inside the first chain) + last LayerNorm (+ logits and the
// cross-attention k/v projections of every decoder layer when training_outputs)
int encoder_forward(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, const float* P, float* ws,
                    const int32_t* seq, const int32_t* pos, const int32_t* neg, float p, const uint32_t* seed,
                    uint32_t b_offset, bool training_outputs, void* st, bool packed = false) {
  const int T = (int)w.T, d = (int)w.d, H = (int)w.H, hd = d / H, L = (int)w.L, B = (int)w.B, prec = c->prec, nl = c->num_layers;
  const int64_t Td = up64(w.T * w.d);
  const uint32_t ro = b_offset * (uint32_t)L;
  const int dd = d * d;
  const bool use_seq = adt_seq_supported(prec, L, d, hd) != 0;
  const bool lean = training_outputs && adt_seq_lean(prec, L, d, hd) != 0;   // bf16 saved tensors, no LN(x) / qkv (the fused backward recomputes them)
  if (!packed) CK(pack_weights(c, lo, w, P, ws, st));      // packed: adt_sasrec_step_begin* of this step wrote the images
  const float* wp_base = P + lo.posw();
  const void* wp_img = prec == ADT_PREC_BF16 ? (const void*)(ws + w.wpack) : nullptr;
  for (int i = 0; i < nl; ++i) {
    float* x = ws + w.enc_x + i * Td;
    float* y = ws + w.enc_x + (i + 1) * Td;
    float* base = ws + i * w.e_stride;
    float *qn = base + w.e_qn, *qkv = base + w.e_qkv, *o = base + w.e_o, *lse = base + w.e_lse, *h = base + w.e_h,
          *u = base + w.e_u, *rec = base + w.e_rec;
    const float* inw = P + lo.enc(i, E_INW);
    const float* inb = P + lo.enc(i, E_INB);
    if (use_seq) {   // the whole layer in one launch, one workgroup per sequence (adt_seqfwd.cuh)
      const adt::SeqFwdArgs a = enc_layer_seq_args(c, lo, w, P, ws, seq, p, seed, b_offset, i, training_outputs, lean);
      CK(adt_launch_seq_enc_fwd(hd, a, st));
      continue;
    }
    {  // [gather] ; Q = LN1(x) ; q = Q Wq^T + bq ; k, v = x Wk^T, x Wv^T     (sasrec/model.py:34-41, modules.py:646-647)
      adt::FwdChainArgs a = fwd_args(T, L, B, H, seq, p, seed, ro);
      a.x = i == 0 ? nullptr : x; a.E = P + lo.item(); a.P = P + lo.posw(); a.emb_scale = sqrtf((float)d); a.site0 = SITE_EMB_SEQ;
      a.gamma = P + lo.enc(i, E_LN1W); a.beta = P + lo.enc(i, E_LN1B);
      for (int j = 0; j < 3; ++j) { a.W[j] = inw + j * dd; a.b[j] = inb + j * d; }
      a.o0 = x; a.ld0 = d; a.o1 = qn; a.ld1 = d; a.o2 = qkv; a.ld2 = 3 * d;
      CK(adt_launch_fwdchain(prec, 0, a, st));
    }
    CK(adt_attn_fwd(prec, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, B, H, L, hd, 1, p, seed, enc_site(i, 0), b_offset, o, d,
                    lse, reinterpret_cast<uint32_t*>(base + w.e_mask), st));
    {  // h = Q + out_proj(o); h2 = LN2(h); u = relu(drop1(conv1 h2)); y = (h2 + drop2(conv2 u)) * mask   (:648-654)
      adt::FwdChainArgs a = fwd_args(T, L, B, H, seq, p, seed, ro);
      a.x = o; a.r0 = qn; a.site1 = enc_site(i, 1); a.site2 = enc_site(i, 2);
      a.W[0] = P + lo.enc(i, E_OW); a.b[0] = P + lo.enc(i, E_OB);
      a.W[1] = P + lo.enc(i, E_C1W); a.b[1] = P + lo.enc(i, E_C1B);
      a.W[2] = P + lo.enc(i, E_C2W); a.b[2] = P + lo.enc(i, E_C2B);
      a.gamma = P + lo.enc(i, E_LN2W); a.beta = P + lo.enc(i, E_LN2B);
      a.o0 = h; a.ld0 = d; a.o1 = u; a.ld1 = d; a.o2 = y; a.ld2 = d;
      int which = 2;
      if (training_outputs) {
        a.rec = rec; a.Ws = P + lo.enc(i, E_SW); a.bs = P + lo.enc(i, E_SB);
        which = H <= 2 ? 6 : (H <= 4 ? 7 : 8);     // classifier width specialisations
      }
      CK(adt_launch_fwdchain(prec, which, a, st));
    }
  }
  // log_feats = last_layernorm(encoder out); pos/neg logits; [k2, v2] of every decoder layer   (model.py:48, :72-76)
  for (int j0 = 0; j0 == 0 || (training_outputs && !use_seq && j0 < nl); j0 += 2) {
    adt::FwdChainArgs a = fwd_args(T, L, B, H, seq, 0.f, nullptr, ro);
    a.x = ws + w.enc_x + nl * Td; a.gamma = P + lo.lnl_w(); a.beta = P + lo.lnl_b();
    if (j0 == 0) { a.o0 = ws + w.f; a.ld0 = d; }
    if (training_outputs) {
      if (j0 == 0) { a.E = P + lo.item(); a.pos = pos; a.neg = neg; a.pos_logits = ws + w.posl; a.neg_logits = ws + w.negl; }
      a.nkv = use_seq ? 0 : (nl - j0 >= 2 ? 2 : 1);       // the fused decoder layer projects its own cross keys / values
      for (int k = 0; k < a.nkv; ++k) {
        const float* einw = P + lo.dec(j0 + k, D_EINW);
        const float* einb = P + lo.dec(j0 + k, D_EINB);
        a.W[2 * k] = einw + dd; a.W[2 * k + 1] = einw + 2 * dd; a.b[2 * k] = einb + d; a.b[2 * k + 1] = einb + 2 * d;
        float* kv2 = ws + (j0 + k) * w.d_stride + w.d_kv2;
        if (k == 0) { a.o2 = kv2; a.ld2 = 2 * d; } else { a.o3 = kv2; a.ld3 = 2 * d; }
      }
    }
    CK(adt_launch_fwdchain(prec, 5, a, st));
  }
  return 0;
}

// one decoder layer, fused forward: one launch, one workgroup per sequence (adt_seqfwd_tt.cuh / adt_seqfwd.cuh)
adt::SeqFwdArgs dec_layer_seq_args(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, const float* P, float* ws, const int32_t* dec, float p,
                                   const uint32_t* seed, uint32_t b_offset, int i) {
  const int d = (int)w.d, H = (int)w.H, hd = d / H, L = (int)w.L, B_ = (int)w.B, prec = c->prec;
  const int64_t Td = up64(w.T * w.d);
  float* x = ws + w.dec_x + i * Td;
  float* y = ws + w.dec_x + (i + 1) * Td;
  float* base = ws + i * w.d_stride;   // d_* offsets are absolute for layer 0
  float *dn = base + w.d_dn, *qkv = base + w.d_qkv, *o1 = base + w.d_o1, *lse1 = base + w.d_lse1, *a1 = base + w.d_a1,
        *q2 = base + w.d_q2, *kv2 = base + w.d_kv2, *o2 = base + w.d_o2, *lse2 = base + w.d_lse2, *a2 = base + w.d_a2, *u = base + w.d_u;
  adt::SeqFwdArgs a = seq_args(L, B_, H, dec, p, seed, b_offset, hd);
  a.x = i == 0 ? nullptr : x; a.E = P + lo.item(); a.P = P + lo.posw(); a.emb_scale = sqrtf((float)d); a.site_emb = SITE_EMB_DEC;
  a.site_attn = dec_site(i, 0); a.site_attn2 = dec_site(i, 1); a.site1 = dec_site(i, 2); a.site2 = dec_site(i, 3);
  a.gamma = P + lo.dec(i, D_LNW); a.beta = P + lo.dec(i, D_LNB); a.Win = P + lo.dec(i, D_SINW); a.bin = P + lo.dec(i, D_SINB);
  a.Wo = P + lo.dec(i, D_SOW); a.bo = P + lo.dec(i, D_SOB); a.f = ws + w.f; a.Win2 = P + lo.dec(i, D_EINW); a.bin2 = P + lo.dec(i, D_EINB);
  a.Wo2 = P + lo.dec(i, D_EOW); a.bo2 = P + lo.dec(i, D_EOB);
  a.W1 = P + lo.dec(i, D_C1W); a.b1 = P + lo.dec(i, D_C1B); a.W2 = P + lo.dec(i, D_C2W); a.b2 = P + lo.dec(i, D_C2B);
  a.x_out = x; a.xn = dn; a.qkv = qkv; a.o = o1; a.lse = lse1; a.mask = reinterpret_cast<uint32_t*>(base + w.d_mask1);
  a.a1 = a1; a.q2 = q2; a.kv2 = kv2; a.o2 = o2; a.lse2 = lse2; a.mask2 = reinterpret_cast<uint32_t*>(base + w.d_mask2);
  a.h = a2; a.u = u; a.y = y;
  a.wp_base = P + lo.posw(); a.wp_img = ws + w.wpack;
  if (adt_seq_lean(prec, L, d, hd)) { a.xn = nullptr; a.qkv = nullptr; a.saved_bf16 = 1; }
  return a;
}
int dec_layer_seq_fwd(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, const float* P, float* ws, const int32_t* dec, float p,
                      const uint32_t* seed, uint32_t b_offset, int i, void* st) {
  const adt::SeqFwdArgs a = dec_layer_seq_args(c, lo, w, P, ws, dec, p, seed, b_offset, i);
  return adt_launch_seq_dec_fwd((int)(w.d / w.H), a, st);
}

// Training forward + loss assembly of the lean per-sequence path: log_feats is the tail of the last encoder layer's kernel (no k_final_fwd launch)
// and the pos / neg logits + BCE seed are formed by the backward's first side kernel, which gathers E[pos], E[neg] and log_feats anyway
// (adt_logits_bce_scatter; adt_sasrec_backward with phase bit 4).  Returns 1 when the shape is not covered (nothing launched).
bool bce_deferred(const adt_sasrec_cfg* c) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_FWD_FUSED"); on = (e && atoi(e) == 0) ? 0 : 1; }
  const int d = c->hidden, hd = d / c->num_heads;
  return on && d == 64 && c->num_layers <= 4 && adt_seq_lean(c->prec, c->maxlen, d, hd) != 0;
}
// ADT_EMBED3=0: the two embedding gradients and the positive-logit rows keep their own scatters (default: one pass, adt_embed_bwd3 -- deferred
// path only, without the sorted form)
bool embed3_on(const WS& w) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_EMBED3"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on != 0 && !item_det(w) && w.d == 64;
}
// ADT_BCE_MERGED=0: the forward's logits / BCE kernel (training bit 2) goes to the side stream instead of into the loss launch
bool bce_merged() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_BCE_MERGED"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on != 0;
}
struct RingRef { const int32_t* ring; int64_t slot_ints; int nslots; uint32_t* state; uint32_t* consumed; int32_t* staging; };
// bce_side (training bit 2 of adt_sasrec_forward_loss*): the logits + BCE + item-row scatter kernel of the deferred path is launched HERE, on
// the library's side stream beside the streaming loss pass (both are memory passes; under the backward's first chain kernels it stretched the
// latency-bound cross-attention backward by about its own duration), and joined by adt_sasrec_backward(phase bit 5) in front of the first
// kernel that reads d log_feats.  The item-table replicas are zero: adt_sasrec_step_begin* of this step.
int forward_loss_lean(const adt_sasrec_cfg* c, const Layout& lo, const WS& w, const float* P, float* ws, const int32_t* seq, const int32_t* dec,
                      const int32_t* pos, const int32_t* neg, bool bce_side, float p, const uint32_t* seed, uint32_t b_offset, const float* lambdas1,
                      const float* lambdas2, const RingRef& rr, void* st) {
  const int d = (int)w.d, H = (int)w.H, hd = d / H, nl = c->num_layers, T = (int)w.T;
  if (!bce_deferred(c)) return 1;
  const int64_t Td = up64(w.T * w.d), rec = up64(w.T * w.H * w.H);
  for (int i = 0; i < nl; ++i) {
    adt::SeqFwdArgs a = enc_layer_seq_args(c, lo, w, P, ws, seq, p, seed, b_offset, i, true, true);
    if (i == nl - 1) { a.lnl_gamma = P + lo.lnl_w(); a.lnl_beta = P + lo.lnl_b(); a.f_out = ws + w.f; }
    CK(adt_launch_seq_enc_fwd(hd, a, st));
  }
  for (int j = 0; j < nl; ++j) CK(dec_layer_seq_fwd(c, lo, w, P, ws, dec, p, seed, b_offset, j, st));
  // reconstruction + independence seeds in one launch (the BCE block is the backward's: no logits yet)
  float* loss = ws + w.loss;
  const float *A[4], *Bm[4], *rc[4];
  float *GA[4], *GB[4], *lm[4], *dr[4], *ln[4];
  // Reconstruction seeds: only the one the backward's FIRST kernels read as a plain input (d / d decoder output) is written here.  The others
  // were written to be added to the input gradient of a block that re-reads that very input: k_seqtt_attn_pre_bwd forms them from its own x
  // rows and the other stack's rows (SeqBwdArgs::seed_other) -- 39 MB of stores here and as many loads there less per step.
  for (int i = 0; i < nl; ++i) {
    A[i] = ws + w.enc_x + i * Td; Bm[i] = ws + w.dec_x + (nl - i) * Td;
    GA[i] = nullptr; GB[i] = i == 0 ? ws + w.g_dec_x + (nl - i) * Td : nullptr;      // (i >= 1: not even read here -- the encoder block's backward adds the loss term)
    lm[i] = loss + 64 * (2 + i);
    rc[i] = ws + i * w.e_stride + w.e_rec; dr[i] = ws + w.g_rec + i * rec; ln[i] = loss + 64 * (2 + nl + i);
  }
  // host-fed step: the next batch's ring slot is copied to the staging buffer in two halves, one inside this launch, one inside the embedding
  // scatter at the end of the backward (which then marks it staged) -- 819 KB over PCIe do not fit behind either launch alone
  if (rr.ring && rr.staging && rr.state && bce_side && embed3_on(w)) adt_loss_seeds_split_prefetch();
  const bool merged = bce_side && bce_merged();
  if (merged)      // ... as the first workgroups of the loss launch itself: no second stream, no fork / join
    adt_loss_seeds_attach_logits(ws + w.f, P + lo.item(), pos, neg, ws + w.norms, T, ws + w.posl, ws + w.negl, ws + w.g_pos, ws + w.g_neg, ws + w.loss,
                                 ws + w.g_f, item_det(w) ? nullptr : ws + w.rep, NREP, w.rep_stride, embed3_on(w) ? 1 : 0);
  SideStream* const sd = (bce_side && !merged && (side_sites() & 1)) ? side_stream((hipStream_t)st) : nullptr;
  if (bce_side && !merged) CK(side_mark(sd, 0, st));
  // (+ the next step's id batch, if its producer has published it: the PCIe read runs under this streaming pass)
  CK(adt_loss_seeds_prefetch(nullptr, nullptr, nullptr, 0, ws + w.norms, nullptr, nullptr, loss, nl, A, Bm, w.T * w.d, lambdas1, GA, 0, GB, lm,
                             H > 1 ? nl : 0, rc, T, H, lambdas2[nl - 1], dr, ln, rr.ring, rr.slot_ints, rr.nslots, 4 * (int64_t)w.T + 4, rr.state,
                             rr.consumed, rr.staging, st));
  if (bce_side && !merged) {
    void* s2 = nullptr;
    CK(side_enter(sd, 0, st, &s2));
    CK(adt_logits_bce_scatter_ex(ws + w.f, P + lo.item(), pos, neg, ws + w.norms, T, ws + w.posl, ws + w.negl, ws + w.g_pos, ws + w.g_neg, ws + w.loss,
                                 ws + w.g_f, item_det(w) ? nullptr : ws + w.rep, NREP, w.rep_stride, embed3_on(w) ? 1 : 0, s2));
    if (sd && hipEventRecord(sd->join_ev[0], sd->s) != hipSuccess) return adt_set_error("forward_loss: logits event");
  }
  return 0;
}

}  // namespace

extern "C" {

int64_t adt_sasrec_param_layout(const adt_sasrec_cfg* cfg, int64_t* offsets) {
  Layout lo;
  if (!make_layout(cfg, &lo)) return adt_set_error("param_layout: bad cfg");
  if (offsets)
    for (int i = 0; i < 4 + 30 * cfg->num_layers; ++i) offsets[i] = lo.off[i];
  return lo.total;
}

int64_t adt_sasrec_workspace_floats(const adt_sasrec_cfg* cfg, int B) {
  WS w;
  make_ws(cfg, B, &w);
  return w.total;
}

int64_t adt_sasrec_ws_offset(const adt_sasrec_cfg* cfg, int B, int what, int layer) {
  WS w;
  make_ws(cfg, B, &w);
  const int64_t Td = up64(w.T * w.d), rec = up64(w.T * w.H * w.H);
  switch (what) {
    case ADT_WS_ENC_X: return w.enc_x + layer * Td;
    case ADT_WS_DEC_X: return w.dec_x + layer * Td;
    case ADT_WS_REC: return layer * w.e_stride + w.e_rec;
    case ADT_WS_POS_LOGITS: return w.posl;
    case ADT_WS_NEG_LOGITS: return w.negl;
    case ADT_WS_F: return w.f;
    case ADT_WS_G_ENC_X: return w.g_enc_x + layer * Td;
    case ADT_WS_G_DEC_X: return w.g_dec_x + layer * Td;
    case ADT_WS_G_REC: return w.g_rec + layer * rec;
    case ADT_WS_G_POS: return w.g_pos;
    case ADT_WS_G_NEG: return w.g_neg;
    case ADT_WS_LOSS: return w.loss;
    case ADT_WS_NORMS: return w.norms;
    case ADT_WS_SCAL: return w.scal;
  }
  return adt_set_error("ws_offset: unknown id %d", what);
}

int adt_sasrec_forward(const adt_sasrec_cfg* c, const float* P, float* ws, const int32_t* seq, const int32_t* dec,
                       const int32_t* pos, const int32_t* neg, int B, int training, const uint32_t* seed,
                       uint32_t b_offset, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  const int T = (int)w.T, d = (int)w.d, H = (int)w.H, hd = d / H, L = (int)w.L, prec = c->prec;
  const int64_t Td = up64(w.T * w.d);
  const bool packed = (training & 2) != 0;       // bit 1: the weight images were packed by adt_sasrec_step_begin* of this step
  training &= 1;
  const float p = training ? c->dropout : 0.f;
  const uint32_t ro = b_offset * (uint32_t)L;
  CK(encoder_forward(c, lo, w, P, ws, seq, pos, neg, p, seed, b_offset, true, st, packed));
  const float* f = ws + w.f;
  const int dd = d * d;
  const int B_ = (int)w.B;
  for (int i = 0; i < c->num_layers; ++i) {
    float* x = ws + w.dec_x + i * Td;
    float* y = ws + w.dec_x + (i + 1) * Td;
    float* base = ws + i * w.d_stride;   // d_* offsets are absolute for layer 0
    float *dn = base + w.d_dn, *qkv = base + w.d_qkv, *o1 = base + w.d_o1, *lse1 = base + w.d_lse1, *a1 = base + w.d_a1,
          *q2 = base + w.d_q2, *kv2 = base + w.d_kv2, *o2 = base + w.d_o2, *lse2 = base + w.d_lse2, *a2 = base + w.d_a2,
          *u = base + w.d_u;
    const float* sinw = P + lo.dec(i, D_SINW);
    const float* sinb = P + lo.dec(i, D_SINB);
    const float* einw = P + lo.dec(i, D_EINW);
    const float* einb = P + lo.dec(i, D_EINB);
    if (adt_seq_supported(prec, L, d, hd)) {   // the whole layer in one launch, one workgroup per sequence (adt_seqfwd.cuh)
      CK(dec_layer_seq_fwd(c, lo, w, P, ws, dec, p, seed, b_offset, i, st));
      continue;
    }
    {  // [gather] ; D = LN(x) ; qkv = D Win^T + b                          (sasrec/model.py:53-59, modules.py:668-670)
      adt::FwdChainArgs a = fwd_args(T, L, B_, H, dec, p, seed, ro);
      a.x = i == 0 ? nullptr : x; a.E = P + lo.item(); a.P = P + lo.posw(); a.emb_scale = sqrtf((float)d); a.site0 = SITE_EMB_DEC;
      a.gamma = P + lo.dec(i, D_LNW); a.beta = P + lo.dec(i, D_LNB);
      for (int j = 0; j < 3; ++j) { a.W[j] = sinw + j * dd; a.b[j] = sinb + j * d; }
      a.o0 = x; a.ld0 = d; a.o1 = dn; a.ld1 = d; a.o2 = qkv; a.ld2 = 3 * d;
      CK(adt_launch_fwdchain(prec, 1, a, st));
    }
    CK(adt_attn_fwd(prec, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, B_, H, L, hd, 1, p, seed, dec_site(i, 0), b_offset, o1, d,
                    lse1, reinterpret_cast<uint32_t*>(base + w.d_mask1), st));
    {  // a1 = out_proj(o1); q2 = a1 Wq^T + bq
      adt::FwdChainArgs a = fwd_args(T, L, B_, H, dec, 0.f, nullptr, ro);
      a.x = o1; a.W[0] = P + lo.dec(i, D_SOW); a.b[0] = P + lo.dec(i, D_SOB); a.W[1] = einw; a.b[1] = einb;
      a.o0 = a1; a.ld0 = d; a.o2 = q2; a.ld2 = d;
      CK(adt_launch_fwdchain(prec, 3, a, st));
    }
    CK(adt_attn_fwd(prec, q2, d, kv2, 2 * d, kv2 + d, 2 * d, B_, H, L, hd, 1, p, seed, dec_site(i, 1), b_offset, o2, d, lse2,
                    reinterpret_cast<uint32_t*>(base + w.d_mask2), st));
    {  // a2 = out_proj(o2); y = (D + a2 + drop2(conv2 relu(drop1(conv1 a2)))) * mask       (:673-676, :629-633)
      adt::FwdChainArgs a = fwd_args(T, L, B_, H, dec, p, seed, ro);
      a.x = o2; a.r0 = dn; a.site1 = dec_site(i, 2); a.site2 = dec_site(i, 3);
      a.W[0] = P + lo.dec(i, D_EOW); a.b[0] = P + lo.dec(i, D_EOB);
      a.W[1] = P + lo.dec(i, D_C1W); a.b[1] = P + lo.dec(i, D_C1B);
      a.W[2] = P + lo.dec(i, D_C2W); a.b[2] = P + lo.dec(i, D_C2B);
      a.o0 = a2; a.ld0 = d; a.o1 = u; a.ld1 = d; a.o2 = y; a.ld2 = d;
      CK(adt_launch_fwdchain(prec, 4, a, st));
    }
  }
  return 0;
}

// adt_sasrec_forward (training) + adt_sasrec_loss_seed_nz in one call; on the lean per-sequence path (adt_sasrec_bce_deferred) without the
// k_final_fwd launch and without the BCE block of the loss assembly: the caller must then run adt_sasrec_backward with phase bit 4 (+ 16).
// The loss slots must have been zeroed (adt_sasrec_step_begin*).
int adt_sasrec_bce_deferred(const adt_sasrec_cfg* c) { return check_cfg(c) == 0 && bce_deferred(c) ? 1 : 0; }
int adt_sasrec_forward_loss(const adt_sasrec_cfg* c, const float* P, float* ws, const int32_t* seq, const int32_t* dec, const int32_t* pos,
                            const int32_t* neg, int B, int training, const uint32_t* seed, uint32_t b_offset, const float* lambdas1,
                            const float* lambdas2, void* st) {
  return adt_sasrec_forward_loss_prefetch(c, P, ws, seq, dec, pos, neg, B, training, seed, b_offset, lambdas1, lambdas2, nullptr, 0, 0, nullptr, nullptr,
                                          nullptr, st);
}

int adt_sasrec_forward_loss_prefetch(const adt_sasrec_cfg* c, const float* P, float* ws, const int32_t* seq, const int32_t* dec, const int32_t* pos,
                                     const int32_t* neg, int B, int training, const uint32_t* seed, uint32_t b_offset, const float* lambdas1,
                                     const float* lambdas2, const int32_t* ring, int64_t slot_ints, int nslots, uint32_t* state, uint32_t* consumed,
                                     int32_t* staging, void* st) {
  CK(check_cfg(c));
  if ((training & 1) && (training & 2)) {      // training forward on weight images packed by this step's adt_sasrec_step_begin*
    Layout lo;
    make_layout(c, &lo);
    WS w;
    make_ws(c, B, &w);
    const RingRef rr{ring, slot_ints, nslots, state, consumed, staging};
    const int rc = forward_loss_lean(c, lo, w, P, ws, seq, dec, pos, neg, (training & 4) != 0, c->dropout, seed, b_offset, lambdas1, lambdas2, rr, st);
    if (rc <= 0) return rc;
  }
  CK(adt_sasrec_forward(c, P, ws, seq, dec, pos, neg, B, training, seed, b_offset, st));
  return adt_sasrec_loss_seed_nz(c, ws, pos, B, lambdas1, lambdas2, st);
}

// Measurement hook (bench.py roofline): launches ONLY the fused forward of decoder layer `layer` on the workspace of a completed
// adt_sasrec_forward of the same batch (its inputs -- log_feats, the layer input, the packed weight images -- are read from there).
int adt_sasrec_probe_dec_layer_fwd(const adt_sasrec_cfg* c, const float* P, float* ws, const int32_t* dec, int B, int training,
                                   const uint32_t* seed, uint32_t b_offset, int layer, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  const int d = (int)w.d, hd = d / (int)w.H;
  if (layer < 0 || layer >= c->num_layers) return adt_set_error("probe: layer %d", layer);
  if (!adt_seq_supported(c->prec, (int)w.L, d, hd)) return adt_set_error("probe: the fused decoder layer does not cover this configuration");
  return dec_layer_seq_fwd(c, lo, w, P, ws, dec, training ? c->dropout : 0.f, seed, b_offset, layer, st);
}

int adt_sasrec_step_begin(const adt_sasrec_cfg* c, float* ws, int B, uint32_t* seed, uint32_t seed_inc, const float* norms_src, const float* P,
                          float* G, int64_t n, float* scal, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  int offs[256];
  const int npack = pack_offsets(c, lo, offs);
  return adt_step_begin_launch(seed, seed_inc, ws + w.norms, norms_src, ws + w.loss, 64 * (2 + 2 * c->num_layers), scal, G, n, P + lo.item(),
                               (int64_t)(c->item_num + 1) * c->hidden, ws + w.rep, (int64_t)NREP * w.rep_stride + (int64_t)NREPP * w.prep_stride,
                               P + lo.posw(), ws + w.wpack, offs, npack, st);      // item-table replicas + parameter replicas (adjacent): one fill
}

int adt_sasrec_step_begin_ring(const adt_sasrec_cfg* c, float* ws, int B, uint32_t* seed, uint32_t seed_inc, const int32_t* ring, int64_t slot_ints,
                               int nslots, int32_t* ids_dst, uint32_t* state, uint32_t* consumed, const float* P, float* G, int64_t n, float* scal,
                               void* st) {
  return adt_sasrec_step_begin_ring_staged(c, ws, B, seed, seed_inc, ring, slot_ints, nslots, ids_dst, state, consumed, nullptr, nullptr, P, G, n, scal, st);
}

int adt_sasrec_step_begin_ring_staged(const adt_sasrec_cfg* c, float* ws, int B, uint32_t* seed, uint32_t seed_inc, const int32_t* ring,
                                      int64_t slot_ints, int nslots, int32_t* ids_dst, uint32_t* state, uint32_t* consumed, const int32_t* staging,
                                      const uint32_t* produced, const float* P, float* G, int64_t n, float* scal, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  int offs[256];
  const int npack = pack_offsets(c, lo, offs);
  return adt_step_begin_ring_launch(seed, seed_inc, ws + w.norms, ws + w.loss, 64 * (2 + 2 * c->num_layers), scal, G, n, P + lo.item(),
                                    (int64_t)(c->item_num + 1) * c->hidden, ring, slot_ints, nslots, ids_dst, 4 * (int64_t)w.T + 4, state, consumed,
                                    staging, produced, ws + w.rep, (int64_t)NREP * w.rep_stride + (int64_t)NREPP * w.prep_stride, P + lo.posw(),
                                    ws + w.wpack, offs, npack, st);
}

static int loss_seed_impl(const adt_sasrec_cfg* c, float* ws, const int32_t* pos, int B, const float* lambdas1, const float* lambdas2,
                          bool zero_loss, void* st);
int adt_sasrec_loss_seed(const adt_sasrec_cfg* c, float* ws, const int32_t* pos, int B, const float* lambdas1,
                         const float* lambdas2, void* st) {
  return loss_seed_impl(c, ws, pos, B, lambdas1, lambdas2, true, st);
}
int adt_sasrec_loss_seed_nz(const adt_sasrec_cfg* c, float* ws, const int32_t* pos, int B, const float* lambdas1,
                            const float* lambdas2, void* st) {
  return loss_seed_impl(c, ws, pos, B, lambdas1, lambdas2, false, st);
}
static int loss_seed_impl(const adt_sasrec_cfg* c, float* ws, const int32_t* pos, int B, const float* lambdas1, const float* lambdas2,
                          bool zero_loss, void* st) {
  WS w;
  make_ws(c, B, &w);
  const int nl = c->num_layers, T = (int)w.T, H = (int)w.H;
  const int64_t Td = up64(w.T * w.d), rec = up64(w.T * w.H * w.H);
  float* loss = ws + w.loss;
  const float* norms = ws + w.norms;
  if (zero_loss && adt::zero_f32_async(loss, (size_t)64 * (2 + 2 * nl), (hipStream_t)st)) return adt_set_error("loss zero");
  if (nl <= 4) {   // one launch for all of them (adt_misc.cuh: k_loss_seeds)
    const float *A[4], *Bm[4], *rc[4];
    float *GA[4], *GB[4], *lm[4], *dr[4], *ln[4];
    for (int i = 0; i < nl; ++i) {
      A[i] = ws + w.enc_x + i * Td; Bm[i] = ws + w.dec_x + (nl - i) * Td; GA[i] = ws + w.g_enc_x + i * Td; GB[i] = ws + w.g_dec_x + (nl - i) * Td;
      lm[i] = loss + 64 * (2 + i);
      rc[i] = ws + i * w.e_stride + w.e_rec; dr[i] = ws + w.g_rec + i * rec; ln[i] = loss + 64 * (2 + nl + i);
    }
    return adt_loss_seeds(ws + w.posl, ws + w.negl, pos, T, norms, ws + w.g_pos, ws + w.g_neg, loss, nl, A, Bm, w.T * w.d, lambdas1, GA, 0, GB, lm,
                          H > 1 ? nl : 0, rc, T, H, lambdas2[nl - 1], dr, ln, st);
  }
  CK(adt_bce_seed(ws + w.posl, ws + w.negl, pos, T, norms, ws + w.g_pos, ws + w.g_neg, loss, st));
  for (int i = 0; i < nl; ++i)   // enc_in[i] pairs with dec_out_rev[i] = DEC_X[nl - i]      (sasrec/main.py:155-158)
    CK(adt_mse_seed(ws + w.enc_x + i * Td, ws + w.dec_x + (nl - i) * Td, w.T * w.d, lambdas1[i], norms, ws + w.g_enc_x + i * Td, 0,
                    ws + w.g_dec_x + (nl - i) * Td, loss + 64 * (2 + i), st));
  if (H > 1)
    for (int l = 0; l < nl; ++l)   // stale loop index: lambdas2[nl-1] for every layer          (sasrec/main.py:169)
      CK(adt_nll_seed(ws + l * w.e_stride + w.e_rec, T, H, lambdas2[nl - 1], norms, ws + w.g_rec + l * rec, loss + 64 * (2 + nl + l), st));
  return 0;
}

// Measurement hook (bench.py's roofline probe; not part of include/adt_hip.h): HIP events recorded on the launch stream right before and right
// after ONE launch inside adt_sasrec_backward -- which = 1: the fused attention-block backward (k_seqtt_attn_pre_bwd) of encoder layer `layer`;
// which = 2: the same kernel's decoder instantiation for decoder layer `layer`; 3: see time_mark; 0 switches the hook off.
static int g_time_which = 0, g_time_layer = 0;
static hipEvent_t g_time_ev0 = nullptr, g_time_ev1 = nullptr;
extern "C" int adt_debug_time_launch(int which, int layer, void* ev_start, void* ev_stop) {
  g_time_which = which; g_time_layer = layer; g_time_ev0 = (hipEvent_t)ev_start; g_time_ev1 = (hipEvent_t)ev_stop;
  return 0;
}
// which = 3: calibration -- BOTH events in front of the launch of which = 1 (nothing between them): what an event pair itself adds to an interval
static inline void time_mark(int which, int layer, bool start, void* st) {
  if (!g_time_ev0 || !g_time_ev1 || g_time_layer != layer) return;
  if (g_time_which == 3 && which == 1 && start) {
    (void)hipEventRecord(g_time_ev0, (hipStream_t)st);
    (void)hipEventRecord(g_time_ev1, (hipStream_t)st);
  } else if (g_time_which == which) {
    (void)hipEventRecord(start ? g_time_ev0 : g_time_ev1, (hipStream_t)st);
  }
}

int adt_sasrec_backward(const adt_sasrec_cfg* c, const float* P, float* G, float* ws, const int32_t* seq,
                        const int32_t* dec, const int32_t* pos, const int32_t* neg, int B, int training,
                        const uint32_t* seed, uint32_t b_offset, int phase, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  const int T = (int)w.T, d = (int)w.d, H = (int)w.H, hd = d / H, L = (int)w.L, prec = c->prec, nl = c->num_layers;
  const int64_t Td = up64(w.T * w.d), recsz = up64(w.T * w.H * w.H);
  const float p = training ? c->dropout : 0.f;
  const uint32_t ro = b_offset * (uint32_t)L;
  float *s1 = ws + w.s1, *s3 = ws + w.s3, *s4 = ws + w.s4, *s5 = ws + w.s5;
  float* gf = ws + w.g_f;
  // the backward chains flush their weight / bias / LayerNorm gradients into NREPP zeroed replicas of the non-item parameters
  // (Gq + lo.xxx() addresses replica 0); each phase folds its range into G at its end
  float* const Gq = ws + w.prep - lo.posw();
  auto BA = [&](const int32_t* ids, float pp, const uint32_t* sd) {
    adt::BwdChainArgs a = bwd_args(T, L, (int)w.B, ids, pp, sd, ro);
    a.nrep = NREPP; a.rep_stride = (size_t)w.prep_stride;
    a.wp_base = P + lo.posw(); a.wp_img = prec == ADT_PREC_BF16 ? (const void*)(ws + w.wpack) : nullptr;   // packed by the forward of this step
    a.nsplit = seq_split((int)w.B);
    return a;
  };
  const int64_t dec_begin = lo.dec(0, 0);
  const float* f = ws + w.f;
  const bool use_seq = adt_seq_supported(prec, L, d, hd) != 0;
  const int lean = adt_seq_lean(prec, L, d, hd);      // what the forward of this step saved (same predicate, same process)
  // weight gradients through private per-sequence partials + one ordered sum instead of float atomics (adt_seqbwd_tt.cuh: sb_dw_tiles)
  const bool parts = w.part_stride > 0 && adt_seq_partials(prec, L, d, hd) != 0;
  auto PART = [&](int layer, int slot) { return parts ? ws + w.part + (int64_t)(16 * layer + slot) * 4096 : nullptr; };
  const char* const no_fallback = "backward: shape L=%d hd=%d left the per-sequence kernels although the partial-gradient path was chosen";
  const bool prep_zeroed = (phase & 4) != 0;      // bit 2: adt_sasrec_step_begin* of this step zeroed the item-table and parameter-gradient replicas
  const bool defer_fold = (phase & 8) != 0 && (phase & 3) == 0;      // bit 3 (one-phase only): adt_sasrec_fold_clip_adam does the last fold
  const bool late_parts = defer_fold && fold_sums_partials(c, w);      // adt_sasrec_fold_clip_adam sums the weight-gradient partials too
  // ... and the bias / LayerNorm / classifier gradient sums, which the chain kernels then STORE per workgroup (no float atomics, no replicas)
  auto VPART = [&](int layer, int k) { return late_parts ? ws + w.vpart + (int64_t)(5 * layer + k) * w.vcall : nullptr; };
  const bool bce_here = (phase & 16) != 0;        // bit 4: the forward was adt_sasrec_forward_loss on the deferred path: logits + BCE seed are formed here
  const bool bce_fwd = (phase & 32) != 0;         // bit 5: ... and launched that kernel itself (training bit 2): only its join is left
  if ((bce_here || bce_fwd) && !bce_deferred(c)) return adt_set_error("backward: phase bit 4 / 5 without the deferred-BCE forward (adt_sasrec_bce_deferred)");
  if (bce_fwd && (bce_here || (phase & 4) == 0 || (phase & 3) == 2)) return adt_set_error("backward: phase bit 5 goes with bit 2, without bit 4, in phase 0 or 1");
  // the forward of the deferred path (forward_loss_lean) does not materialise the reconstruction seeds that k_seqtt_attn_pre_bwd can form itself
  const bool seeds_virtual = bce_here || bce_fwd || (phase & 64) != 0;      // (bit 6: phase 2 of a two-phase backward behind such a forward)
  // ... and on that path the encoder / decoder embedding gradients and the positive-logit rows share ONE scatter at the end (adt_embed_bwd3)
  const bool embed3 = seeds_virtual && embed3_on(w);
  phase &= 3;
  const bool det = item_det(w);                   // item / positional table gradients by sorted segmented sums (no replicas, no float atomics)
  auto logits_scatter = [&](void* s) {      // d log_feats + item rows of pos / neg (+ logits and BCE seed on the deferred path)   (sasrec/model.py:72-76)
    if (bce_here)
      return adt_logits_bce_scatter_ex(f, P + lo.item(), pos, neg, ws + w.norms, T, ws + w.posl, ws + w.negl, ws + w.g_pos, ws + w.g_neg, ws + w.loss,
                                       gf, det ? nullptr : ws + w.rep, NREP, w.rep_stride, embed3 ? 1 : 0, s);
    if (det) return adt_logits_bwd_df(P + lo.item(), pos, neg, ws + w.g_pos, ws + w.g_neg, T, d, gf, d, s);
    return adt_logits_bwd_scatter(f, d, P + lo.item(), pos, neg, ws + w.g_pos, ws + w.g_neg, T, d, gf, d, ws + w.rep, NREP, w.rep_stride, s);
  };
  int32_t* const iwork = reinterpret_cast<int32_t*>(ws + w.isort);
  SideStream* const sd_sort = det ? side_stream((hipStream_t)st) : nullptr;      // the id sort runs beside the decoder's chain kernels, also in the two-phase form
  // (one-phase backward by default: the two-phase form belongs to the data-parallel step, whose capture already carries the collectives'
  // stream; there the side stream measured 0.692 against 0.699 ms on a 1-rank RCCL group and is opt-in: ADT_SIDE_STREAM_DP=1)
  static int dp_on = -1;
  if (dp_on < 0) { const char* e = getenv("ADT_SIDE_STREAM_DP"); dp_on = (e && atoi(e) != 0) ? 1 : 0; }
  SideStream* const sd = (phase == 0 || dp_on) ? side_stream((hipStream_t)st) : nullptr;
  int dec_side = 0;      // the decoder's embedding gradient + partial sums on the side stream: 1 marked, 2 enqueued
  bool dec_parts_done = false;
  if (phase == 0 || phase == 1) {
    // d log_feats (overwrites g_f) and item-table rows of pos/neg      (sasrec/model.py:72-76)
    // item-table replicas and parameter replicas are adjacent in the workspace: one fill
    if (w.prep != w.rep + NREP * w.rep_stride) return adt_set_error("workspace layout: replica areas not adjacent");
    // The item-table rows of the logits (and d log_feats, first read by the reverse of the cross-attention projections) go to the side
    // stream: they run under the first two decoder kernels, which only need the parameter replicas zeroed.
    int logits_side = 0;      // 1: marked, to be enqueued behind the first chain kernel ; 2: enqueued, to be joined ; 3: enqueued by the forward
    if (bce_fwd) logits_side = 3;
    if (det) {
      // sort of the step's ids + gather plan (a function of the ids and of fixed workspace addresses only); joined in front of the sums
      const int32_t* const ids4[4] = {seq, dec, pos, neg};
      const float* const rows4[4] = {ws + w.g_enc_x, ws + w.g_dec_x, f, f};
      const float* const coef4[4] = {nullptr, nullptr, ws + w.g_pos, ws + w.g_neg};
      const int kind4[4] = {0, 0, 1, 1};
      void* s2 = st;
      if (sd_sort) { CK(side_mark(sd_sort, 3, st)); CK(side_enter(sd_sort, 3, st, &s2)); }
      CK(adt_item_sort(ids4, 4, T, c->item_num + 1, rows4, coef4, kind4, ro, iwork, s2));
      if (sd_sort && hipEventRecord(sd_sort->join_ev[3], sd_sort->s) != hipSuccess) return adt_set_error("backward: sort event");
    }
    if (bce_fwd) {
      // nothing to launch: the forward put the kernel on the side stream (or, without one, on this stream); joined below
    } else if (det) {
      if (!prep_zeroed && adt::zero_f32_async(ws + w.prep, (size_t)NREPP * w.prep_stride, (hipStream_t)st)) return adt_set_error("replica zero");
      CK(logits_scatter(st));      // no atomics left in it: a short streaming pass on the caller's stream
    } else if (sd && (side_sites() & 1)) {
      CK(side_mark(sd, 0, st));
      if (!prep_zeroed && adt::zero_f32_async(ws + w.prep, (size_t)NREPP * w.prep_stride, (hipStream_t)st)) return adt_set_error("replica zero");
      logits_side = 1;
    } else {
      if (!prep_zeroed && adt::zero_f32_async(ws + w.rep, (size_t)NREP * w.rep_stride + (size_t)NREPP * w.prep_stride, (hipStream_t)st)) return adt_set_error("replica zero");
      CK(logits_scatter(st));
    }
    for (int i = nl - 1; i >= 0; --i) {
      float* gy = ws + w.g_dec_x + (i + 1) * Td;      // d loss / d (output of decoder layer i), complete
      float* gx = ws + w.g_dec_x + i * Td;            // accumulates d / d (input of layer i)
      const float* x = ws + w.dec_x + i * Td;
      float* base = ws + i * w.d_stride;
      float *qkv = base + w.d_qkv, *o1 = base + w.d_o1, *lse1 = base + w.d_lse1, *a1 = base + w.d_a1,
            *q2 = base + w.d_q2, *kv2 = base + w.d_kv2, *o2 = base + w.d_o2, *lse2 = base + w.d_lse2, *a2 = base + w.d_a2,
            *u = base + w.d_u;
      const float* einw = P + lo.dec(i, D_EINW);
      float* geinw = Gq + lo.dec(i, D_EINW);
      float* geinb = Gq + lo.dec(i, D_EINB);
      const float* sinw = P + lo.dec(i, D_SINW);
      float* gsinw = Gq + lo.dec(i, D_SINW);
      float* gsinb = Gq + lo.dec(i, D_SINB);
      const int dd = d * d;
      {  // FFN + mask + enc_attn.out_proj reverse -> dO2 (s1)
        adt::BwdChainArgs a = BA(dec, p, seed);
        a.site1 = dec_site(i, 2); a.site2 = dec_site(i, 3);
        a.gy = gy; a.u = u; a.xin = a2; a.o = o2; a.saved_bf16 = lean;
        a.W0 = P + lo.dec(i, D_C2W); a.W1 = P + lo.dec(i, D_C1W); a.W2 = P + lo.dec(i, D_EOW);
        a.dW0 = Gq + lo.dec(i, D_C2W); a.dW1 = Gq + lo.dec(i, D_C1W); a.dW2 = Gq + lo.dec(i, D_EOW);
        a.db0 = Gq + lo.dec(i, D_C2B); a.db1 = Gq + lo.dec(i, D_C1B); a.db2 = Gq + lo.dec(i, D_EOB);
        a.out0 = s1;
        a.part[0] = PART(i, PS_D_C2); a.part[1] = PART(i, PS_D_C1); a.part[2] = PART(i, PS_D_EO); a.part_stride = (size_t)w.part_stride;
        a.vpart = VPART(i, 0);
        const int rc = use_seq ? adt_launch_seq_post_bwd(hd, 0, a, st) : 1;
        if (rc < 0) return rc;
        if (rc && parts) return adt_set_error(no_fallback, L, hd);
        if (rc) CK(adt_launch_bwdchain(prec, 1, a, st));
      }
      if (logits_side == 1) {
        void* s2 = nullptr;
        CK(side_enter(sd, 0, st, &s2));
        if (!prep_zeroed && adt::zero_f32_async(ws + w.rep, (size_t)NREP * w.rep_stride, (hipStream_t)s2)) return adt_set_error("replica zero");
        CK(logits_scatter(s2));
        logits_side = 2;
      }
      // cross attention core: dq2 -> s5, dkv2 -> s4
      if (lean) {
        const uint16_t* kvb = reinterpret_cast<const uint16_t*>(kv2);      // rows of 128 bf16: k2 | v2
        // dq2 / dk2 / dv2 go to k_seqtt_mid_bwd only (with `parts` a "not covered" from it is an error), which builds bf16 MFMA operands from
        // them: written as bf16 rows (the same rounding, half the bytes); s4 + d / 2 floats = 64 bf16 elements into the (k | v) row
        CK(adt_attn_bwd_saved_bf16(q2, d, kvb, 2 * d, kvb + d, 2 * d, o2, d, lse2, s1, d, B, H, L, hd, p, seed, dec_site(i, 1), b_offset,
                                   s5, d, s4, 2 * d, parts ? s4 + d / 2 : s4 + d, 2 * d, reinterpret_cast<const uint32_t*>(base + w.d_mask2), parts ? 1 : 0, st));
      } else {
        CK(adt_attn_bwd(prec, q2, d, kv2, 2 * d, kv2 + d, 2 * d, o2, d, lse2, s1, d, B, H, L, hd, 1, p, seed, dec_site(i, 1), b_offset,
                        s5, d, s4, 2 * d, s4 + d, 2 * d, reinterpret_cast<const uint32_t*>(base + w.d_mask2), st));
      }
      if (logits_side == 2) { CK(side_join(sd, 0, st)); logits_side = 0; }      // d log_feats is complete from here on
      if (logits_side == 3) {
        SideStream* const sl = ((side_sites() & 1) && !bce_merged()) ? side_stream((hipStream_t)st) : nullptr;
        if (sl && hipStreamWaitEvent((hipStream_t)st, sl->join_ev[0], 0) != hipSuccess) return adt_set_error("backward: logits join");
        logits_side = 0;
      }
      int mid_rc = 1;
      if (use_seq) {   // both projections' reverse in one launch per sequence (adt_seqpost_tt.cuh)
        adt::BwdChainArgs a = BA(dec, 0.f, nullptr);
        a.dqkv = s5; a.lddqkv = d; a.xin = a1; a.o = o1; a.saved_bf16 = lean; a.dkv2 = s4; a.f = f; a.grad_bf16 = (lean && parts) ? 1 : 0;
        a.W0 = einw; a.W1 = P + lo.dec(i, D_SOW); a.W2 = einw + dd; a.W3 = einw + 2 * dd;
        a.dW0 = geinw; a.dW1 = Gq + lo.dec(i, D_SOW); a.dW2 = geinw + dd; a.dW3 = geinw + 2 * dd;
        a.db0 = geinb; a.db1 = Gq + lo.dec(i, D_SOB); a.db2 = geinb + d; a.db3 = geinb + 2 * d;
        a.out0 = s1; a.out1 = gf; a.acc1 = 1;
        a.part[0] = PART(i, PS_D_EIN); a.part[1] = PART(i, PS_D_SO); a.part[2] = PART(i, PS_D_EIN + 1); a.part[3] = PART(i, PS_D_EIN + 2);
        a.part_stride = (size_t)w.part_stride;
        a.vpart = VPART(i, 1);
        mid_rc = adt_launch_seq_mid_bwd(hd, a, st);
        if (mid_rc < 0) return mid_rc;
        if (mid_rc && parts) return adt_set_error(no_fallback, L, hd);
      }
      if (mid_rc) {
        {  // q2 = a1 Wq^T, a1 = o1 Wo1^T  -> dO1 (s1)
          adt::BwdChainArgs a = BA(dec, 0.f, nullptr);
          a.dqkv = s5; a.lddqkv = d; a.xin = a1; a.o = o1; a.saved_bf16 = lean;
          a.W0 = einw; a.W1 = P + lo.dec(i, D_SOW);
          a.dW0 = geinw; a.dW1 = Gq + lo.dec(i, D_SOW); a.db0 = geinb; a.db1 = Gq + lo.dec(i, D_SOB);
          a.out0 = s1;
          CK(adt_launch_bwdchain(prec, 4, a, st));
        }
        {  // [k2, v2] = f Wkv^T  -> g_f +=
          adt::BwdChainArgs a = BA(dec, 0.f, nullptr);
          a.dkv2 = s4; a.f = f;
          a.W0 = einw + dd; a.W1 = einw + 2 * dd;
          a.dW0 = geinw + dd; a.dW1 = geinw + 2 * dd; a.db0 = geinb + d; a.db1 = geinb + 2 * d;
          a.out0 = gf; a.acc0 = 1;
          CK(adt_launch_bwdchain(prec, 5, a, st));
        }
      }
      bool fused_blk = false;
      if (adt_seq_supported(prec, L, d, hd)) {   // self-attention backward + layer_norm / in-projection backward in one launch per sequence
        adt::SeqBwdArgs a = seq_bwd_args(L, (int)w.B, H, dec, p, seed, dec_site(i, 0), b_offset, hd);
        a.x = x; a.gamma = P + lo.dec(i, D_LNW); a.beta = P + lo.dec(i, D_LNB); a.Win = sinw; a.bin = P + lo.dec(i, D_SINB);
        a.dO = s1; a.o = o1; a.lse = lse1; a.mask = reinterpret_cast<const uint32_t*>(base + w.d_mask1); a.dres = gy;
        a.gx = gx; a.acc = i > 0 ? 1 : 0; a.dWin = gsinw; a.dbin = gsinb; a.dgamma = Gq + lo.dec(i, D_LNW); a.dbeta = Gq + lo.dec(i, D_LNB);
        if (seeds_virtual && i > 0) {      // reconstruction pair (enc_in[nl - i], dec_x[i]): d / d dec_x[i] = -2 lambda (a - b) / n = coef * (x - a)
          a.acc = 0; a.seed_other = ws + w.enc_x + (nl - i) * Td; a.seed_coef = ws + w.norms + 8 + (nl - i);
        }
        a.nrep = NREPP; a.rep_stride = (size_t)w.prep_stride; a.wp_base = P + lo.posw(); a.wp_img = ws + w.wpack; a.saved_bf16 = lean;
        a.part = PART(i, PS_D_SIN); a.part_stride = (size_t)w.part_stride;
        a.vpart = VPART(i, 2);
        a.nsplit = a.part ? attn_split((int)w.B) : 1;
        time_mark(2, i, true, st);
        const int rc = adt_launch_seq_attn_pre_bwd(hd, 1, a, st);
        time_mark(2, i, false, st);
        if (rc < 0) return rc;
        if (rc != 0 && lean) return adt_set_error("backward: the lean forward needs the fused attention-block backward (L=%d hd=%d)", L, hd);
        fused_blk = rc == 0;
      }
      if (!fused_blk) {
        CK(adt_attn_bwd(prec, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, o1, d, lse1, s1, d, B, H, L, hd, 1, p, seed,
                        dec_site(i, 0), b_offset, s3, 3 * d, s3 + d, 3 * d, s3 + 2 * d, 3 * d,
                        reinterpret_cast<const uint32_t*>(base + w.d_mask1), st));
        {  // layer_norm + packed in_proj reverse: gx (+)= LN'(dqkv Win + gy*mask)
          adt::BwdChainArgs a = BA(dec, 0.f, nullptr);
          a.dqkv = s3; a.lddqkv = 3 * d; a.gy = gy; a.xin = x;
          a.W0 = sinw; a.W1 = sinw + dd; a.W2 = sinw + 2 * dd; a.gamma = P + lo.dec(i, D_LNW); a.beta = P + lo.dec(i, D_LNB);
          a.dW0 = gsinw; a.dW1 = gsinw + dd; a.dW2 = gsinw + 2 * dd; a.db0 = gsinb; a.db1 = gsinb + d; a.db2 = gsinb + 2 * d;
          a.dgamma = Gq + lo.dec(i, D_LNW); a.dbeta = Gq + lo.dec(i, D_LNB);
          a.out0 = gx; a.acc0 = i > 0 ? 1 : 0;
          CK(adt_launch_bwdchain(prec, 3, a, st));
        }
      }
    }
    // decoder input embedding (sasrec/model.py:53-59)
    // The decoder's input embedding gradient and the sum of the decoder blocks' partials need nothing from the encoder chain: in a
    // one-phase backward they run on the side stream under it (joined in front of the last fold); in the two-phase form the sum runs
    // beside the embedding gradient.  k_dwpart_reduce adds with atomics and k_replica_reduce2 read-modify-writes the same range of G:
    // the fold always comes behind the join.
    // (embed3: the decoder embedding's rows go out with the encoder's at the very end -- with the sums left to the optimizer's fold there is then
    // nothing for the side stream here, and no fork)
    if (phase == 0 && sd && (side_sites() & 2) && !(embed3 && (late_parts || !parts))) {
      CK(side_mark(sd, 1, st));
      dec_side = 1;                      // enqueued behind the first kernel of the encoder phase
    } else if (phase == 0) {
      if (!det && !embed3) CK(adt_embed_bwd_rep(dec, ws + w.g_dec_x, T, L, d, p, seed, SITE_EMB_DEC, ro, G + lo.posw(), ws + w.rep, NREP, w.rep_stride, st));
    } else if (det) {      // two-phase form: the decoder blocks' partial sums and the fold of the decoder range (nothing of the tables yet)
      if (parts) CK(reduce_partials(c, lo, w, G, ws, false, true, st));
      CK(adt_replica_reduce(G + dec_begin, Gq + dec_begin, lo.total - dec_begin, NREPP, w.prep_stride, st));
    } else {
      void* s2 = nullptr;
      CK(side_mark(sd, 1, st));
      if (!embed3) CK(adt_embed_bwd_rep(dec, ws + w.g_dec_x, T, L, d, p, seed, SITE_EMB_DEC, ro, G + lo.posw(), ws + w.rep, NREP, w.rep_stride, st));
      CK(side_enter(sd, 1, st, &s2));
      if (parts) CK(reduce_partials(c, lo, w, G, ws, false, true, s2));
      CK(side_join(sd, 1, st));
      CK(adt_replica_reduce2(G + lo.item(), ws + w.rep, (int64_t)(c->item_num + 1) * d, NREP, w.rep_stride, G + dec_begin, Gq + dec_begin,
                             lo.total - dec_begin, NREPP, w.prep_stride, st));
    }
  }
  if (phase == 0 || phase == 2) {
    // last_layernorm: g_enc_x[nl] = LN'(g_f) -- inside the last encoder block's first backward kernel when that is the per-sequence kernel
    // with stored vector sums (k_seqtt_post_bwd<ENC>: BwdChainArgs::lnl_x)
    const bool lnl_fused = parts && use_seq && lnl_fuse_on();      // (without the stored sums: float atomics into the replicas, like its other vectors)
    if (lnl_fused) {
      // (nothing to launch: BwdChainArgs::lnl_x of the first encoder kernel below)
    } else if (late_parts) {
      const int nb = adt_layernorm_bwd_parts(gf, d, ws + w.enc_x + nl * Td, d, P + lo.lnl_w(), LN_EPS, T, d, ws + w.g_enc_x + nl * Td, d, 0,
                                             ws + w.lnpart, LN_PART_BLOCKS, st);
      if (nb < 0) return nb;
    } else
    CK(adt_layernorm_bwd_rep(gf, d, ws + w.enc_x + nl * Td, d, P + lo.lnl_w(), LN_EPS, T, d, ws + w.g_enc_x + nl * Td, d, 0,
                             Gq + lo.lnl_w(), Gq + lo.lnl_b(), NREPP, w.prep_stride, st));
    auto dec_side_enter = [&]() -> int {      // behind the first kernel of the encoder phase (which keeps the caller's queue)
      if (dec_side != 1) return 0;
      void* s2 = nullptr;
      CK(side_enter(sd, 1, st, &s2));
      if (!det && !embed3) CK(adt_embed_bwd_rep(dec, ws + w.g_dec_x, T, L, d, p, seed, SITE_EMB_DEC, ro, G + lo.posw(), ws + w.rep, NREP, w.rep_stride, s2));
      if (parts && !late_parts) { CK(reduce_partials(c, lo, w, G, ws, false, true, s2)); dec_parts_done = true; }
      dec_side = 2;
      return 0;
    };
    if (!lnl_fused) CK(dec_side_enter());
    for (int i = nl - 1; i >= 0; --i) {
      float* gy = ws + w.g_enc_x + (i + 1) * Td;
      float* gx = ws + w.g_enc_x + i * Td;     // already holds the reconstruction seed for enc_in[i]
      const float* x = ws + w.enc_x + i * Td;
      float* base = ws + i * w.e_stride;
      float *qkv = base + w.e_qkv, *o = base + w.e_o, *lse = base + w.e_lse, *h = base + w.e_h, *u = base + w.e_u,
            *rec = base + w.e_rec;
      const float* inw = P + lo.enc(i, E_INW);
      float* ginw = Gq + lo.enc(i, E_INW);
      float* ginb = Gq + lo.enc(i, E_INB);
      const int dd = d * d;
      {  // FFN + mask + forward_layernorm + out_proj reverse -> dh (s5), dO (s1)
        adt::BwdChainArgs a = BA(seq, p, seed);
        a.site1 = enc_site(i, 1); a.site2 = enc_site(i, 2);
        a.gy = gy; a.u = u; a.xin = h; a.o = o; a.saved_bf16 = lean;
        a.W0 = P + lo.enc(i, E_C2W); a.W1 = P + lo.enc(i, E_C1W); a.W2 = P + lo.enc(i, E_OW);
        a.gamma = P + lo.enc(i, E_LN2W); a.beta = P + lo.enc(i, E_LN2B);
        a.dW0 = Gq + lo.enc(i, E_C2W); a.dW1 = Gq + lo.enc(i, E_C1W); a.dW2 = Gq + lo.enc(i, E_OW);
        a.db0 = Gq + lo.enc(i, E_C2B); a.db1 = Gq + lo.enc(i, E_C1B); a.db2 = Gq + lo.enc(i, E_OB);
        a.dgamma = Gq + lo.enc(i, E_LN2W); a.dbeta = Gq + lo.enc(i, E_LN2B);
        a.out0 = s5; a.out1 = s1;
        int which = 0;
        if (H > 1 && H <= 4) {   // independence-head classifier reverse, fused (sasrec/modules.py:648-649; main.py:160-169)
          which = H <= 2 ? 6 : 7;
          a.rec = rec; a.drec = ws + w.g_rec + i * recsz; a.Ws = P + lo.enc(i, E_SW); a.dWs = Gq + lo.enc(i, E_SW);
          a.dbs = Gq + lo.enc(i, E_SB); a.H = H;
        }
        a.part[0] = PART(i, PS_E_C2); a.part[1] = PART(i, PS_E_C1); a.part[2] = PART(i, PS_E_O); a.part_stride = (size_t)w.part_stride;
        a.vpart = VPART(i, 3);
        if (lnl_fused && i == nl - 1) {
          a.gy = gf; a.lnl_x = ws + w.enc_x + nl * Td; a.lnl_gamma = P + lo.lnl_w(); a.lnl_eps = LN_EPS;
          a.vpart2 = late_parts ? ws + w.vpart + (int64_t)(5 * nl) * w.vcall : nullptr;
          a.lnl_dgamma = Gq + lo.lnl_w(); a.lnl_dbeta = Gq + lo.lnl_b();
        }
        const int rc = use_seq ? adt_launch_seq_post_bwd(hd, 1, a, st) : 1;
        if (rc < 0) return rc;
        if (rc && parts) return adt_set_error(no_fallback, L, hd);
        if (lnl_fused && i == nl - 1) CK(dec_side_enter());
        if (rc) CK(adt_launch_bwdchain(prec, which, a, st));
      }
      if (H > 4)   // wider classifiers: separate kernel
        CK(adt_headcls_bwd(o, d, P + lo.enc(i, E_SW), rec, ws + w.g_rec + i * recsz, (int)w.B, L, H, hd, s1, d, G + lo.enc(i, E_SW),
                           G + lo.enc(i, E_SB), st));
      bool fused_blk = false;
      if (use_seq) {   // attention backward + LayerNorm / in-projection backward in one launch per sequence (adt_seqbwd_tt.cuh)
        adt::SeqBwdArgs a = seq_bwd_args(L, (int)w.B, H, seq, p, seed, enc_site(i, 0), b_offset, hd);
        a.x = x; a.gamma = P + lo.enc(i, E_LN1W); a.beta = P + lo.enc(i, E_LN1B); a.Win = inw; a.bin = P + lo.enc(i, E_INB);
        a.dO = s1; a.o = o; a.lse = lse; a.mask = reinterpret_cast<const uint32_t*>(base + w.e_mask); a.dres = s5;
        a.gx = gx; a.acc = 1; a.dWin = ginw; a.dbin = ginb; a.dgamma = Gq + lo.enc(i, E_LN1W); a.dbeta = Gq + lo.enc(i, E_LN1B);
        if (seeds_virtual) {               // reconstruction pair (enc_in[i], dec_x[nl - i]): d / d enc_in[i] = coef * (x - b)
          a.acc = 0; a.seed_other = ws + w.dec_x + (nl - i) * Td; a.seed_coef = ws + w.norms + 8 + i;
          if (i > 0) { a.seed_loss = ws + w.loss + 64 * (2 + i); a.seed_norms = ws + w.norms; }      // pair i >= 1 is not touched by the loss pass at all
        }
        a.nrep = NREPP; a.rep_stride = (size_t)w.prep_stride; a.wp_base = P + lo.posw(); a.wp_img = ws + w.wpack; a.saved_bf16 = lean;
        a.part = PART(i, PS_E_IN); a.part_stride = (size_t)w.part_stride;
        a.vpart = VPART(i, 4);
        a.nsplit = a.part ? attn_split((int)w.B) : 1;
        time_mark(1, i, true, st);
        const int rc = adt_launch_seq_attn_pre_bwd(hd, 0, a, st);
        time_mark(1, i, false, st);
        if (rc < 0) return rc;
        if (rc != 0 && lean) return adt_set_error("backward: the lean forward needs the fused attention-block backward (L=%d hd=%d)", L, hd);
        fused_blk = rc == 0;
      }
      if (!fused_blk) {
        CK(adt_attn_bwd(prec, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, o, d, lse, s1, d, (int)w.B, H, L, hd, 1, p, seed,
                        enc_site(i, 0), b_offset, s3, 3 * d, s3 + d, 3 * d, s3 + 2 * d, 3 * d,
                        reinterpret_cast<const uint32_t*>(base + w.e_mask), st));
        {  // attention_layernorm + in_proj reverse: gx += LN'(dq Wq + dh) + dk Wk + dv Wv
          adt::BwdChainArgs a = BA(seq, 0.f, nullptr);
          a.dqkv = s3; a.lddqkv = 3 * d; a.dh = s5; a.xin = x;
          a.W0 = inw; a.W1 = inw + dd; a.W2 = inw + 2 * dd; a.gamma = P + lo.enc(i, E_LN1W); a.beta = P + lo.enc(i, E_LN1B);
          a.dW0 = ginw; a.dW1 = ginw + dd; a.dW2 = ginw + 2 * dd; a.db0 = ginb; a.db1 = ginb + d; a.db2 = ginb + 2 * d;
          a.dgamma = Gq + lo.enc(i, E_LN1W); a.dbeta = Gq + lo.enc(i, E_LN1B);
          a.out0 = gx; a.acc0 = 1;
          CK(adt_launch_bwdchain(prec, 2, a, st));
        }
      }
    }
    if (phase == 2 && !det && adt::zero_f32_async(ws + w.rep, (size_t)NREP * w.rep_stride, (hipStream_t)st)) return adt_set_error("replica zero");
    // the sum of the encoder blocks' partials runs beside the embedding gradient ; the fold (read-modify-write of the same range) behind both.
    // The side stream is in order: the join behind that sum also covers the decoder's work queued on it earlier -- a join of its own in front
    // of the embedding gradient was one more cross-queue wait (5-9 us) on the caller's stream.
    {
      // (nothing to put beside the embedding gradient when the optimizer's fold sums the partials: no empty fork / join pair then)
      SideStream* const sd2 = ((side_sites() & 4) && parts && !late_parts) ? sd : nullptr;
      if (dec_side == 2 && !sd2) CK(side_join(sd, 1, st));
      dec_side = 0;
      void* s2 = nullptr;
      CK(side_mark(sd2, 2, st));
      if (det) {
        // item table: every item's rows (encoder ids, decoder ids, positive / negative items of the logits) summed by one owner in sorted
        // order ; positional table: one owner per position.  (sasrec/model.py:34-41, :53-59, :72-76 reversed)
        if (sd_sort && hipStreamWaitEvent((hipStream_t)st, sd_sort->join_ev[3], 0) != hipSuccess) return adt_set_error("backward: sort join");
        const uint32_t site4[4] = {SITE_EMB_SEQ, SITE_EMB_DEC, 0u, 0u};
        const int32_t* const ids2[2] = {seq, dec};
        const float* const dx2[2] = {ws + w.g_enc_x, ws + w.g_dec_x};
        CK(adt_item_segsum_posemb(iwork, 4, T, c->item_num + 1, 0xFu, site4, p, seed, sqrtf((float)d), G + lo.item(), prep_zeroed ? 0 : 1,
                                  ids2, dx2, site4, 2, (int)w.B, L, ro, G + lo.posw(), st));
      } else if (embed3) {      // encoder + decoder embedding rows + the positive-logit rows: one atomic row-add per token where their ids line up
        CK(adt_embed_bwd3(seq, dec, pos, ws + w.g_enc_x, ws + w.g_dec_x, f, ws + w.g_pos, T, L, p, seed, SITE_EMB_SEQ, SITE_EMB_DEC, ro, G + lo.posw(),
                          ws + w.rep, NREP, w.rep_stride, st));
      } else {
        CK(adt_embed_bwd_rep(seq, ws + w.g_enc_x, T, L, d, p, seed, SITE_EMB_SEQ, ro, G + lo.posw(), ws + w.rep, NREP, w.rep_stride, st));
      }
      CK(side_enter(sd2, 2, st, &s2));
      if (parts && !late_parts) CK(reduce_partials(c, lo, w, G, ws, true, phase == 0 && !dec_parts_done, s2));
      CK(side_join(sd2, 2, st));
    }
    if (!defer_fold)
      CK(adt_replica_reduce2(G + lo.item(), ws + w.rep, (int64_t)(c->item_num + 1) * d, det ? 0 : NREP, w.rep_stride, G + lo.posw(), Gq + lo.posw(),
                             (phase == 0 ? lo.total : dec_begin) - lo.posw(), NREPP, w.prep_stride, st));
  }
  return 0;
}

static int fold_impl(bool adam, const adt_sasrec_cfg* c, float* ws, int B, float* P, float* G, float* M, float* V, float wd, float clip, float lr,
                     float b1, float b2, float eps, float* scal, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  float* const Gq = ws + w.prep - lo.posw();
  if (fold_sums_partials(c, w)) {
    int slots[256], offs[256];
    const int ns = partial_slots(c, lo, true, true, slots, offs);
    int nwg[256];
    for (int i = 0; i < ns; ++i) nwg[i] = partial_nwg(slots[i], B);
    // the stored bias / LayerNorm / classifier sums (BwdChainArgs::vpart: the kernels' sRed layouts) -> 64-float chunks of G
    int vsrc[128], vnwg[128], vstr[128], voff[128];
    int nv = 0;
    const int nsplit_wg = B * seq_split(B), nattn_wg = B * attn_split(B), d = c->hidden;
    auto chunk = [&](int layer, int k, int t0, int nw, int64_t goff) {
      vsrc[nv] = (int)((int64_t)(5 * layer + k) * w.vcall + t0); vnwg[nv] = nw; vstr[nv] = 512; voff[nv] = (int)goff; ++nv;
    };
    for (int i = 0; i < c->num_layers; ++i) {
      chunk(i, 0, 256, nsplit_wg, lo.dec(i, D_C2B)); chunk(i, 0, 320, nsplit_wg, lo.dec(i, D_C1B)); chunk(i, 0, 384, nsplit_wg, lo.dec(i, D_EOB));
      chunk(i, 1, 0, nsplit_wg, lo.dec(i, D_EINB)); chunk(i, 1, 64, nsplit_wg, lo.dec(i, D_SOB));
      chunk(i, 1, 128, nsplit_wg, lo.dec(i, D_EINB) + d); chunk(i, 1, 192, nsplit_wg, lo.dec(i, D_EINB) + 2 * d);
      chunk(i, 2, 0, nattn_wg, lo.dec(i, D_LNW)); chunk(i, 2, 64, nattn_wg, lo.dec(i, D_LNB));
      for (int j = 0; j < 3; ++j) chunk(i, 2, 128 + 64 * j, nattn_wg, lo.dec(i, D_SINB) + j * d);
      chunk(i, 3, 0, nsplit_wg, lo.enc(i, E_LN2W)); chunk(i, 3, 64, nsplit_wg, lo.enc(i, E_LN2B));
      if (c->num_heads > 1) { chunk(i, 3, 128, nsplit_wg, lo.enc(i, E_SW)); chunk(i, 3, 192, nsplit_wg, lo.enc(i, E_SB)); }
      chunk(i, 3, 256, nsplit_wg, lo.enc(i, E_C2B)); chunk(i, 3, 320, nsplit_wg, lo.enc(i, E_C1B)); chunk(i, 3, 384, nsplit_wg, lo.enc(i, E_OB));
      chunk(i, 4, 0, nattn_wg, lo.enc(i, E_LN1W)); chunk(i, 4, 64, nattn_wg, lo.enc(i, E_LN1B));
      for (int j = 0; j < 3; ++j) chunk(i, 4, 128 + 64 * j, nattn_wg, lo.enc(i, E_INB) + j * d);
    }
    if (lnl_fuse_on()) {      // the last LayerNorm: per-workgroup sums of the first encoder kernel of the backward (BwdChainArgs::vpart2)
      chunk(c->num_layers, 0, 0, nsplit_wg, lo.lnl_w()); chunk(c->num_layers, 0, 64, nsplit_wg, lo.lnl_b());
    } else {      // per-block sums of k_ln_bwd (dgamma | dbeta, 128 floats per block)
      const int T = B * c->maxlen, nb = (T + 15) / 16 < LN_PART_BLOCKS ? (T + 15) / 16 : LN_PART_BLOCKS;
      const int base = (int)(w.lnpart - w.vpart);
      vsrc[nv] = base; vnwg[nv] = nb; vstr[nv] = 128; voff[nv] = (int)lo.lnl_w(); ++nv;
      vsrc[nv] = base + 64; vnwg[nv] = nb; vstr[nv] = 128; voff[nv] = (int)lo.lnl_b(); ++nv;
    }
    if (!adam)
      return adt_fold_parts(P, G, lo.total, G + lo.item(), ws + w.rep, (int64_t)(c->item_num + 1) * c->hidden, item_det(w) ? 0 : NREP, w.rep_stride,
                            G + lo.posw(), Gq + lo.posw(), lo.total - lo.posw(), 0, w.prep_stride, ws + w.part, w.part_stride, nwg, slots, offs, ns,
                            ws + w.vpart, vsrc, vnwg, vstr, voff, nv, ws + w.gnpart, scal, st);
    return adt_fold_parts_clip_adam(P, G, M, V, lo.total, G + lo.item(), ws + w.rep, (int64_t)(c->item_num + 1) * c->hidden, item_det(w) ? 0 : NREP,
                                    w.rep_stride, G + lo.posw(), Gq + lo.posw(), lo.total - lo.posw(), 0 /* no replicas: partials */, w.prep_stride,
                                    ws + w.part, w.part_stride, nwg, slots, offs, ns, ws + w.vpart, vsrc, vnwg, vstr, voff, nv, ws + w.gnpart,
                                    wd, clip, lr, b1, b2, eps, scal, st);
  }
  if (!adam)      // the replica fold the backward left out (phase bit 8)
    return adt_replica_reduce2(G + lo.item(), ws + w.rep, (int64_t)(c->item_num + 1) * c->hidden, item_det(w) ? 0 : NREP, w.rep_stride, G + lo.posw(),
                               Gq + lo.posw(), lo.total - lo.posw(), NREPP, w.prep_stride, st);
  return adt_fold_clip_adam(P, G, M, V, lo.total, G + lo.item(), ws + w.rep, (int64_t)(c->item_num + 1) * c->hidden, item_det(w) ? 0 : NREP, w.rep_stride, G + lo.posw(),
                            Gq + lo.posw(), lo.total - lo.posw(), NREPP, w.prep_stride, wd, clip, lr, b1, b2, eps, scal, st);
}

int adt_sasrec_fold_clip_adam(const adt_sasrec_cfg* c, float* ws, int B, float* P, float* G, float* M, float* V, float wd, float clip, float lr,
                              float b1, float b2, float eps, float* scal, void* st) {
  return fold_impl(true, c, ws, B, P, G, M, V, wd, clip, lr, b1, b2, eps, scal, st);
}

int adt_sasrec_fold_grads(const adt_sasrec_cfg* c, float* ws, int B, float* P, float* G, float* scal, void* st) {
  return fold_impl(false, c, ws, B, P, G, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, scal, st);
}

int adt_sasrec_predict(const adt_sasrec_cfg* c, const float* P, float* ws, const int32_t* seq, const int32_t* cand,
                       int B, int C, float* logits, int32_t* rank, void* st) {
  CK(check_cfg(c));
  Layout lo;
  make_layout(c, &lo);
  WS w;
  make_ws(c, B, &w);
  const int T = (int)w.T, d = (int)w.d, L = (int)w.L;
  CK(encoder_forward(c, lo, w, P, ws, seq, nullptr, nullptr, 0.f, nullptr, 0, false, st));
  // final_feat = log_feats[:, -1, :]  (sasrec/model.py:89): row b*L + L-1, i.e. ld = L*d starting at (L-1)*d
  return adt_score_rank(ws + w.f + (int64_t)(L - 1) * d, L * d, P + lo.item(), cand, B, C, d, logits, rank, st);
}


// ---- one layer per call (the supernet's candidate layers): include/adt_hip.h --------------------------------------------------------
namespace {
struct LayerSave { float *o, *h, *u, *lse, *mask, *a1, *q2, *kv2, *o2, *lse2, *mask2; int64_t total; };
LayerSave layer_save(float* base, int B, int L, int H, int dec) {
  const int64_t T = (int64_t)B * L, row = up64(T * 32), ls = up64((int64_t)B * H * L), mk = up64((int64_t)B * H * L * 8);
  LayerSave s{};
  int64_t o = 0;
  auto take = [&](int64_t n) { float* r = base ? base + o : nullptr; o += n; return r; };
  s.o = take(row); s.h = take(row); s.u = take(row); s.lse = take(ls); s.mask = take(mk);
  if (dec) { s.a1 = take(row); s.q2 = take(row); s.kv2 = take(2 * row); s.o2 = take(row); s.lse2 = take(ls); s.mask2 = take(mk); }
  s.total = o;
  return s;
}
}  // namespace

int adt_seq_layer_supported(int prec, int L, int d, int hd) { return adt_seq_lean(prec, L, d, hd) && adt_seq_partials(prec, L, d, hd) ? 1 : 0; }

int64_t adt_seq_layer_save_floats(int B, int L, int H, int dec) { return layer_save(nullptr, B, L, H, dec).total; }

int adt_seq_enc_layer_fwd(int B, int L, int H, const int32_t* ids, const float* x, const adt_enc_layer_ptrs* P, const float* wp_base,
                          const void* wp_img, float p, const uint32_t* seed, uint32_t site_attn, uint32_t site_ffn1, uint32_t site_ffn2,
                          uint32_t b_offset, int training, float* save, float* y, float y_scale, int y_acc, float* rec, void* st) {
  const int hd = 64 / H;
  if (!adt_seq_layer_supported(ADT_PREC_BF16, L, 64, hd)) return adt_set_error("seq_enc_layer_fwd: L=%d H=%d is not covered", L, H);
  const LayerSave s = layer_save(save, B, L, H, 0);
  adt::SeqFwdArgs a = seq_args(L, B, H, ids, p, seed, b_offset, hd);
  a.x = x; a.site_attn = site_attn; a.site1 = site_ffn1; a.site2 = site_ffn2;
  a.gamma = P->ln1_w; a.beta = P->ln1_b; a.Win = P->in_w; a.bin = P->in_b; a.Wo = P->out_w; a.bo = P->out_b;
  a.gamma2 = P->ln2_w; a.beta2 = P->ln2_b; a.W1 = P->c1_w; a.b1 = P->c1_b; a.W2 = P->c2_w; a.b2 = P->c2_b;
  a.y = y; a.y_scale = y_scale; a.y_acc = y_acc; a.wp_base = wp_base; a.wp_img = wp_img;
  if (training) { a.o = s.o; a.h = s.h; a.u = s.u; a.lse = s.lse; a.mask = reinterpret_cast<uint32_t*>(s.mask); a.saved_bf16 = 1; }
  else { a.lse = s.lse; }
  if (rec && H > 1) { a.rec = rec; a.Ws = P->cls_w; a.bs = P->cls_b; }
  return adt_launch_seq_enc_fwd(hd, a, st);
}

int adt_seq_enc_layer_bwd(int B, int L, int H, const int32_t* ids, const float* x, const adt_enc_layer_ptrs* P, const adt_enc_layer_ptrs* G,
                          const float* wp_base, const void* wp_img, float p, const uint32_t* seed, uint32_t site_attn, uint32_t site_ffn1,
                          uint32_t site_ffn2, uint32_t b_offset, const float* save, const float* gy, float gy_scale, const float* rec,
                          const float* drec, float* gx, int gx_acc, float* scratch, void* st) {
  const int hd = 64 / H, T = B * L;
  if (!adt_seq_layer_supported(ADT_PREC_BF16, L, 64, hd)) return adt_set_error("seq_enc_layer_bwd: L=%d H=%d is not covered", L, H);
  const LayerSave s = layer_save(const_cast<float*>(save), B, L, H, 0);
  float *dO = scratch, *dh = scratch + up64((int64_t)T * 64);
  {  // FFN + mask + forward_layernorm + out_proj (+ head classifier) reverse -> dh, dO
    adt::BwdChainArgs a = bwd_args(T, L, B, ids, p, seed, b_offset * (uint32_t)L);
    a.site1 = site_ffn1; a.site2 = site_ffn2; a.gy = gy; a.gy_scale = gy_scale; a.u = s.u; a.xin = s.h; a.o = s.o; a.saved_bf16 = 1;
    a.W0 = P->c2_w; a.W1 = P->c1_w; a.W2 = P->out_w; a.gamma = P->ln2_w; a.beta = P->ln2_b;
    a.dW0 = G->c2_w; a.dW1 = G->c1_w; a.dW2 = G->out_w; a.db0 = G->c2_b; a.db1 = G->c1_b; a.db2 = G->out_b; a.dgamma = G->ln2_w; a.dbeta = G->ln2_b;
    a.out0 = dh; a.out1 = dO; a.wp_base = wp_base; a.wp_img = wp_img;
    if (H > 1 && drec) { a.rec = rec; a.drec = drec; a.Ws = P->cls_w; a.dWs = G->cls_w; a.dbs = G->cls_b; a.H = H; }
    const int rc = adt_launch_seq_post_bwd(hd, 1, a, st);
    if (rc) return rc < 0 ? rc : adt_set_error("seq_enc_layer_bwd: post chain not covered");
  }
  adt::SeqBwdArgs a = seq_bwd_args(L, B, H, ids, p, seed, site_attn, b_offset, hd);
  a.x = x; a.gamma = P->ln1_w; a.beta = P->ln1_b; a.Win = P->in_w; a.bin = P->in_b;
  a.dO = dO; a.o = s.o; a.lse = s.lse; a.mask = reinterpret_cast<const uint32_t*>(s.mask); a.dres = dh; a.saved_bf16 = 1;
  a.gx = gx; a.acc = gx_acc; a.dWin = G->in_w; a.dbin = G->in_b; a.dgamma = G->ln1_w; a.dbeta = G->ln1_b;
  a.wp_base = wp_base; a.wp_img = wp_img;
  const int rc = adt_launch_seq_attn_pre_bwd(hd, 0, a, st);
  return rc <= 0 ? rc : adt_set_error("seq_enc_layer_bwd: attention block not covered");
}

int adt_seq_dec_layer_fwd(int B, int L, int H, const int32_t* ids, const float* x, const float* feats, const adt_dec_layer_ptrs* P,
                          const float* wp_base, const void* wp_img, float p, const uint32_t* seed, uint32_t site_slf, uint32_t site_enc,
                          uint32_t site_ffn1, uint32_t site_ffn2, uint32_t b_offset, float* save, float* y, float y_scale, int y_acc, void* st) {
  const int hd = 64 / H;
  if (!adt_seq_layer_supported(ADT_PREC_BF16, L, 64, hd)) return adt_set_error("seq_dec_layer_fwd: L=%d H=%d is not covered", L, H);
  const LayerSave s = layer_save(save, B, L, H, 1);
  adt::SeqFwdArgs a = seq_args(L, B, H, ids, p, seed, b_offset, hd);
  a.x = x; a.f = feats; a.site_attn = site_slf; a.site_attn2 = site_enc; a.site1 = site_ffn1; a.site2 = site_ffn2;
  a.gamma = P->ln_w; a.beta = P->ln_b; a.Win = P->sin_w; a.bin = P->sin_b; a.Wo = P->so_w; a.bo = P->so_b;
  a.Win2 = P->ein_w; a.bin2 = P->ein_b; a.Wo2 = P->eo_w; a.bo2 = P->eo_b; a.W1 = P->c1_w; a.b1 = P->c1_b; a.W2 = P->c2_w; a.b2 = P->c2_b;
  a.o = s.o; a.lse = s.lse; a.mask = reinterpret_cast<uint32_t*>(s.mask); a.a1 = s.a1; a.q2 = s.q2; a.kv2 = s.kv2; a.o2 = s.o2;
  a.lse2 = s.lse2; a.mask2 = reinterpret_cast<uint32_t*>(s.mask2); a.h = s.h; a.u = s.u; a.saved_bf16 = 1;
  a.y = y; a.y_scale = y_scale; a.y_acc = y_acc; a.wp_base = wp_base; a.wp_img = wp_img;
  return adt_launch_seq_dec_fwd(hd, a, st);
}

int adt_seq_dec_layer_bwd(int B, int L, int H, const int32_t* ids, const float* x, const float* feats, const adt_dec_layer_ptrs* P,
                          const adt_dec_layer_ptrs* G, const float* wp_base, const void* wp_img, float p, const uint32_t* seed,
                          uint32_t site_slf, uint32_t site_enc, uint32_t site_ffn1, uint32_t site_ffn2, uint32_t b_offset, const float* save,
                          const float* gy, float gy_scale, float* gx, int gx_acc, float* gfeats, float* scratch, void* st) {
  const int hd = 64 / H, T = B * L, d = 64;
  if (!adt_seq_layer_supported(ADT_PREC_BF16, L, 64, hd)) return adt_set_error("seq_dec_layer_bwd: L=%d H=%d is not covered", L, H);
  const LayerSave s = layer_save(const_cast<float*>(save), B, L, H, 1);
  const int64_t Td = up64((int64_t)T * 64);
  float *s1 = scratch, *s5 = scratch + Td, *s4 = scratch + 2 * Td;      // dO (T x 64), dq2 (T x 64), dk2 | dv2 (T x 128)
  const uint32_t ro = b_offset * (uint32_t)L;
  {  // FFN + mask + enc_attn.out_proj reverse -> dO2 (s1)
    adt::BwdChainArgs a = bwd_args(T, L, B, ids, p, seed, ro);
    a.site1 = site_ffn1; a.site2 = site_ffn2; a.gy = gy; a.gy_scale = gy_scale; a.u = s.u; a.xin = s.h; a.o = s.o2; a.saved_bf16 = 1;
    a.W0 = P->c2_w; a.W1 = P->c1_w; a.W2 = P->eo_w; a.dW0 = G->c2_w; a.dW1 = G->c1_w; a.dW2 = G->eo_w;
    a.db0 = G->c2_b; a.db1 = G->c1_b; a.db2 = G->eo_b; a.out0 = s1; a.wp_base = wp_base; a.wp_img = wp_img;
    const int rc = adt_launch_seq_post_bwd(hd, 0, a, st);
    if (rc) return rc < 0 ? rc : adt_set_error("seq_dec_layer_bwd: post chain not covered");
  }
  {  // cross attention core: dq2 -> s5, dk2 | dv2 -> s4
    const uint16_t* kvb = reinterpret_cast<const uint16_t*>(s.kv2);
    CK(adt_attn_bwd_saved_bf16(s.q2, d, kvb, 2 * d, kvb + d, 2 * d, s.o2, d, s.lse2, s1, d, B, H, L, hd, p, seed, site_enc, b_offset, s5, d, s4,
                               2 * d, s4 + d / 2, 2 * d, reinterpret_cast<const uint32_t*>(s.mask2), 1, st));      // bf16 rows for k_seqtt_mid_bwd below
  }
  {  // enc_attn q / k / v projections and slf_attn.out_proj reverse -> dO1 (s1), d feats +=
    adt::BwdChainArgs a = bwd_args(T, L, B, ids, 0.f, nullptr, ro);
    a.dqkv = s5; a.lddqkv = d; a.xin = s.a1; a.o = s.o; a.saved_bf16 = 1; a.dkv2 = s4; a.f = feats; a.grad_bf16 = 1;
    a.W0 = P->ein_w; a.W1 = P->so_w; a.W2 = P->ein_w + d * d; a.W3 = P->ein_w + 2 * d * d;
    a.dW0 = G->ein_w; a.dW1 = G->so_w; a.dW2 = G->ein_w + d * d; a.dW3 = G->ein_w + 2 * d * d;
    a.db0 = G->ein_b; a.db1 = G->so_b; a.db2 = G->ein_b + d; a.db3 = G->ein_b + 2 * d;
    a.out0 = s1; a.out1 = gfeats; a.acc1 = 1; a.wp_base = wp_base; a.wp_img = wp_img;
    const int rc = adt_launch_seq_mid_bwd(hd, a, st);
    if (rc) return rc < 0 ? rc : adt_set_error("seq_dec_layer_bwd: mid chain not covered");
  }
  adt::SeqBwdArgs a = seq_bwd_args(L, B, H, ids, p, seed, site_slf, b_offset, hd);
  a.x = x; a.gamma = P->ln_w; a.beta = P->ln_b; a.Win = P->sin_w; a.bin = P->sin_b;
  a.dO = s1; a.o = s.o; a.lse = s.lse; a.mask = reinterpret_cast<const uint32_t*>(s.mask); a.dres = gy; a.dres_scale = gy_scale; a.saved_bf16 = 1;
  a.gx = gx; a.acc = gx_acc; a.dWin = G->sin_w; a.dbin = G->sin_b; a.dgamma = G->ln_w; a.dbeta = G->ln_b;
  a.wp_base = wp_base; a.wp_img = wp_img;
  const int rc = adt_launch_seq_attn_pre_bwd(hd, 1, a, st);
  return rc <= 0 ? rc : adt_set_error("seq_dec_layer_bwd: attention block not covered");
}

}  // extern "C"

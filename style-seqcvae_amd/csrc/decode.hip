// Eval-mode decode step and constrained-beam-search bookkeeping for gfx950.
// Reference: UpDownCaptioner._decode_step with training=False
// (var_updown/var_updown/models/updown_captioner.py:371-455), UpDownCell.forward eval branch
// (var_updown/var_updown/modules/updown_cell.py:200-229), ConstrainedBeamSearch.search
// (updown-baseline/updown/modules/cbs.py:59-277).
//
// Differences in mechanism, not in result: the reference re-materialises
// feats.repeat(...) for every beam row on every step, so both of its lru-caches miss and avg / pv are
// recomputed per step; here per-image terms are computed once (ssc_decode_prepare) and rows index
// their image (row g -> image g / rows_per_image, batch-major).
#include <algorithm>
#include <initializer_list>

#include "ssc_common.h"
#include "beam_common.h"

namespace {

inline size_t r4(size_t x) { return (x + 3) & ~(size_t)3; }
inline size_t r64(size_t x) { return (x + 63) & ~(size_t)63; }

struct ImgLayout {
  size_t mask, avg, pv, ga_avg, wsum_att, wsum_dec, wz, wc, emb_gates, pd, scales, scratch, scratch_floats, total;
  // 2xFP16 products of the large calls: the weights that meet a recurrent state, already split into their fp16 pieces with the scale
  // of their product (ssc_split_f16; Hk = H rounded up to 32 four-byte words per row) - the product kernels then copy the pieces
  // instead of forming them in each of the 40-150 row tiles that read a weight tile (ssc_gemm_seg.B16)
  size_t pw_att_h1, pw_att_hd, pw_dec_hd, pw_dec_h1, pw_q, pw_out;
  int Hk;
  bool planes;
  int Fp, Hp, Zp, Sp;
  bool token_table;   // emb_gates holds the (V, 4H) table emb . W_ih^att[:, :E]^T
  bool att_table;     // pd holds P[img, r, :] = W_ih^dec[:, :F] v_{img,r} (nimg*R x 4H): the decoder gates' attended-feature term per region
};
// The embedding's contribution to the attention LSTM gates depends on the token only: from this many images per call on it is
// formed once per call for the whole vocabulary ((V, E) x (E, 4H): 96 GFLOP at C4, 0.5 ms) and the cell kernel picks the row
// of the beam's last token (ssc_lstm_fwd_desc.add0_rows), instead of a K = E segment of the gate product in every step
// (5000 x 4800 x 1000 per step at C4: 0.25 ms x 20 steps).  Below it (a handful of rows per step) the segment is cheaper.
constexpr int DEC_TOKEN_TABLE_MIN_IMAGES = 8;
constexpr int DEDUP_MAX_ROWS = 1 << 20;   // rows of one decode step that go through the parent / live-row lists
}  // namespace
int ssc_g_beam_reg = ssc_env_int("SSC_BEAM_REG", 1);   // 0: the LDS-staged selection kernel for every vocabulary size
int ssc_g_dec_dedup = ssc_env_int("SSC_DEC_DEDUP", 1);   // ssc_debug_set("dec_dedup"): products fed only by the parent's states run on distinct parents
int ssc_g_dec_att_table = ssc_env_int("SSC_DEC_ATT_TABLE", 1);   // ssc_debug_set("dec_att_table"): 0 = attended features + K = F segment in every step
namespace {
ImgLayout img_layout(const ssc_model_cfg* c, int nimg, int R) {
  ImgLayout l;
  l.Fp = (int)r4(c->F);
  size_t o = 0;
  l.mask = o; o += r64((size_t)nimg * R);
  l.avg = o; o += r64((size_t)nimg * l.Fp);
  l.pv = o; o += r64((size_t)nimg * R * c->A);
  l.ga_avg = o; o += r64((size_t)nimg * 4 * c->H);
  // weight views that stay fixed for the whole decode (prepared with the image context): the pre-summed recurrent blocks
  // W_ih[:, h-block] + W_hh of the attention / decoder LSTM (both multiply the same state) and a 16-B aligned copy of the
  // z-block of W_ih^dec (it starts at an odd column when the sentiment column is present)
  l.Hp = (int)r4(c->H); l.Zp = (int)r4(c->Z);
  l.wsum_att = o; o += r64((size_t)4 * c->H * l.Hp);
  l.wsum_dec = o; o += r64((size_t)4 * c->H * l.Hp);
  l.wz = o; o += r64((size_t)4 * c->H * l.Zp);
  // SENTIMENT_VAE = 2 with the whole pooled attribute vector as conditioning (S = Z columns): their block of W_ih^dec, 16-byte rows
  l.Sp = (int)r4(c->S);
  l.wc = o; o += r64(c->kld_mode == 2 && c->S > 1 ? (size_t)4 * c->H * l.Sp : 0);
  l.token_table = nimg >= DEC_TOKEN_TABLE_MIN_IMAGES;
  l.emb_gates = o; o += r64(l.token_table ? (size_t)c->V * 4 * c->H : 0);
  // The attended-feature segment of the decoder gate product, att . W_ih^dec[:, :F]^T with att = sum_r alpha_r v_r, is linear in
  // att: sum_r alpha_r (v_r . W^T).  The R region terms of an image are formed once per call (one (nimg R) x 4H x F product) and
  // the decoder cell contracts them with the step's attention weights (ssc_lstm_fwd_img, K = R): the largest product of a decode
  // step (K = F + 2H + Z = 4576) loses its K = F = 2048 segment, and the weighted feature sum itself is no longer needed.
  l.att_table = ssc_g_dec_att_table != 0 && R <= 128;
  l.pd = o; o += r64(l.att_table ? (size_t)nimg * R * 4 * c->H : 0);
  l.Hk = (c->H + 31) / 32 * 32;
  l.planes = c->gemm_mode == 3 && l.token_table && !c->tied;
  const size_t wrow = l.planes ? (size_t)l.Hk : 0;
  l.pw_att_h1 = o; o += r64((size_t)4 * c->H * wrow);
  l.pw_att_hd = o; o += r64((size_t)4 * c->H * wrow);
  l.pw_dec_hd = o; o += r64((size_t)4 * c->H * wrow);
  l.pw_dec_h1 = o; o += r64((size_t)4 * c->H * wrow);
  l.pw_q = o; o += r64((size_t)c->A * wrow);
  l.pw_out = o; o += r64((size_t)c->V * wrow);
  l.scales = o; o += 64;   // power-of-two operand scales of the 2xFP16 products (ssc_model_cfg.gemm_mode 3), see DecScales
  size_t a = (size_t)nimg * R * c->A, b = (size_t)nimg * 4 * c->H;
  l.scratch_floats = 33 * (a > b ? a : b);
  l.scratch = o; o += r64(l.scratch_floats);
  l.total = o;
  return l;
}

struct StepLayout {
  size_t emb, q, att, z, pm, c1, attn_logits, proj, wcol, slabs, slab_floats, total;
  size_t p_h1, p_hd, p_h1o, p_hdo;   // gemm_mode 3: the recurrent states (previous h1, hd; new h1, hd) split into their fp16 pieces (ssc_gemm_seg.A16), G x Hk words
  int Hk;
  size_t dedup;   // int32: [0] = number of distinct parents, [1] = number of live rows, [4 .. 4+G) = the parents' representative rows (ascending),
                  // [4+G .. 4+2G) = slot of every row, [4+2G ..) = previous-state row of every row, [4+3G ..) = live rows (ascending)
  int Ep, Ap, Fp, Zp;
};
StepLayout step_layout(const ssc_model_cfg* c, int G, int R) {
  StepLayout l;
  l.Ep = (int)r4(c->E); l.Ap = (int)r4(c->A); l.Fp = (int)r4(c->F); l.Zp = (int)r4(c->Z);
  size_t o = 0;
  l.emb = o; o += r64((size_t)G * l.Ep);
  l.q = o; o += r64((size_t)G * l.Ap);
  l.att = o; o += r64((size_t)G * l.Fp);
  l.z = o; o += r64((size_t)G * l.Zp);
  l.pm = o; o += r64(c->kld_mode == 2 ? (size_t)G * l.Zp : 0);   // SENTIMENT_VAE = 2: the pooled prior mean of every row
  l.c1 = o; o += r64(c->kld_mode == 2 ? (size_t)G : 0);           //   and its first entry (the one conditioning column of "senti_word_net")
  l.attn_logits = o; o += r64((size_t)G * R);
  l.proj = o; o += r64(c->tied ? (size_t)G * l.Ep : 0);
  l.wcol = o; o += r64((size_t)4 * c->H);
  l.Hk = (c->H + 31) / 32 * 32;
  const size_t prow = c->gemm_mode == 3 && !c->tied && G >= 512 ? (size_t)l.Hk : 0;
  l.p_h1 = o; o += r64((size_t)G * prow);
  l.p_hd = o; o += r64((size_t)G * prow);
  l.p_h1o = o; o += r64((size_t)G * prow);
  l.p_hdo = o; o += r64((size_t)G * prow);
  l.dedup = o; o += r64((size_t)6 * G + 8);   // (+ two counts per workgroup of the list kernels)
  size_t skinny = (size_t)33 * G * 4 * c->H;
  size_t full = (size_t)16 * 1024 * 1024;  // 64 MB: split-K slabs of the large GEMMs
  l.slab_floats = skinny > full ? skinny : full;
  l.slabs = o; o += r64(l.slab_floats);
  l.total = o;
  return l;
}

struct Seg {
  const float* A; int lda;
  const float* B; int ldb;
  int K;
  const float* A16 = nullptr; const float* B16 = nullptr;   // the operands' pre-split fp16 pieces (ld16 words per row), or nullptr
  int ld16 = 0;
};

// Power-of-two operand scales for the 2xFP16 form of the large products (ssc_gemm_desc.a_scale / b_scale), kept in the image buffer:
// recurrent states, z, the projected hidden state lie in [-1, 1] or a few units around it -> a constant 2^6; everything that comes
// from data or from the checkpoint (weights, embedding table, region features) is measured once per image context (ssc_pow2_scale,
// largest magnitude -> [2^12, 2^13]).  One pair per product: operands of one product that differ in scale share the smaller factor.
enum { SC_ACT = 0, SC_ACTF, SC_ATTW, SC_Q, SC_DEC, SC_OUT, SC_PROJ, SC_EMB, SC_WV, SC_FEAT, SC_TMP, SC_COUNT };   // (SC_ATTW .. SC_WV: functions of the parameters alone)
struct DecScales {
  const float* base;   // nullptr: no scales (every factor 1)
  const float* at(int i) const { return base ? base + i : nullptr; }
};

void fill_desc(ssc_gemm_desc& d, std::initializer_list<Seg> segs, int M, int N) {
  d = ssc_gemm_desc{};
  int i = 0;
  for (const Seg& s : segs) {
    if (s.K <= 0) continue;
    d.seg[i].A = s.A; d.seg[i].lda = s.lda; d.seg[i].B = s.B; d.seg[i].ldb = s.ldb; d.seg[i].K = s.K;
    d.seg[i].A16 = s.A16; d.seg[i].lda16 = s.A16 ? s.ld16 : 0; d.seg[i].B16 = s.B16; d.seg[i].ldb16 = s.B16 ? s.ld16 : 0;
    ++i;
  }
  d.nseg = i; d.M = M; d.N = N; d.a_kc = 1; d.b_kc = 1;
}

int gemm_nt(hipStream_t st, float* ws, size_t ws_floats, std::initializer_list<Seg> segs, int M, int N, float* Cc, int ldc,
            const float* bias = nullptr, const float* sa = nullptr, const float* sb = nullptr) {
  ssc_gemm_desc d;
  fill_desc(d, segs, M, N);
  d.C = Cc; d.ldc = ldc; d.bias = bias; d.splits = 0; d.workspace = ws; d.workspace_floats = ws_floats;
  d.a_scale = sa; d.b_scale = sb;
  return ssc_gemm(&d, st);
}

int gemm_slabs(hipStream_t st, float* ws, size_t ws_floats, std::initializer_list<Seg> segs, int M, int N, int* nslab,
               const int* m_count = nullptr, const int* a_rows = nullptr, const float* sa = nullptr, const float* sb = nullptr) {
  ssc_gemm_desc d;
  fill_desc(d, segs, M, N);
  d.m_count = m_count; d.a_rows = a_rows;   // product over the listed A rows only, output rows compact (slab row i <-> a_rows[i])
  d.a_scale = sa; d.b_scale = sb;
  return ssc_gemm_slabs_auto(&d, ws, ws_floats, nslab, st);
}

// ---------------------------------------------------------------------------------------------------
// beam-search kernels.  Selection order: value descending, index ascending ("k-pass selection":
// pass k finds the best candidate strictly after the previous pick in that order).
// ---------------------------------------------------------------------------------------------------
// NORM: `lp` holds un-normalised logits; the row's log-sum-exp is taken here with exactly the arithmetic of
// log_softmax_kernel (256 threads, strided partial maxima / sums, block_reduce order), so lp[v] - lse is bit-identical to
// what ssc_log_softmax would have stored.  The row is staged in LDS when it fits (`staged`), so HBM sees it once.
template <bool NORM>
__device__ __forceinline__ const float* row_prepare(const float* __restrict__ row, int V, bool staged, float* srow, float* shr,
                                                    float& lse) {
  lse = 0.f;
  if (!NORM) return row;
  float mx = -INFINITY;
  // thread t owns v = t, t + 256, ... (the same assignment and order as log_softmax_kernel: bit-identical sums); its loads are
  // issued eight at a time - one at a time, this 40 KB row cost ~40 dependent memory round trips per thread
  for (int v0 = threadIdx.x; v0 < V; v0 += 8 * blockDim.x) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = row[min(v0 + u * (int)blockDim.x, V - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = v0 + u * (int)blockDim.x;
      if (v < V) {
        if (staged) srow[v] = x[u];
        mx = fmaxf(mx, x[u]);
      }
    }
  }
  mx = dec_block_reduce(mx, shr, true);   // (its barriers also publish srow)
  const float* src = staged ? srow : row;
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += expf(src[v] - mx);
  s = dec_block_reduce(s, shr, false);
  lse = mx + logf(s);
  return src;
}

// first step: per (b, s): top-`beam` over v of (fsm[b,0,s,v] ? lp[b,v] : -inf)       cbs.py:127-145
template <bool NORM>
__global__ __launch_bounds__(256) void beam_first_kernel(const float* __restrict__ lp, int ldlp,
                                                         const uint8_t* __restrict__ fsm, const int* __restrict__ mach, int S,
                                                         int V, int beam, int64_t* __restrict__ pred,
                                                         float* __restrict__ lp_out, int staged) {
  extern __shared__ float srow[];
  __shared__ Cand sh[4];
  __shared__ float shr[16];
  int b = blockIdx.x / S, s = blockIdx.x % S;
  float lse;
  const float* row = row_prepare<NORM>(lp + (size_t)b * ldlp, V, staged != 0, srow, shr, lse);
  const uint8_t* m = fsm ? fsm + (((size_t)(mach ? mach[b] : b) * S + 0) * S + s) * V : nullptr;   // nullptr: the trivial one-state machine
  Cand prev{INFINITY, -1};
  for (int k = 0; k < beam; ++k) {
    Cand best{-INFINITY, -1};
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      float x = (!m || m[v]) ? (NORM ? row[v] - lse : row[v]) : -INFINITY;
      bool after_prev = (prev.i < 0) || (x < prev.v) || (x == prev.v && v > prev.i);
      if (after_prev && (best.i < 0 || better(x, v, best))) best = Cand{x, v};
    }
    best = block_best(best, sh);
    if (threadIdx.x == 0) {
      pred[(size_t)blockIdx.x * beam + k] = best.i;
      lp_out[(size_t)blockIdx.x * beam + k] = best.v;
    }
    prev = best;
  }
}

// later steps, part A: per (source row g=(b,s,k), target state i): masked top-`per_node`     cbs.py:177-209
template <bool NORM>
__global__ __launch_bounds__(256) void beam_row_topk_kernel(const float* __restrict__ lp, int ldlp,
                                                            const uint8_t* __restrict__ fsm, const int* __restrict__ mach,
                                                            const int64_t* __restrict__ last_pred, int S, int V, int beam,
                                                            int per_node, int end_index, float* __restrict__ sval,
                                                            int64_t* __restrict__ sidx, int staged) {
  extern __shared__ float srow[];
  __shared__ Cand sh[4];
  __shared__ float shr[16];
  int g = blockIdx.x, i = blockIdx.y;
  int b = g / (S * beam), s = (g / beam) % S, k = g % beam;
  const uint8_t* m = fsm ? fsm + (((size_t)(mach ? mach[b] : b) * S + s) * S + i) * V : nullptr;
  bool ended = last_pred[g] == end_index;   // workgroup-uniform; an ended beam never looks at its row
  float lse = 0.f;
  const float* row = lp + (size_t)g * ldlp;
  if (!ended) row = row_prepare<NORM>(row, V, staged != 0, srow, shr, lse);
  // scratch layout (b, i, s, k, n)
  size_t base = ((((size_t)b * S + i) * S + s) * beam + k) * per_node;
  Cand prev{INFINITY, -1};
  for (int n = 0; n < per_node; ++n) {
    Cand best{-INFINITY, -1};
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      float x;
      if (m && !m[v]) x = -1e20f;
      else if (ended) x = v == end_index ? 0.f : -INFINITY;
      else x = NORM ? row[v] - lse : row[v];
      bool after_prev = (prev.i < 0) || (x < prev.v) || (x == prev.v && v > prev.i);
      if (after_prev && (best.i < 0 || better(x, v, best))) best = Cand{x, v};
    }
    best = block_best(best, sh);
    if (threadIdx.x == 0) {
      sval[base + n] = best.v;
      sidx[base + n] = best.i;
    }
    prev = best;
  }
}

// The same selection with the row in REGISTERS (V <= 256 * BEAM_REG_NV): thread t holds v = t, t + 256, ... - the assignment of
// row_prepare, so the maxima, the sums (same per-thread order, same block reduction) and therefore lse and lp[v] - lse are
// bit-identical to it.  One memory round trip per row (all loads of a thread in flight at once), no LDS row: the staged form
// took five dependent load batches, four passes over a 40 KB LDS row and three workgroups per CU (160 us for 5000 x 10000).
constexpr int BEAM_REG_NV = 40;
template <bool NORM>
__global__ __launch_bounds__(256) void beam_row_topk_reg_kernel(const float* __restrict__ lp, int ldlp,
                                                                const uint8_t* __restrict__ fsm, const int* __restrict__ mach,
                                                                const int64_t* __restrict__ last_pred, int S, int V, int beam,
                                                                int per_node, int end_index, float* __restrict__ sval,
                                                                int64_t* __restrict__ sidx) {
  __shared__ Cand sh[4];
  __shared__ float shr[16];
  const int g = blockIdx.x, i = blockIdx.y, t = threadIdx.x;
  const int b = g / (S * beam), s = (g / beam) % S, k = g % beam;
  const uint8_t* m = fsm ? fsm + (((size_t)(mach ? mach[b] : b) * S + s) * S + i) * V : nullptr;
  const bool ended = last_pred[g] == end_index;   // workgroup-uniform; an ended beam never looks at its row
  const float* row = lp + (size_t)g * ldlp;
  float x[BEAM_REG_NV];
  float lse = 0.f;
  if (!ended) {
#pragma unroll
    for (int u = 0; u < BEAM_REG_NV; ++u) x[u] = row[min(t + u * 256, V - 1)];
    if (NORM) {
      float mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < BEAM_REG_NV; ++u)
        if (t + u * 256 < V) mx = fmaxf(mx, x[u]);
      mx = dec_block_reduce(mx, shr, true);
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < BEAM_REG_NV; ++u)
        if (t + u * 256 < V) sum += expf(x[u] - mx);
      sum = dec_block_reduce(sum, shr, false);
      lse = mx + logf(sum);
    }
    if (NORM) {
#pragma unroll
      for (int u = 0; u < BEAM_REG_NV; ++u) x[u] -= lse;
    }
  } else {
#pragma unroll
    for (int u = 0; u < BEAM_REG_NV; ++u) x[u] = t + u * 256 == end_index ? 0.f : -INFINITY;
  }
  if (m) {   // (uniform; the mask bytes of a thread all in flight at once)
    uint8_t mk[BEAM_REG_NV];
#pragma unroll
    for (int u = 0; u < BEAM_REG_NV; ++u) mk[u] = m[min(t + u * 256, V - 1)];
#pragma unroll
    for (int u = 0; u < BEAM_REG_NV; ++u)
      if (!mk[u]) x[u] = -1e20f;
  }
  const size_t base = ((((size_t)b * S + i) * S + s) * beam + k) * per_node;   // scratch layout (b, i, s, k, n)
  Cand prev{INFINITY, -1};
  for (int n = 0; n < per_node; ++n) {
    Cand best{-INFINITY, -1};
#pragma unroll
    for (int u = 0; u < BEAM_REG_NV; ++u) {
      const int v = t + u * 256;
      const float y = x[u];
      const bool after_prev = (prev.i < 0) || (y < prev.v) || (y == prev.v && v > prev.i);
      if (v < V && after_prev && (best.i < 0 || better(y, v, best))) best = Cand{y, v};
    }
    best = block_best(best, sh);
    if (t == 0) {
      sval[base + n] = best.v;
      sidx[base + n] = best.i;
    }
    prev = best;
  }
}

// back-trace of the beam (cbs.py:252-277): preds (steps,B,SB), backptrs (steps-1,B,SB) -> out (B,SB,steps)
__global__ void beam_backtrace_kernel(const int64_t* __restrict__ preds, const int64_t* __restrict__ backptrs, int steps,
                                      int B, int SB, int64_t* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B * SB) return;
  int b = j / SB;
  int64_t idx = j % SB;
  size_t plane = (size_t)B * SB;
  int64_t* o = out + (size_t)j * steps;
  for (int t = steps - 1; t >= 0; --t) {
    o[t] = preds[(size_t)t * plane + (size_t)b * SB + idx];
    if (t > 0) idx = backptrs[(size_t)(t - 1) * plane + (size_t)b * SB + idx];
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int ld, const int64_t* __restrict__ backptr,
                                   int rows_per_batch, int Wd, float* __restrict__ dst) {
  int row = blockIdx.y;
  int b = row / rows_per_batch;
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Wd) return;
  size_t srow = (size_t)b * rows_per_batch + backptr[row];
  dst[(size_t)row * ld + x] = src[srow * ld + x];
}

// Beams of one group (S * beam consecutive rows) that descend from the same parent hold identical recurrent states after the
// re-ordering of cbs.py:236-250.  rep = the first live row of each (group, parent) class; out[0] = number of classes,
// out[4 .. ) their representative rows in ascending order, out[4 + n .. ) the class index (slot) of every row.
// out[4 + 2n .. ): prow = the row of every row's previous state; `ungathered` (the states are still in the previous step's row
// order): prow[g] = (g - g % group) + parent[g] and the representative rows are the classes' parent rows; else prow[g] = g.
// row_lp (optional): only LIVE rows count - a row whose running log-prob is <= -1e19 (no finite beam: ssc_beam_desc.skip_dead) or
// whose last token is END (an ended beam re-emits END whatever its logits are, cbs.py:177-181) needs no decode step: its class is
// not listed (unless a live row shares it), its slot is 0, and out[1] / out[4 + 3n .. ) = the number / the ascending list of live
// rows for the products that run on every (live) row.  The parent of a live row was itself live, so live rows never read a row
// that was skipped.
// Two launches over workgroups of whole groups (<= 256 rows each, a thread per row): PHASE 0 counts every workgroup's classes and
// live rows, PHASE 1 sums the counts of the workgroups before it and writes the lists.  (Round 3's single-workgroup form searched
// every row's group serially: 8 us at 10000 rows in groups of 5, 400 us at 19200 rows in groups of 80 - a constrained search.)
template <int PHASE>
__global__ __launch_bounds__(256) void dedup_rows_kernel(const int64_t* __restrict__ parent, int n, int group, int gpb,
                                                         int* __restrict__ out, int* __restrict__ counts, int ungathered,
                                                         const float* __restrict__ row_lp, const int64_t* __restrict__ tokens,
                                                         int end_index) {
  __shared__ int first[256];
  __shared__ int s_slot[256];
  __shared__ int wsum[2][4];
  __shared__ int red[2][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rpb = gpb * group;   // rows of this workgroup (<= 256)
  const int g = blockIdx.x * rpb + t;
  const bool valid = t < rpb && g < n;
  int* urows = out + 4;
  int* slot = out + 4 + n;
  int* prow = out + 4 + 2 * n;
  int* lrows = out + 4 + 3 * n;
  first[t] = 0x7fffffff;
  const bool live = valid && (!row_lp || (row_lp[g] > -1e19f && tokens[g] != end_index));
  const int p = valid ? (int)parent[g] : 0;
  const int cls = (t / group) * group + p;   // (p < group: a back-pointer within the group)
  __syncthreads();
  if (live) atomicMin(&first[cls], t);
  __syncthreads();
  const int f = live ? first[cls] : -1;
  const bool rep = live && f == t;
  const unsigned long long brep = __ballot(rep), blive = __ballot(live);
  if (lane == 0) { wsum[0][wave] = __popcll(brep); wsum[1][wave] = __popcll(blive); }
  __syncthreads();
  const int nrep = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
  const int nlive = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
  if (PHASE == 0) {
    if (t == 0) { counts[2 * blockIdx.x] = nrep; counts[2 * blockIdx.x + 1] = nlive; }
    if (valid) { urows[g] = 0; lrows[g] = 0; }   // entries past the counts are never used; keep them in range
    return;
  }
  int orep = 0, olive = 0;   // the classes / live rows of the workgroups before this one
  for (int j = t; j < (int)blockIdx.x; j += 256) { orep += counts[2 * j]; olive += counts[2 * j + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { orep += __shfl_xor(orep, o, 64); olive += __shfl_xor(olive, o, 64); }
  if (lane == 0) { red[0][wave] = orep; red[1][wave] = olive; }
  __syncthreads();
  orep = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  olive = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  int rrep = __popcll(brep & below), rlive = __popcll(blive & below);
  for (int w = 0; w < wave; ++w) { rrep += wsum[0][w]; rlive += wsum[1][w]; }
  const int prev_row = g - g % group + p;
  if (rep) {
    urows[orep + rrep] = ungathered ? prev_row : g;
    s_slot[t] = orep + rrep;
  }
  if (live) lrows[olive + rlive] = g;
  __syncthreads();
  if (valid) {
    slot[g] = live ? s_slot[f] : 0;
    prow[g] = (ungathered && live) ? prev_row : g;
  }
  if (blockIdx.x == gridDim.x - 1 && t == 0) { out[0] = orep + nrep; out[1] = olive + nlive; }
}

__global__ void dec_add2d_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int cols,
                                 float* __restrict__ o, int ldo) {
  int r = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < cols) o[(size_t)r * ldo + x] = a[(size_t)r * lda + x] + b[(size_t)r * ldb + x];
}

}  // namespace

int ssc_decode_att_table_enabled() { return ssc_g_dec_att_table != 0; }
int ssc_g_dec_parts = ssc_env_int("SSC_DEC_PARTS", 1);   // ssc_debug_set("dec_parts"): 0 = the searches of ssc_decode_search always write logits
int ssc_decode_parts_enabled() { return ssc_g_dec_parts != 0; }

extern "C" size_t ssc_decode_image_bytes(const ssc_model_cfg* cfg, int nimg, int R) {
  if (!cfg || nimg <= 0 || R <= 0) return 0;
  return img_layout(cfg, nimg, R).total * sizeof(float);
}

extern "C" int ssc_decode_prepare(const ssc_model_cfg* cfg, const ssc_params* p, const float* feats, int nimg, int R,
                                  void* imgbuf, size_t imgbuf_bytes, void* stream) {
  return ssc_decode_prepare_from(cfg, p, feats, nimg, R, imgbuf, imgbuf_bytes, nullptr, 0, 0, stream);
}

extern "C" int ssc_decode_prepare_from(const ssc_model_cfg* cfg, const ssc_params* p, const float* feats, int nimg, int R,
                                       void* imgbuf, size_t imgbuf_bytes, const void* prev_imgbuf, int prev_nimg, int prev_R,
                                       void* stream) {
  SscGemmModeScope mode_scope(cfg);   // the numerics mode of this cfg, for every product the call issues
  if (!cfg || !p || !feats || !imgbuf || nimg <= 0 || R <= 0 || R > 256) return SSC_EINVAL;
  if (prev_imgbuf && (prev_nimg <= 0 || prev_R <= 0 || prev_imgbuf == imgbuf)) return SSC_EINVAL;
  ImgLayout l = img_layout(cfg, nimg, R);
  if (imgbuf_bytes < l.total * sizeof(float)) return SSC_EWORKSPACE;
  float* W = (float*)imgbuf;
  hipStream_t st = (hipStream_t)stream;
  const int F = cfg->F, A = cfg->A, E = cfg->E, H4 = 4 * cfg->H;
  SSC_TRY(ssc_feat_prep(feats, nimg, R, F, W + l.mask, W + l.avg, st));
  // 2xFP16 products (cfg->gemm_mode 3): the operands' power-of-two scales (DecScales), measured here once per image context.
  // Weight groups whose blocks are also pre-summed (W_ih[:, h-block] + W_hh) get one bit of head room.
  float* SC = W + l.scales;
  if (cfg->gemm_mode == 3) {
    const int H = cfg->H;
    float* tmp = SC + SC_TMP;
    SSC_TRY(ssc_fill(SC + SC_ACT, 1, 64.f, st));
    SSC_TRY(ssc_fill(SC + SC_ACTF, 1, 64.f, st));
    SSC_TRY(ssc_pow2_scale(feats, (size_t)nimg * R, F, F, 13, SC + SC_FEAT, 0, tmp, st));
    SSC_TRY(ssc_pow2_scale(feats, (size_t)nimg * R, F, F, 13, SC + SC_ACTF, 1, tmp, st));   // products that mix states and attended features
  }
  if (cfg->gemm_mode == 3 && prev_imgbuf) {
    // the weights' scales depend on the parameters alone: an earlier context of the same, unchanged parameters hands them over
    // (the caller's promise, as for the per-token gate table below) - measuring them reads every weight once, ~1 ms at C4's sizes
    const ImgLayout pl0 = img_layout(cfg, prev_nimg, prev_R);
    if (hipMemcpyAsync(SC + SC_ATTW, (const float*)prev_imgbuf + pl0.scales + SC_ATTW, (size_t)(SC_FEAT - SC_ATTW) * sizeof(float),
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
      return SSC_EHIP;
  } else if (cfg->gemm_mode == 3) {
    const int H = cfg->H;
    float* tmp = SC + SC_TMP;
    SSC_TRY(ssc_pow2_scale(p->att_w_ih, H4, E + F + 2 * H, p->ld_att_w_ih, 12, SC + SC_ATTW, 0, tmp, st));
    SSC_TRY(ssc_pow2_scale(p->att_w_hh, H4, H, p->ld_att_w_hh, 12, SC + SC_ATTW, 1, tmp, st));
    SSC_TRY(ssc_pow2_scale(p->dec_w_ih, H4, F + 2 * H + cfg->S + cfg->Z, p->ld_dec_w_ih, 12, SC + SC_DEC, 0, tmp, st));
    SSC_TRY(ssc_pow2_scale(p->dec_w_hh, H4, H, p->ld_dec_w_hh, 12, SC + SC_DEC, 1, tmp, st));
    SSC_TRY(ssc_pow2_scale(p->wq, A, H, p->ld_wq, 13, SC + SC_Q, 0, tmp, st));
    SSC_TRY(ssc_pow2_scale(p->wv, A, F, p->ld_wv, 13, SC + SC_WV, 0, tmp, st));
    SSC_TRY(ssc_pow2_scale(p->emb, cfg->V, E, p->ld_emb, 13, SC + SC_EMB, 0, tmp, st));
    if (cfg->tied) {
      SSC_TRY(ssc_pow2_scale(p->emb, cfg->V, E, p->ld_emb, 13, SC + SC_OUT, 0, tmp, st));
      SSC_TRY(ssc_pow2_scale(p->proj_w, E, H, p->ld_proj_w, 13, SC + SC_PROJ, 0, tmp, st));
    } else {
      SSC_TRY(ssc_pow2_scale(p->out_w, cfg->V, H, p->ld_out_w, 13, SC + SC_OUT, 0, tmp, st));
    }
  }
  const DecScales sc{cfg->gemm_mode == 3 ? SC : nullptr};
  SSC_TRY(gemm_nt(st, W + l.scratch, l.scratch_floats, {{feats, F, p->wv, p->ld_wv, F}}, nimg * R, A, W + l.pv, A, nullptr,
                  sc.at(SC_FEAT), sc.at(SC_WV)));
  SSC_TRY(gemm_nt(st, W + l.scratch, l.scratch_floats, {{W + l.avg, F, p->att_w_ih + E, p->ld_att_w_ih, F}}, nimg, H4,
                  W + l.ga_avg, H4, nullptr, sc.at(SC_FEAT), sc.at(SC_ATTW)));
  if (l.token_table) {
    // the per-token gate table is a function of the weights alone (V x 4H x E: 0.6 ms at C4's sizes, 1.3 % of a 50-image call):
    // an earlier context of the same, unchanged parameters hands it over by a copy
    const ImgLayout pl = prev_imgbuf ? img_layout(cfg, prev_nimg, prev_R) : ImgLayout{};
    if (prev_imgbuf && pl.token_table) {
      if (hipMemcpyAsync(W + l.emb_gates, (const float*)prev_imgbuf + pl.emb_gates, (size_t)cfg->V * H4 * sizeof(float),
                         hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SSC_EHIP;
    } else {
      SSC_TRY(gemm_nt(st, W + l.scratch, l.scratch_floats, {{p->emb, p->ld_emb, p->att_w_ih, p->ld_att_w_ih, E}}, cfg->V, H4,
                      W + l.emb_gates, H4, nullptr, sc.at(SC_EMB), sc.at(SC_ATTW)));
    }
  }
  // (the attended-feature table l.pd is formed by the first decode step that asks for it: ssc_decode_step_desc.att_table = 2)
  {
    const int H = cfg->H, Z = cfg->Z, S = cfg->S;
    dim3 grid(ssc_cdiv(H, 256), H4);
    SSC_LAUNCH(dec_add2d_kernel, grid, dim3(256), 0, st, p->att_w_ih + E + F, p->ld_att_w_ih, p->att_w_hh, p->ld_att_w_hh, H,
                       W + l.wsum_att, l.Hp);
    SSC_CHECK_LAUNCH();
    SSC_LAUNCH(dec_add2d_kernel, grid, dim3(256), 0, st, p->dec_w_ih + F + H, p->ld_dec_w_ih, p->dec_w_hh, p->ld_dec_w_hh, H,
                       W + l.wsum_dec, l.Hp);
    SSC_CHECK_LAUNCH();
    // (zero pad columns: the z / conditioning K-segments of the decode step run over r4(Z) / r4(S) columns - a K that is no multiple
    // of 4 would push its segment onto the 4 B/lane kernel, in a launch of its own)
    if (l.Zp != Z && hipMemsetAsync(W + l.wz, 0, (size_t)H4 * l.Zp * sizeof(float), st) != hipSuccess) return SSC_EHIP;
    if (hipMemcpy2DAsync(W + l.wz, (size_t)l.Zp * sizeof(float), p->dec_w_ih + F + 2 * H + S, (size_t)p->ld_dec_w_ih * sizeof(float),
                         (size_t)Z * sizeof(float), H4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return SSC_EHIP;
    if (cfg->kld_mode == 2 && S > 1) {
      if (hipMemsetAsync(W + l.wc, 0, (size_t)H4 * l.Sp * sizeof(float), st) != hipSuccess) return SSC_EHIP;
      if (hipMemcpy2DAsync(W + l.wc, (size_t)l.Sp * sizeof(float), p->dec_w_ih + F + 2 * H, (size_t)p->ld_dec_w_ih * sizeof(float),
                           (size_t)S * sizeof(float), H4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SSC_EHIP;
    }
    if (l.planes) {   // (after the sums above and the scales: all on this stream)
      const float* wr = p->att_w_ih + E + F;
      SSC_TRY(ssc_split_f16(W + l.wsum_att, H4, H, l.Hp, SC + SC_ATTW, W + l.pw_att_h1, l.Hk, nullptr, nullptr, st));
      SSC_TRY(ssc_split_f16(wr + H, H4, H, p->ld_att_w_ih, SC + SC_ATTW, W + l.pw_att_hd, l.Hk, nullptr, nullptr, st));
      SSC_TRY(ssc_split_f16(W + l.wsum_dec, H4, H, l.Hp, SC + SC_DEC, W + l.pw_dec_hd, l.Hk, nullptr, nullptr, st));
      SSC_TRY(ssc_split_f16(p->dec_w_ih + F, H4, H, p->ld_dec_w_ih, SC + SC_DEC, W + l.pw_dec_h1, l.Hk, nullptr, nullptr, st));
      SSC_TRY(ssc_split_f16(p->wq, A, H, p->ld_wq, SC + SC_Q, W + l.pw_q, l.Hk, nullptr, nullptr, st));
      SSC_TRY(ssc_split_f16(p->out_w, cfg->V, H, p->ld_out_w, SC + SC_OUT, W + l.pw_out, l.Hk, nullptr, nullptr, st));
    }
  }
  return SSC_OK;
}

extern "C" int ssc_decode_planes_ld(const ssc_model_cfg* cfg) { return cfg ? (cfg->H + 31) / 32 * 32 : 0; }

extern "C" size_t ssc_decode_step_workspace_bytes(const ssc_model_cfg* cfg, int G, int R) {
  if (!cfg || G <= 0 || R <= 0) return 0;
  return step_layout(cfg, G, R).total * sizeof(float);
}

int ssc_g_dec_planes = ssc_env_int("SSC_DEC_PLANES", 1);   // ssc_debug_set("dec_planes"): 0 = the 2xFP16 products split their operands themselves
int ssc_g_dec_ungathered = ssc_env_int("SSC_DEC_UNGATHERED", 1);   // ssc_debug_set("dec_ungathered"): 0 = the caller re-orders the states
extern "C" int ssc_decode_ungathered_ok(const ssc_model_cfg* cfg, int nimg, int G, int group, int att_table) {
  if (!cfg || nimg <= 0 || G <= 0) return 0;
  // the conditions under which ssc_decode_step reads every previous state through its row lists (`dedup` there, and the table)
  if (group != 0 && (group < 2 || G % group != 0)) return 0;   // (group 0: asked before the group size is known)
  return ssc_g_dec_ungathered && att_table != 0 && ssc_g_dec_att_table != 0 &&
         nimg >= DEC_TOKEN_TABLE_MIN_IMAGES && G >= 512 && G <= DEDUP_MAX_ROWS && group <= 255 && ssc_g_dec_dedup && cfg->H % 4 == 0;
}

extern "C" int ssc_decode_step(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_decode_step_desc* d, void* workspace,
                               size_t workspace_bytes, void* stream) {
  SscGemmModeScope mode_scope(cfg);   // the numerics mode of this cfg, for every product the call issues
  if (!cfg || !p || !d || !workspace) return SSC_EINVAL;
  const int G = d->G, R = d->R, rpi = d->rows_per_image;
  if (G <= 0 || R <= 0 || R > 256 || rpi <= 0 || G % rpi != 0) return SSC_EINVAL;
  if (!d->feats || !d->imgbuf || !d->tokens || !d->eps || !d->h1 || !d->c1 || !d->hd || !d->cd || !d->h1_out ||
      !d->c1_out || !d->hd_out || !d->cd_out || !d->alpha)
    return SSC_EINVAL;
  const bool sv2 = cfg->kld_mode == 2;   // SENTIMENT_VAE = 2: prior mean and conditioning from the attention-pooled attribute means
  if (sv2 ? (!d->obj_atts || (cfg->S != 1 && cfg->S != cfg->Z) || cfg->pm_scale != 0.f) : ((cfg->S || cfg->pm_scale != 0.f) && !d->sentiment))
    return SSC_EINVAL;
  const int nimg = G / rpi;
  const ImgLayout il = img_layout(cfg, nimg, R);
  const StepLayout l = step_layout(cfg, G, R);
  if (workspace_bytes < l.total * sizeof(float)) return SSC_EWORKSPACE;
  float* W = (float*)workspace;
  const float* I = (const float*)d->imgbuf;
  hipStream_t st = (hipStream_t)stream;
  const int E = cfg->E, H = cfg->H, A = cfg->A, F = cfg->F, Z = cfg->Z, S = cfg->S, V = cfg->V, H4 = 4 * H;
  float* slabs = W + l.slabs;
  int ns = 0;
  if (d->att_table < 0 || d->att_table > 2 || (d->att_table && !il.att_table)) return SSC_EINVAL;
  const bool att_table = d->att_table != 0;
  const DecScales sc{cfg->gemm_mode == 3 ? I + il.scales : nullptr};   // operand scales of the 2xFP16 products (ssc_decode_prepare)
  if (d->att_table == 2)   // P[img, r, :] = W_ih^dec[:, :F] v_{img,r}: one (nimg R) x 4H x F product per image context
    SSC_TRY(gemm_nt(st, slabs, l.slab_floats, {{d->feats, F, p->dec_w_ih, p->ld_dec_w_ih, F}}, nimg * R, H4,
                    const_cast<float*>(I) + il.pd, H4, nullptr, sc.at(SC_FEAT), sc.at(SC_DEC)));

  // Beams that share their parent: the products fed only by the previous step's h1 / hd run on the distinct parents
  // (ssc_decode_step_desc.parent).  Used where the token's gate term is not part of the product (token table) and the rows are
  // many enough for the 128x128 kernels with device-side row lists.
  const bool dedup = d->parent && d->group > 1 && d->group <= 255 && G <= DEDUP_MAX_ROWS && G % d->group == 0 && il.token_table &&
                     !d->emb_override && G >= 512 && ssc_g_dec_dedup && H % 4 == 0;   // (16-byte operand rows: the products over the row lists are then ONE launch, one slab)
  const int* ucount = nullptr; const int* urows = nullptr; const int* slot = nullptr; const int* prow = nullptr;
  // rows that need no step at all (ssc_decode_step_desc.row_lp): every product runs on the live rows / live parent classes only; the
  // per-row kernels still visit the skipped rows and leave garbage there, which no live row ever reads
  const bool live = dedup && att_table && d->row_lp && !cfg->tied;
  const int* lcount = nullptr; const int* lrows = nullptr;
  // un-gathered states: every reader of h1 / c1 / hd / cd must go through the row lists
  if (d->ungathered && !(dedup && att_table)) return SSC_EINVAL;
  if (dedup) {
    int* dd = reinterpret_cast<int*>(W + l.dedup);
    const int gpb = std::max(1, 256 / d->group), nwg = ssc_cdiv(G, gpb * d->group);
    int* counts = dd + 4 + 4 * G;
    SSC_LAUNCH(dedup_rows_kernel<0>, dim3(nwg), dim3(256), 0, st, d->parent, G, d->group, gpb, dd, counts, d->ungathered ? 1 : 0,
               live ? d->row_lp : nullptr, d->tokens, d->end_index);
    SSC_CHECK_LAUNCH();
    SSC_LAUNCH(dedup_rows_kernel<1>, dim3(nwg), dim3(256), 0, st, d->parent, G, d->group, gpb, dd, counts, d->ungathered ? 1 : 0,
               live ? d->row_lp : nullptr, d->tokens, d->end_index);
    SSC_CHECK_LAUNCH();
    ucount = dd; urows = dd + 4; slot = dd + 4 + G; prow = dd + 4 + 2 * G;
    if (live) { lcount = dd + 1; lrows = dd + 4 + 3 * G; }
  }
  // a product over every LIVE row, written to the rows' own places (ssc_gemm with one row list for A and C; one launch, no split)
  auto gemm_live = [&](std::initializer_list<Seg> segs, int N, float* Cc, int ldc, const float* bias, const float* sa, const float* sb) -> int {
    ssc_gemm_desc g;
    fill_desc(g, segs, G, N);
    g.C = Cc; g.ldc = ldc; g.bias = bias; g.splits = 1; g.workspace = nullptr; g.workspace_floats = 0;
    g.m_count = lcount; g.a_rows = lrows; g.c_rows = lrows;
    g.a_scale = sa; g.b_scale = sb;
    return ssc_gemm(&g, st);
  };
  float* slabs_u = slabs + (size_t)G * H4;   // second product of the decoder gates (distinct parents): behind the first one's rows
  // 2xFP16 products of a large call: the states are split into their fp16 pieces once per step (the distinct parents' previous
  // states here, the new h1 / hd behind their cells) instead of in every column tile of the products that read them (38-79 times);
  // the weights' pieces come from the image context.  Every other form of a product ignores the pieces.
  const bool planes = il.planes && dedup && att_table && ssc_g_dec_planes && l.p_hd != l.p_h1;
  const int Hk = l.Hk;
  const float* PH1 = nullptr; const float* PHD = nullptr; float* PH1O = nullptr; float* PHDO = nullptr;
  auto PW = [&](size_t off) -> const float* { return planes ? I + off : nullptr; };
  if (planes) {
    // the previous states' pieces: the caller's (what the previous step's cells left), else split here - the distinct parents' rows only
    PH1 = static_cast<const float*>(d->h1_planes); PHD = static_cast<const float*>(d->hd_planes);
    if (!PH1) { SSC_TRY(ssc_split_f16(d->h1, G, H, H, sc.at(SC_ACT), W + l.p_h1, Hk, urows, ucount, st)); PH1 = W + l.p_h1; }
    if (!PHD) { SSC_TRY(ssc_split_f16(d->hd, G, H, H, sc.at(SC_ACT), W + l.p_hd, Hk, urows, ucount, st)); PHD = W + l.p_hd; }
    // the new states' pieces are written by the cells themselves (ssc_lstm_fwd_desc.h_planes)
    PH1O = d->h1_planes_out ? static_cast<float*>(d->h1_planes_out) : W + l.p_h1o;
    PHDO = d->hd_planes_out ? static_cast<float*>(d->hd_planes_out) : W + l.p_hdo;
    if ((reinterpret_cast<uintptr_t>(PH1) | reinterpret_cast<uintptr_t>(PHD) | reinterpret_cast<uintptr_t>(PH1O) | reinterpret_cast<uintptr_t>(PHDO)) & 15) return SSC_EALIGN;
  }

  // embedding + attention LSTM (updown_captioner.py:430, updown_cell.py:143-148)
  {
    const float* wr = p->att_w_ih + E + F;
    ssc_lstm_fwd_desc f{};
    if (il.token_table && !d->emb_override) {   // the embedding's gate term comes from the per-token table, row = the beam's last token
      SSC_TRY(gemm_slabs(st, slabs, l.slab_floats, {{d->h1, H, I + il.wsum_att, il.Hp, H, PH1, PW(il.pw_att_h1), Hk},
                                                    {d->hd, H, wr + H, p->ld_att_w_ih, H, PHD, PW(il.pw_att_hd), Hk}}, G, H4,
                         &ns, ucount, urows, sc.at(SC_ACT), sc.at(SC_ATTW)));
      if (dedup && ns != 1) return SSC_EINVAL;   // (row lists and split-K slabs do not combine; G >= 512 never splits)
      f.slab_rows = slot;
      f.add0 = I + il.emb_gates; f.ld_add0 = H4; f.add0_rows = d->tokens;
    } else {
      SSC_TRY(ssc_embed_gather(p->emb, p->ld_emb, d->tokens, G, E, W + l.emb, l.Ep, st));
      SSC_TRY(gemm_slabs(st, slabs, l.slab_floats,
                         {{W + l.emb, l.Ep, p->att_w_ih, p->ld_att_w_ih, E}, {d->h1, H, I + il.wsum_att, il.Hp, H},
                          {d->hd, H, wr + H, p->ld_att_w_ih, H}}, G, H4, &ns, nullptr, nullptr, sc.at(SC_ACT), sc.at(SC_ATTW)));
    }
    f.B = G; f.H = H;
    f.slabs = slabs; f.nslab = ns; f.slab_stride = (size_t)G * H4;
    f.add1 = I + il.ga_avg; f.ld_add1 = H4; f.rows_per_add1 = rpi;
    f.b_ih = p->att_b_ih; f.b_hh = p->att_b_hh;
    f.c_prev = d->c1; f.ld_cprev = H; f.c_prev_rows = prow;
    f.c_out = d->c1_out; f.ld_cout = H; f.h_out = d->h1_out; f.ld_hout = H;
    if (live) { f.rows = lrows; f.row_count = lcount; }   // (rows nobody reads are not computed: their h1 / c1 stay stale)
    if (planes) { f.h_planes = PH1O; f.ld_hplanes = Hk; f.planes_scale = sc.at(SC_ACT); }
    SSC_TRY(ssc_lstm_fwd(&f, st));
  }
  // attention over the image's regions (attention.py:69-95, updown_cell.py:151-158)
  if (live) SSC_TRY(gemm_live({{d->h1_out, H, p->wq, p->ld_wq, H, PH1O, PW(il.pw_q), Hk}}, A, W + l.q, l.Ap, nullptr, sc.at(SC_ACT), sc.at(SC_Q)));
  else SSC_TRY(gemm_nt(st, slabs, l.slab_floats, {{d->h1_out, H, p->wq, p->ld_wq, H, PH1O, PW(il.pw_q), Hk}}, G, A, W + l.q, l.Ap, nullptr, sc.at(SC_ACT), sc.at(SC_Q)));
  if (att_table && live)
    SSC_TRY(ssc_attn_weights_rows(W + l.q, l.Ap, I + il.pv, p->wa, I + il.mask, G, R, A, rpi, W + l.attn_logits, d->alpha, lrows, lcount, st));
  else if (att_table)
    SSC_TRY(ssc_attn_weights(W + l.q, l.Ap, I + il.pv, p->wa, I + il.mask, G, R, A, rpi, W + l.attn_logits, d->alpha, st));
  else
    SSC_TRY(ssc_attn_fwd(W + l.q, l.Ap, I + il.pv, p->wa, I + il.mask, d->feats, G, R, A, F, rpi, W + l.attn_logits, d->alpha,
                         W + l.att, l.Fp, st));
  // z ~ N(prior_mean, prior_var) (updown_cell.py:200-208)
  if (sv2) {   // prior mean = sum_r alpha_r obj_atts_r (updown_cell.py:160-163)
    SSC_TRY(ssc_attn_pool(d->alpha, d->obj_atts, G, R, Z, rpi, W + l.pm, l.Zp, st));
    SSC_TRY(ssc_latent_prior_sample_pm(d->eps, Z, W + l.pm, l.Zp, d->prior_var, Z, nullptr, 0.f, cfg->prior_var, G, Z, W + l.z, l.Zp, st));
    if (S == 1) SSC_TRY(ssc_copy_strided(W + l.pm, l.Zp, G, W + l.c1, st));   // c = prior_mean[:, 0] (updown_cell.py:171-172)
    if (d->prior_mean_out && hipMemcpy2DAsync(d->prior_mean_out, (size_t)Z * sizeof(float), W + l.pm, (size_t)l.Zp * sizeof(float),
                                              (size_t)Z * sizeof(float), G, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return SSC_EHIP;
  } else if (d->prior_mean || d->prior_var) {   // the caller's own prior (updown_captioner.py:371-381)
    SSC_TRY(ssc_latent_prior_sample_pm(d->eps, Z, d->prior_mean, Z, d->prior_var, Z, cfg->pm_scale != 0.f ? d->sentiment : nullptr,
                                       cfg->pm_scale, cfg->prior_var, G, Z, W + l.z, l.Zp, st));
  } else
    SSC_TRY(ssc_latent_prior_sample(d->eps, Z, cfg->pm_scale != 0.f ? d->sentiment : nullptr, cfg->pm_scale, cfg->prior_var,
                                    G, Z, W + l.z, l.Zp, st));
  // decoder LSTM (updown_cell.py:211-229)
  {
    int ns_u = 0;
    const int KC = sv2 && S > 1 ? il.Sp : 0;   // conditioning block as a K-segment over its 16-byte padded width (a segment with K = 0 is dropped)
    const Seg cseg{W + l.pm, l.Zp, I + il.wc, il.Sp, KC};
    if (att_table && dedup) {   // hd' segment on the distinct parents, [h1 | z] on every row
      SSC_TRY(gemm_slabs(st, slabs_u, l.slab_floats - (size_t)G * H4, {{d->hd, H, I + il.wsum_dec, il.Hp, H, PHD, PW(il.pw_dec_hd), Hk}}, G, H4, &ns_u, ucount, urows,
                         sc.at(SC_ACT), sc.at(SC_DEC)));
      if (ns_u != 1) return SSC_EINVAL;
      if (live) {
        SSC_TRY(gemm_live({{d->h1_out, H, p->dec_w_ih + F, p->ld_dec_w_ih, H, PH1O, PW(il.pw_dec_h1), Hk}, {W + l.z, l.Zp, I + il.wz, il.Zp, l.Zp}, cseg}, H4, slabs, H4, nullptr,
                          sc.at(SC_ACT), sc.at(SC_DEC)));
        ns = 1;
      } else
        SSC_TRY(gemm_slabs(st, slabs, (size_t)G * H4, {{d->h1_out, H, p->dec_w_ih + F, p->ld_dec_w_ih, H, PH1O, PW(il.pw_dec_h1), Hk}, {W + l.z, l.Zp, I + il.wz, il.Zp, l.Zp}, cseg},
                           G, H4, &ns, nullptr, nullptr, sc.at(SC_ACT), sc.at(SC_DEC)));
    } else if (att_table)   // the attended-feature segment comes from the per-image table inside the cell kernel
      SSC_TRY(gemm_slabs(st, slabs, l.slab_floats,
                         {{d->h1_out, H, p->dec_w_ih + F, p->ld_dec_w_ih, H}, {d->hd, H, I + il.wsum_dec, il.Hp, H},
                          {W + l.z, l.Zp, I + il.wz, il.Zp, l.Zp}, cseg}, G, H4, &ns, nullptr, nullptr, sc.at(SC_ACT), sc.at(SC_DEC)));
    else
      SSC_TRY(gemm_slabs(st, slabs, l.slab_floats,
                         {{W + l.att, l.Fp, p->dec_w_ih, p->ld_dec_w_ih, F}, {d->h1_out, H, p->dec_w_ih + F, p->ld_dec_w_ih, H},
                          {d->hd, H, I + il.wsum_dec, il.Hp, H}, {W + l.z, l.Zp, I + il.wz, il.Zp, l.Zp}, cseg}, G, H4, &ns, nullptr, nullptr,
                         sc.at(SC_ACTF), sc.at(SC_DEC)));
    ssc_lstm_fwd_desc f{};
    f.B = G; f.H = H;
    f.slabs = slabs; f.nslab = ns; f.slab_stride = (size_t)G * H4;
    if (ns_u) { f.slabs2 = slabs_u; f.nslab2 = ns_u; f.slab2_stride = (size_t)G * H4; f.slab2_rows = slot; }
    f.b_ih = p->dec_b_ih; f.b_hh = p->dec_b_hh;
    if (S == 1) {   // one conditioning column: the sentiment (SENTIMENT_VAE = 1) or the pooled prior mean's first entry (2, "senti_word_net")
      SSC_TRY(ssc_copy_strided(p->dec_w_ih + F + 2 * H, p->ld_dec_w_ih, H4, W + l.wcol, st));
      f.sent = sv2 ? W + l.c1 : d->sentiment; f.wcol = W + l.wcol; f.ldwcol = 1;
    }
    f.c_prev = d->cd; f.ld_cprev = H; f.c_prev_rows = prow;
    f.c_out = d->cd_out; f.ld_cout = H; f.h_out = d->hd_out; f.ld_hout = H;
    if (planes) { f.h_planes = PHDO; f.ld_hplanes = Hk; f.planes_scale = sc.at(SC_ACT); }
    if (att_table) SSC_TRY(ssc_lstm_fwd_img(&f, d->alpha, R, I + il.pd, R, rpi, st));
    else SSC_TRY(ssc_lstm_fwd(&f, st));
  }
  // vocabulary log-probabilities (updown_captioner.py:444-450); skipped when only the cell output is wanted
  if (d->topk_part) {   // records per (row, 128-column tile) instead of the (G, V) logits (ssc_beam_step_parts selects from them)
    if (cfg->tied || cfg->gemm_mode == 2) return SSC_EINVAL;
    ssc_gemm_desc g;
    fill_desc(g, {{d->hd_out, H, p->out_w, p->ld_out_w, H, PHDO, PW(il.pw_out), Hk}}, G, V);
    g.bias = p->out_b; g.splits = 1; g.topk_part = d->topk_part;
    g.a_scale = sc.at(SC_ACT); g.b_scale = sc.at(SC_OUT);
    if (live) { g.m_count = lcount; g.a_rows = lrows; g.c_rows = lrows; }
    return ssc_gemm(&g, st);
  }
  if (!d->log_probs) return SSC_OK;
  if (cfg->tied) {
    SSC_TRY(gemm_nt(st, slabs, l.slab_floats, {{d->hd_out, H, p->proj_w, p->ld_proj_w, H}}, G, E, W + l.proj, l.Ep, nullptr,
                    sc.at(SC_ACT), sc.at(SC_PROJ)));
    SSC_TRY(ssc_bias_tanh(W + l.proj, l.Ep, G, E, p->proj_b, st));
    SSC_TRY(gemm_nt(st, slabs, l.slab_floats, {{W + l.proj, l.Ep, p->emb, p->ld_emb, E}}, G, V, d->log_probs, V, nullptr,
                    sc.at(SC_ACT), sc.at(SC_OUT)));
  } else if (live) {
    SSC_TRY(gemm_live({{d->hd_out, H, p->out_w, p->ld_out_w, H, PHDO, PW(il.pw_out), Hk}}, V, d->log_probs, V, p->out_b, sc.at(SC_ACT), sc.at(SC_OUT)));
  } else {
    SSC_TRY(gemm_nt(st, slabs, l.slab_floats, {{d->hd_out, H, p->out_w, p->ld_out_w, H, PHDO, PW(il.pw_out), Hk}}, G, V, d->log_probs, V, p->out_b,
                    sc.at(SC_ACT), sc.at(SC_OUT)));
  }
  if (!d->raw_logits) SSC_TRY(ssc_log_softmax(d->log_probs, V, G, V, d->log_probs, V, st));
  return SSC_OK;
}

namespace {
constexpr size_t BEAM_STAGE_MAX = 64 * 1024;   // a vocabulary row is staged in LDS up to this size
}  // namespace
// first step (dense scans: once per search); mach (B) or NULL: the machine of batch entry b
int ssc_beam_first_dense(bool norm, const float* lp, int ldlp, const uint8_t* fsm, const int* mach, int B, int S, int V, int beam,
                         int64_t* pred, float* lp_out, hipStream_t st) {
  if (!lp || (!fsm && S != 1) || !pred || !lp_out || B <= 0 || S <= 0 || V <= 0 || beam <= 0 || beam > V || ldlp < V) return SSC_EINVAL;
  const int staged = norm && (size_t)V * sizeof(float) <= BEAM_STAGE_MAX;
  const size_t lds = staged ? (size_t)V * sizeof(float) : 0;
  if (norm) SSC_LAUNCH(beam_first_kernel<true>, dim3(B * S), dim3(256), lds, st, lp, ldlp, fsm, mach, S, V, beam, pred, lp_out, staged);
  else SSC_LAUNCH(beam_first_kernel<false>, dim3(B * S), dim3(256), 0, st, lp, ldlp, fsm, mach, S, V, beam, pred, lp_out, 0);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}
// later steps, part A with the dense machine: one masked scan per (row, target state)
int ssc_beam_rows_dense(bool norm, const float* lp, int ldlp, const uint8_t* fsm, const int* mach, const int64_t* last_pred, int B,
                        int S, int V, int beam, int per_node, int end_index, float* scratch_val, int64_t* scratch_idx,
                        hipStream_t st) {
  const int staged = norm && (size_t)V * sizeof(float) <= BEAM_STAGE_MAX;
  const size_t lds = staged ? (size_t)V * sizeof(float) : 0;
  if (V <= 256 * BEAM_REG_NV && ssc_g_beam_reg) {
    if (norm)
      SSC_LAUNCH(beam_row_topk_reg_kernel<true>, dim3(B * S * beam, S), dim3(256), 0, st, lp, ldlp, fsm, mach, last_pred, S, V, beam,
                 per_node, end_index, scratch_val, scratch_idx);
    else
      SSC_LAUNCH(beam_row_topk_reg_kernel<false>, dim3(B * S * beam, S), dim3(256), 0, st, lp, ldlp, fsm, mach, last_pred, S, V, beam,
                 per_node, end_index, scratch_val, scratch_idx);
  } else if (norm)
    SSC_LAUNCH(beam_row_topk_kernel<true>, dim3(B * S * beam, S), dim3(256), lds, st, lp, ldlp, fsm, mach, last_pred, S, V, beam,
                       per_node, end_index, scratch_val, scratch_idx, staged);
  else
    SSC_LAUNCH(beam_row_topk_kernel<false>, dim3(B * S * beam, S), dim3(256), 0, st, lp, ldlp, fsm, mach, last_pred, S, V, beam,
                       per_node, end_index, scratch_val, scratch_idx, 0);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}
namespace {
int beam_first_impl(bool norm, const float* lp, int ldlp, const uint8_t* fsm, int B, int S, int V, int beam, int64_t* pred,
                    float* lp_out, hipStream_t st) {
  return ssc_beam_first_dense(norm, lp, ldlp, fsm, nullptr, B, S, V, beam, pred, lp_out, st);
}
int beam_step_impl(bool norm, const float* lp, int ldlp, const uint8_t* fsm, const int64_t* last_pred, const float* last_lp, int B,
                   int S, int V, int beam, int per_node, int end_index, int64_t* pred, float* lp_out, int64_t* backptr,
                   float* scratch_val, int64_t* scratch_idx, hipStream_t st) {
  if (!lp || (!fsm && S != 1) || !last_pred || !last_lp || !pred || !lp_out || !backptr || !scratch_val || !scratch_idx) return SSC_EINVAL;
  if (B <= 0 || S <= 0 || V <= 0 || beam <= 0 || per_node <= 0 || per_node > V || ldlp < V || end_index < 0 ||
      end_index >= V || beam > S * beam * per_node)
    return SSC_EINVAL;
  SSC_TRY(ssc_beam_rows_dense(norm, lp, ldlp, fsm, nullptr, last_pred, B, S, V, beam, per_node, end_index, scratch_val, scratch_idx, st));
  return ssc_beam_merge(scratch_val, scratch_idx, last_lp, B, S, beam, per_node, pred, lp_out, backptr, end_index, nullptr, 0, 0,
                        nullptr, st);
}
}  // namespace

extern "C" int ssc_beam_first(const float* log_probs, int ldlp, const uint8_t* fsm, int B, int S, int V, int beam,
                              int64_t* pred, float* lp_out, void* stream) {
  return beam_first_impl(false, log_probs, ldlp, fsm, B, S, V, beam, pred, lp_out, (hipStream_t)stream);
}
extern "C" int ssc_beam_first_logits(const float* logits, int ldlp, const uint8_t* fsm, int B, int S, int V, int beam,
                                     int64_t* pred, float* lp_out, void* stream) {
  return beam_first_impl(true, logits, ldlp, fsm, B, S, V, beam, pred, lp_out, (hipStream_t)stream);
}

extern "C" int ssc_beam_step(const float* log_probs, int ldlp, const uint8_t* fsm, const int64_t* last_pred,
                             const float* last_lp, int B, int S, int V, int beam, int per_node, int end_index,
                             int64_t* pred, float* lp_out, int64_t* backptr, float* scratch_val, int64_t* scratch_idx,
                             void* stream) {
  return beam_step_impl(false, log_probs, ldlp, fsm, last_pred, last_lp, B, S, V, beam, per_node, end_index, pred, lp_out,
                        backptr, scratch_val, scratch_idx, (hipStream_t)stream);
}
extern "C" int ssc_beam_step_logits(const float* logits, int ldlp, const uint8_t* fsm, const int64_t* last_pred,
                                    const float* last_lp, int B, int S, int V, int beam, int per_node, int end_index,
                                    int64_t* pred, float* lp_out, int64_t* backptr, float* scratch_val,
                                    int64_t* scratch_idx, void* stream) {
  return beam_step_impl(true, logits, ldlp, fsm, last_pred, last_lp, B, S, V, beam, per_node, end_index, pred, lp_out,
                        backptr, scratch_val, scratch_idx, (hipStream_t)stream);
}

extern "C" int ssc_gather_rows(const float* src, int ld, const int64_t* backptr, int B, int rows_per_batch, int Wd,
                               float* dst, void* stream) {
  if (!src || !backptr || !dst || B <= 0 || rows_per_batch <= 0 || Wd <= 0 || ld < Wd || src == dst) return SSC_EINVAL;
  SSC_LAUNCH(gather_rows_kernel, dim3(ssc_cdiv(Wd, 256), B * rows_per_batch), dim3(256), 0, (hipStream_t)stream, src,
                     ld, backptr, rows_per_batch, Wd, dst);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_beam_backtrace(const int64_t* preds, const int64_t* backptrs, int steps, int B, int SB, int64_t* out,
                                  void* stream) {
  if (!preds || !out || steps <= 0 || B <= 0 || SB <= 0 || (steps > 1 && !backptrs)) return SSC_EINVAL;
  SSC_LAUNCH(beam_backtrace_kernel, dim3(ssc_cdiv(B * SB, 64)), dim3(64), 0, (hipStream_t)stream, preds, backptrs,
                     steps, B, SB, out);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// Sequence-level runner: the T-step teacher-forced training forward of the Style-SeqCVAE captioner and
// its hand-derived BPTT, launched from C++ on one HIP stream (graph-capturable, no host sync, no
// allocation).  Reference: UpDownCaptioner.forward training branch
// (var_updown/var_updown/models/updown_captioner.py:228-323), _decode_step (:371-455),
// UpDownCell.forward (var_updown/var_updown/modules/updown_cell.py:86-231),
// BottomUpTopDownAttention (updown-baseline/updown/modules/attention.py:36-125); the backward is
// what autograd derives for them.
//
// Output-preserving restructurings:
//  * time-invariant gate terms hoisted: emb_t W_ih^att[:, :E] for all t as one (T*B) x E GEMM, avg W_ih^att[:, E:E+F]
//    once; pv = Wv v once;
//  * torch.cat inputs never built: each K-segment of a gate GEMM reads its own source buffer;
//  * vocabulary projection + CE over all T steps at once (M = T*B);
//  * every weight gradient is one GEMM with K = T*B after the time loop.
#include <cstdlib>
#include <initializer_list>

#include "ssc_common.h"
#include "ssc_debug.h"

namespace {

inline size_t r4(size_t x) { return (x + 3) & ~(size_t)3; }

struct Layout {
  int B, R, L, T;
  int V, E, H, A, F, Z, S, tied;
  int Ep, Hp, Ap, Fp, Zp, Vp, H4, XW;  // padded leading dims ; XW = ld of dx = [datt | dh1 | dhd' | (dc)]
  int D, Dp, SC, NX;   // D: width of the attention-pooled prior mean (kld_mode 2: SENTIMENT_VAE = 2, D = Z), else 0; SC: how many of its
                       // leading entries condition the language LSTMs (= S: all D with LATENT_EMBEDDING "glove", 1 with "senti_word_net",
                       // updown_cell.py:169-172); NX = columns of dx
  size_t wc_e, wc_d, pool, dpm;   // D > 0: 16-byte aligned zero-padded copies of the c-blocks of W_ih^enc / W_ih^dec (H4 x Dp), the pooled c of every step (T*B x Dp), the KL term's gradient on the prior mean (B x Dp)
  size_t total = 0;                    // floats
  size_t act;   // int32: [0] = number of (t, b) rows with a real target (w = 1), [4 ...] = their row numbers t*B+b, ascending
  size_t live;  // int32: same layout; rows (t, b) with w = 1 at step t OR ANY LATER step of caption b (see build_active_rows_kernel)
  size_t tok, w, nvalid, sent_all, wcol_e, wcol_d, mask, avg, pv, emb, ga_static, ga_avg;
  size_t h1, c1, he, ce, hd, cd, gates_a, gates_e, gates_d, q, attn_logits, alpha, att, mu, lv, z, mulv;
  size_t slabs, slab_floats, logits, lse, proj;
  size_t sl_q, sl_mulv, sl_gh1, sl_ghd, sl_ghd2, sl_ghe, sl_dqw, sl_dhe, sl_dz, small_floats, wsum_att, wsum_dec, wz;
  size_t sl_ga, sl_ge, sl_gd, gate_floats;   // split-K slab regions of the three gate products (forward)
  // backward
  size_t dhdv, dga, dge, dgd, dga_sum, g_h1, g_c1, g_he, g_ce, g_cd, dz, dmulv, dx, dalpha, dq, dpv, dwa, demb, dproj;

  size_t take(size_t n) {
    size_t o = total;
    total += (n + 63) & ~(size_t)63;
    return o;
  }
};

Layout make_layout(const ssc_model_cfg* c, int B, int R, int L) {
  Layout l;
  l.B = B; l.R = R; l.L = L; l.T = L + 1;
  l.V = c->V; l.E = c->E; l.H = c->H; l.A = c->A; l.F = c->F; l.Z = c->Z; l.S = c->S; l.tied = c->tied;
  l.Ep = (int)r4(l.E); l.Hp = (int)r4(l.H); l.Ap = (int)r4(l.A); l.Fp = (int)r4(l.F); l.Zp = (int)r4(l.Z);
  l.Vp = (int)r4(l.V); l.H4 = 4 * l.H;
  l.D = c->kld_mode == 2 ? c->Z : 0; l.Dp = (int)r4(l.D); l.SC = l.D ? c->S : 0;
  // with a pooled conditioning block the input-gradient products of the language LSTMs run over the c-block columns too (they
  // follow the hd' block in W_ih^enc / W_ih^dec), rounded up to 16-byte rows: the row padding of W_ih^enc / columns of the
  // z-block of W_ih^dec, whose products land in pad columns of dx that nobody reads
  l.NX = l.D ? (int)r4(l.F + 2 * l.H + l.SC) : l.F + 2 * l.H;
  l.XW = (int)r4(l.NX);
  const size_t T = l.T, TB = T * B, T1B = (T + 1) * (size_t)B;
  l.tok = l.take(2 * (size_t)(L + 2) * B);
  l.w = l.take(TB);
  l.nvalid = l.take(B);
  l.act = l.take(TB + 4);
  l.live = l.take(TB + 4);
  l.sent_all = l.take(TB);
  l.wcol_e = l.take(l.H4); l.wcol_d = l.take(l.H4);
  l.mask = l.take((size_t)B * R);
  l.avg = l.take((size_t)B * l.Fp);
  l.pv = l.take((size_t)B * R * l.A);
  l.emb = l.take(TB * l.Ep);
  l.ga_static = l.take(TB * l.H4);
  l.ga_avg = l.take((size_t)B * l.H4);
  l.h1 = l.take(T1B * l.Hp); l.c1 = l.take(T1B * l.Hp);
  l.he = l.take(T1B * l.Hp); l.ce = l.take(T1B * l.Hp);
  l.hd = l.take(T1B * l.Hp); l.cd = l.take(T1B * l.Hp);
  l.gates_a = l.take(TB * l.H4); l.gates_e = l.take(TB * l.H4); l.gates_d = l.take(TB * l.H4);
  l.q = l.take(TB * l.Ap);
  l.attn_logits = l.take((size_t)B * R);
  l.alpha = l.take(TB * R);
  l.att = l.take(TB * l.Fp);
  l.mu = l.take(TB * l.Zp); l.lv = l.take(TB * l.Zp); l.z = l.take(TB * l.Zp);
  l.mulv = l.take((size_t)B * 2 * l.Z);
  // slab workspace: skinny GEMMs use up to 32 splits of (B x 4H); full GEMMs never exceed ~1024 tiles * 4096
  size_t skinny = (size_t)33 * B * l.H4;
  size_t full = (size_t)16 * 1024 * 1024;  // 64 MB: split-K slabs of the large GEMMs
  l.slab_floats = skinny > full ? skinny : full;
  l.slabs = l.take(l.slab_floats);
  // per-consumer slab regions of the small per-step GEMMs whose split-K reduction is fused into their consumer
  {
    size_t w = (size_t)(l.Hp > l.Ap ? l.Hp : l.Ap);
    if ((size_t)2 * l.Z > w) w = (size_t)2 * l.Z;
    l.small_floats = (size_t)40 * B * w;
    l.sl_q = l.take(l.small_floats); l.sl_mulv = l.take(l.small_floats);
    l.sl_gh1 = l.take(l.small_floats); l.sl_ghd = l.take(l.small_floats); l.sl_ghe = l.take(l.small_floats);
    l.sl_ghd2 = l.take(l.small_floats);
    l.sl_dqw = l.take(l.small_floats); l.sl_dhe = l.take(l.small_floats); l.sl_dz = l.take(l.small_floats);
  }
  // forward gate products: a product's recurrent part (issued at the start of the step) and its attention-dependent part
  // (issued after the attention) leave their slabs back to back in one region, the cell kernel sums them as one list
  l.gate_floats = (size_t)34 * B * l.H4;
  l.sl_ga = l.take(l.gate_floats); l.sl_ge = l.take(l.gate_floats); l.sl_gd = l.take(l.gate_floats);
  l.wsum_att = l.take((size_t)l.H4 * l.Hp);   // W_ih^att[:, h1-block] + W_hh^att  (both multiply h1')
  l.wsum_dec = l.take((size_t)l.H4 * l.Hp);   // W_ih^dec[:, hd-block] + W_hh^dec  (both multiply hd')
  l.wz = l.take((size_t)l.H4 * l.Zp);         // 16-B aligned copy of the z-block of W_ih^dec
  l.wc_e = l.take((size_t)l.H4 * l.Dp); l.wc_d = l.take((size_t)l.H4 * l.Dp);
  l.pool = l.take(TB * l.Dp); l.dpm = l.take((size_t)B * l.Dp);
  l.logits = l.take(TB * l.Vp);
  l.lse = l.take(2 * TB);
  l.proj = l.take(l.tied ? TB * l.Ep : 0);
  // backward
  l.dhdv = l.take(TB * l.Hp);
  l.dga = l.take(TB * l.H4); l.dge = l.take(TB * l.H4); l.dgd = l.take(TB * l.H4);
  l.dga_sum = l.take((size_t)B * l.H4);
  l.g_h1 = l.take((size_t)B * l.Hp); l.g_c1 = l.take((size_t)B * l.Hp);
  l.g_he = l.take((size_t)B * l.Hp); l.g_ce = l.take((size_t)B * l.Hp);
  l.g_cd = l.take((size_t)B * l.Hp);
  l.dz = l.take((size_t)B * l.Zp);
  l.dmulv = l.take(TB * 2 * l.Z);
  l.dx = l.take((size_t)B * l.XW);
  l.dalpha = l.take((size_t)B * R);
  l.dq = l.take(TB * l.Ap);
  l.dpv = l.take((size_t)B * R * l.A);
  l.dwa = l.take((size_t)B * l.A);
  l.demb = l.take(TB * l.Ep);
  l.dproj = l.take(l.tied ? TB * l.Ep : 0);
  return l;
}

struct Seg {
  const float* A; int lda;
  const float* B; int ldb;
  int K;
};

struct Ctx {
  hipStream_t st;
  float* slabs;
  size_t slab_floats;
  const int* act_count = nullptr;  // device: number of (t, b) rows with a non-pad target ...
  const int* act_rows = nullptr;   // ... and their row numbers (build_active_rows_kernel); nullptr = no compaction
  const int* live_count = nullptr; // device: number of (t, b) rows that can carry a non-zero gate gradient ...
  const int* live_rows = nullptr;  // ... and their row numbers
};

void fill_desc(ssc_gemm_desc& d, bool a_kc, bool b_kc, std::initializer_list<Seg> segs, int M, int N) {
  d = ssc_gemm_desc{};
  int i = 0;
  for (const Seg& s : segs) {
    if (s.K <= 0) continue;
    d.seg[i].A = s.A; d.seg[i].lda = s.lda; d.seg[i].B = s.B; d.seg[i].ldb = s.ldb; d.seg[i].K = s.K;
    ++i;
  }
  d.nseg = i;
  d.M = M; d.N = N;
  d.a_kc = a_kc; d.b_kc = b_kc;
}

// full GEMM into C (split-K through the shared slab workspace when it helps)
int gemm(const Ctx& c, bool a_kc, bool b_kc, std::initializer_list<Seg> segs, int M, int N, float* C, int ldc,
         const float* bias = nullptr, int accumulate = 0) {
  ssc_gemm_desc d;
  fill_desc(d, a_kc, b_kc, segs, M, N);
  d.C = C; d.ldc = ldc; d.bias = bias; d.accumulate = accumulate;
  d.splits = 0;
  d.workspace = c.slabs; d.workspace_floats = c.slab_floats;
  return ssc_gemm(&d, c.st);
}

// Products over the (t, b) rows of the caption batch skip rows on the device (ssc_gemm_desc row compaction).  Two lists:
//   act  - rows whose target is a real token (loss weight w = 1): the only rows whose logits are ever used, and the only
//          rows with a non-zero dlogits -> vocabulary head forward, its backward dHDv;
//   live - rows (t, b) with w = 1 at step t or at ANY LATER step of caption b: every other row lies in the padding
//          suffix, receives no gradient from the loss, the KL term or a later step, and so has dG = 0 exactly -> all
//          weight-gradient products and the embedding-gradient product.  A row with w = 0 INSIDE a caption (an in-caption
//          @@UNKNOWN@@, id 0 = the padding id: updown_captioner.py:265-278) is live: BPTT carries the
//          gradient of the later steps through it (the LSTM backward is not masked by w).
// When the operands do not qualify (alignment, fp32-MFMA mode) the product runs over all rows as before.
//   gemm_rows: C[r] = A[r] . B for the listed rows r only (other rows of C are left as they are)
//   gemm_dw:   C = A^T B summed over the live rows only
int gemm_rows(const Ctx& c, bool b_kc, std::initializer_list<Seg> segs, int M, int N, float* C, int ldc, const float* bias = nullptr,
              bool live = false) {
  ssc_gemm_desc d;
  fill_desc(d, true, b_kc, segs, M, N);
  d.C = C; d.ldc = ldc; d.bias = bias; d.accumulate = 0;
  d.splits = 0;
  d.workspace = c.slabs; d.workspace_floats = c.slab_floats;
  if (c.act_rows) {
    const int* cnt = live ? c.live_count : c.act_count;
    const int* rows = live ? c.live_rows : c.act_rows;
    d.m_count = cnt; d.a_rows = rows; d.c_rows = rows;
    const int rc = ssc_gemm(&d, c.st);
    if (rc != SSC_EALIGN && rc != SSC_EINVAL) return rc;
    d.m_count = d.a_rows = d.c_rows = nullptr;
  }
  return ssc_gemm(&d, c.st);
}
// The weight-gradient products of a backward phase are independent of each other: they are queued and issued together
// (ssc_gemm_dw_group: grouped launches of the wave-specialised kernel, several rounds of workgroups each).
struct DwBatch {
  static constexpr int MAX = 16;
  ssc_gemm_desc d[MAX];
  int n = 0;
};
int queue_dw(const Ctx& c, DwBatch& q, const float* A, int lda, const float* Bm, int ldb, int K, int M, int N, float* C, int ldc) {
  if (q.n >= DwBatch::MAX) return SSC_EINVAL;
  ssc_gemm_desc& d = q.d[q.n++];
  fill_desc(d, false, false, {{A, lda, Bm, ldb, K}}, M, N);
  d.C = C; d.ldc = ldc;
  d.splits = 0;
  d.workspace = c.slabs; d.workspace_floats = c.slab_floats;
  if (c.live_rows) { d.k_count = c.live_count; d.ka_rows = c.live_rows; d.kb_rows = c.live_rows; }
  return SSC_OK;
}
int flush_dw(const Ctx& c, DwBatch& q) {
  if (!q.n) return SSC_OK;
  const ssc_gemm_desc* dp[DwBatch::MAX];
  for (int i = 0; i < q.n; ++i) dp[i] = &q.d[i];
  // row compaction needs the 3xBF16 kernels and 16-B operands: a product that does not qualify runs over all rows
  for (int i = 0; i < q.n; ++i) {
    ssc_gemm_desc& d = q.d[i];
    const bool al = ssc_aligned16(d.seg[0].A) && ssc_aligned16(d.seg[0].B) && !(d.seg[0].lda & 3) && !(d.seg[0].ldb & 3) && !(d.M & 3) && !(d.N & 3);
    if (!al) d.k_count = d.ka_rows = d.kb_rows = nullptr;
  }
  int grouped = 1;
  (void)ssc_debug_get("dw_group", &grouped);   // include/ssc_debug.h: 0 = one 4-wave launch per product (A/B switch)
  if (!grouped) {
    int rc1 = SSC_OK;
    for (int i = 0; i < q.n && rc1 == SSC_OK; ++i) {
      rc1 = ssc_gemm(&q.d[i], c.st);
      if (rc1 == SSC_EINVAL || rc1 == SSC_EALIGN) {
        q.d[i].k_count = q.d[i].ka_rows = q.d[i].kb_rows = nullptr;
        rc1 = ssc_gemm(&q.d[i], c.st);
      }
    }
    q.n = 0;
    return rc1;
  }
  int rc = ssc_gemm_dw_group(dp, q.n, c.st);
  if (rc == SSC_EINVAL || rc == SSC_EALIGN) {  // e.g. the exact-fp32 mode refuses compaction: whole sums instead
    for (int i = 0; i < q.n; ++i) q.d[i].k_count = q.d[i].ka_rows = q.d[i].kb_rows = nullptr;
    rc = ssc_gemm_dw_group(dp, q.n, c.st);
  }
  q.n = 0;
  return rc;
}
int gemm_dw(const Ctx& c, const float* A, int lda, const float* Bm, int ldb, int K, int M, int N, float* C, int ldc) {
  DwBatch q;
  SSC_TRY(queue_dw(c, q, A, lda, Bm, ldb, K, M, N, C, ldc));
  return flush_dw(c, q);
}

// Workgroup 0: ascending list of the rows t*B+b whose target token is not padding (w = 1).  Workgroup 1: ascending list of
// the rows that are not in the padding SUFFIX of their caption (w = 1 at step t or at a later step of caption b).
// One workgroup per list, ordered block scan.
__global__ __launch_bounds__(1024) void build_active_rows_kernel(const float* __restrict__ w, int n, int B, int* __restrict__ act,
                                                                 int* __restrict__ live) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const bool suffix = blockIdx.x == 1;
  int* out = suffix ? live : act;
  auto flag = [&](int i) -> bool {
    if (!suffix) return w[i] != 0.f;
    for (int r = i; r < n; r += B)   // same caption, this step and every later one
      if (w[r] != 0.f) return true;
    return false;
  };
  const int per = (n + 1023) / 1024;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += flag(i);
  part[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - cnt;
  for (int i = lo; i < hi; ++i)
    if (flag(i)) out[4 + pos++] = i;
  if (tid == 1023) out[0] = part[1023];
  // entries past the count are never read by a product; keep them in range anyway
  for (int i = part[1023] + tid; i < n; i += 1024) out[4 + i] = 0;
}

// GEMM that leaves its split-K slabs (M x N, ld N) in `region` for a fused epilogue / consumer
int gemm_to_slabs(const Ctx& c, float* region, size_t cap, bool a_kc, bool b_kc, std::initializer_list<Seg> segs, int M,
                  int N, int* nslab) {
  ssc_gemm_desc d;
  fill_desc(d, a_kc, b_kc, segs, M, N);
  return ssc_gemm_slabs_auto(&d, region, cap, nslab, c.st);
}
int gemm_to_slabs(const Ctx& c, std::initializer_list<Seg> segs, int M, int N, int* nslab) {
  return gemm_to_slabs(c, c.slabs, c.slab_floats, true, true, segs, M, N, nslab);
}

__global__ void add2d_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int cols,
                             float* __restrict__ o, int ldo) {
  int r = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < cols) o[(size_t)r * ldo + x] = a[(size_t)r * lda + x] + b[(size_t)r * ldb + x];
}
int add2d(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* o, int ldo, hipStream_t st) {
  SSC_LAUNCH(add2d_kernel, dim3(ssc_cdiv(cols, 256), rows), dim3(256), 0, st, a, lda, b, ldb, cols, o, ldo);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}
// once per call, ONE launch: the pre-summed recurrent blocks W_ih[:, h-block] + W_hh of the attention / decoder LSTM (SURVEY
// A.5), the 16-byte aligned copy of the z-block of W_ih^dec, and the rank-1 sentiment columns of W_ih^enc / W_ih^dec made
// contiguous.  One workgroup per gate row n.
struct ViewArgs {
  const float *att_ih, *att_hh, *dec_ih, *dec_hh, *dec_z, *enc_s, *dec_s;   // row-n bases: [n * ld + column offset]
  int ld_att_ih, ld_att_hh, ld_dec_ih, ld_dec_hh, ld_enc_ih;
  int H, Hp, Z, Zp, S, D, Dp, SC;
  float *wsum_att, *wsum_dec, *wz, *wcol_e, *wcol_d, *wc_e, *wc_d;
};
__global__ void weight_views_kernel(const ViewArgs a) {
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < a.H; k += blockDim.x) {
    a.wsum_att[(size_t)n * a.Hp + k] = a.att_ih[(size_t)n * a.ld_att_ih + k] + a.att_hh[(size_t)n * a.ld_att_hh + k];
    a.wsum_dec[(size_t)n * a.Hp + k] = a.dec_ih[(size_t)n * a.ld_dec_ih + k] + a.dec_hh[(size_t)n * a.ld_dec_hh + k];
  }
  // (pad columns Z..Zp-1 are zero: the backward multiplies with all Zp columns so that the product keeps 16-byte rows when Z is
  // no multiple of 4 - the reference's shipped Z_SPACE is 150)
  for (int k = tid; k < a.Zp; k += blockDim.x) a.wz[(size_t)n * a.Zp + k] = k < a.Z ? a.dec_z[(size_t)n * a.ld_dec_ih + k] : 0.f;
  if (a.S == 1 && !a.D && tid == 0) {
    a.wcol_e[n] = a.enc_s[(size_t)n * a.ld_enc_ih];
    a.wcol_d[n] = a.dec_s[(size_t)n * a.ld_dec_ih];
  }
  for (int k = tid; k < a.Dp; k += blockDim.x) {   // SENTIMENT_VAE = 2: the 150 conditioning columns, zero-padded to 16-byte rows
    a.wc_e[(size_t)n * a.Dp + k] = k < a.SC ? a.enc_s[(size_t)n * a.ld_enc_ih + k] : 0.f;
    a.wc_d[(size_t)n * a.Dp + k] = k < a.SC ? a.dec_s[(size_t)n * a.ld_dec_ih + k] : 0.f;
  }
}
int prepare_weight_views(const Layout& l, const ssc_params* p, float* W, hipStream_t st) {
  const int E = l.E, F = l.F, H = l.H, S = l.S;
  ViewArgs a;
  a.att_ih = p->att_w_ih + E + F; a.att_hh = p->att_w_hh; a.dec_ih = p->dec_w_ih + F + H; a.dec_hh = p->dec_w_hh;
  a.dec_z = p->dec_w_ih + F + 2 * H + S; a.enc_s = p->enc_w_ih + F + 2 * H; a.dec_s = p->dec_w_ih + F + 2 * H;
  a.ld_att_ih = p->ld_att_w_ih; a.ld_att_hh = p->ld_att_w_hh; a.ld_dec_ih = p->ld_dec_w_ih; a.ld_dec_hh = p->ld_dec_w_hh;
  a.ld_enc_ih = p->ld_enc_w_ih;
  a.H = H; a.Hp = l.Hp; a.Z = l.Z; a.Zp = l.Zp; a.S = S; a.D = l.D; a.Dp = l.Dp; a.SC = l.SC;
  a.wc_e = W + l.wc_e; a.wc_d = W + l.wc_d;
  a.wsum_att = W + l.wsum_att; a.wsum_dec = W + l.wsum_dec; a.wz = W + l.wz; a.wcol_e = W + l.wcol_e; a.wcol_d = W + l.wcol_d;
  SSC_LAUNCH(weight_views_kernel, dim3(l.H4), dim3(256), 0, st, a);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// several buffers set to one value in ONE launch (the per-call zero fills: 16 launches of ~5 us each before)
struct FillList {
  static constexpr int MAX = 16;
  float* p[MAX];
  size_t n[MAX];
  int count = 0;
  void add(float* ptr, size_t len) { if (len && count < MAX) { p[count] = ptr; n[count] = len; ++count; } }
};
__global__ void fill_many_kernel(const FillList f, float v) {
  float* p = f.p[blockIdx.y];
  const size_t n = f.n[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int fill_many(const FillList& f, float v, hipStream_t st) {
  if (f.count == 0) return SSC_OK;
  SSC_LAUNCH(fill_many_kernel, dim3(256, f.count), dim3(256), 0, st, f, v);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

int check_cfg(const ssc_model_cfg* c, const ssc_params* p, const ssc_batch* b) {
  if (!c || !p || !b) return SSC_EINVAL;
  if (c->V <= 1 || c->E <= 0 || c->H <= 0 || c->A <= 0 || c->F <= 0 || c->Z <= 0) return SSC_EINVAL;
  if (c->kld_mode < 0 || c->kld_mode > 2) return SSC_EINVAL;
  // kld_mode 2 (SENTIMENT_VAE = 2): the conditioning block is the whole pooled attribute vector (S = Z) or its first entry (S = 1)
  if (c->kld_mode == 2 ? ((c->S != 1 && c->S != c->Z) || c->pm_scale != 0.f || !b->obj_atts) : (c->S != 0 && c->S != 1)) return SSC_EINVAL;
  if (b->B <= 0 || b->R <= 0 || b->L <= 0 || b->R > 256) return SSC_EINVAL;
  if (!b->feats || !b->caps || !b->eps) return SSC_EINVAL;
  if (((c->S == 1 && c->kld_mode != 2) || c->pm_scale != 0.f) && !b->sentiment) return SSC_EINVAL;
  if (!p->emb || !p->att_w_ih || !p->att_w_hh || !p->att_b_ih || !p->att_b_hh || !p->wq || !p->wv || !p->wa ||
      !p->enc_w_ih || !p->enc_w_hh || !p->enc_b_ih || !p->enc_b_hh || !p->dec_w_ih || !p->dec_w_hh || !p->dec_b_ih ||
      !p->dec_b_hh || !p->fc_mean_w || !p->fc_mean_b || !p->fc_lv_w || !p->fc_lv_b)
    return SSC_EINVAL;
  if (c->tied ? (!p->proj_w || !p->proj_b) : (!p->out_w || !p->out_b)) return SSC_EINVAL;
  return SSC_OK;
}

// include/ssc_debug.h: hipEvent pair around the time loop of the last forward / backward call
struct LoopProf {
  hipEvent_t e[4];
  bool created = false, on = false, fwd = false, bwd = false;
} g_loop;

}  // namespace

extern "C" int ssc_prof_loop_enable(int on) {
  if (on && !g_loop.created) {
    for (hipEvent_t& e : g_loop.e)
      if (hipEventCreate(&e) != hipSuccess) return SSC_EHIP;
    g_loop.created = true;
  }
  g_loop.on = on != 0;
  g_loop.fwd = g_loop.bwd = false;
  return SSC_OK;
}
extern "C" int ssc_prof_loop_ms(float* fwd_loop_ms, float* bwd_loop_ms) {
  if (!fwd_loop_ms || !bwd_loop_ms) return SSC_EINVAL;
  *fwd_loop_ms = *bwd_loop_ms = -1.f;
  if (g_loop.fwd) {
    if (hipEventSynchronize(g_loop.e[1]) != hipSuccess || hipEventElapsedTime(fwd_loop_ms, g_loop.e[0], g_loop.e[1]) != hipSuccess) return SSC_EHIP;
  }
  if (g_loop.bwd) {
    if (hipEventSynchronize(g_loop.e[3]) != hipSuccess || hipEventElapsedTime(bwd_loop_ms, g_loop.e[2], g_loop.e[3]) != hipSuccess) return SSC_EHIP;
  }
  return SSC_OK;
}

extern "C" size_t ssc_train_workspace_bytes(const ssc_model_cfg* cfg, int B, int R, int L) {
  if (!cfg || B <= 0 || R <= 0 || L <= 0) return 0;
  return make_layout(cfg, B, R, L).total * sizeof(float);
}

extern "C" void* ssc_train_workspace_view(const ssc_model_cfg* cfg, int B, int R, int L, void* workspace, int which,
                                          int* ld) {
  if (!cfg || !workspace) return nullptr;
  Layout l = make_layout(cfg, B, R, L);
  float* w = (float*)workspace;
  int dummy;
  if (!ld) ld = &dummy;
  switch (which) {
    case 0: *ld = l.Hp; return w + l.h1;
    case 1: *ld = l.Hp; return w + l.c1;
    case 2: *ld = l.Hp; return w + l.he;
    case 3: *ld = l.Hp; return w + l.ce;
    case 4: *ld = l.Hp; return w + l.hd;
    case 5: *ld = l.Hp; return w + l.cd;
    case 6: *ld = l.R; return w + l.alpha;
    case 7: *ld = l.Zp; return w + l.mu;
    case 8: *ld = l.Zp; return w + l.lv;
    case 9: *ld = l.Vp; return w + l.logits;
    case 10: *ld = l.B; return w + l.tok;
    case 11: *ld = l.Fp; return w + l.att;
    default: return nullptr;
  }
}

extern "C" int ssc_train_fwd(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* bt, void* workspace,
                             size_t workspace_bytes, float* loss, float* kld, void* stream) {
  SscGemmModeScope mode_scope(cfg);   // the numerics mode of this cfg, for every product the call issues
  SSC_TRY(check_cfg(cfg, p, bt));
  if (!workspace || !loss || !kld) return SSC_EINVAL;
  if (!ssc_aligned16(workspace)) return SSC_EALIGN;
  const Layout l = make_layout(cfg, bt->B, bt->R, bt->L);
  if (workspace_bytes < l.total * sizeof(float)) return SSC_EWORKSPACE;
  float* W = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  Ctx c{st, W + l.slabs, l.slab_floats};
  const int B = l.B, R = l.R, T = l.T, E = l.E, H = l.H, A = l.A, F = l.F, Z = l.Z, S = l.S, V = l.V, H4 = l.H4;
  const int TB = T * B;
  int64_t* tok = (int64_t*)(W + l.tok);
  const size_t sH = (size_t)B * l.Hp;  // per-step stride of the state histories

  // ---- per-sequence precompute --------------------------------------------------------------
  SSC_TRY(ssc_prep_tokens(bt->caps, B, l.L, cfg->pad, cfg->boundary, tok, W + l.w, W + l.nvalid, st));
  int* act = (int*)(W + l.act);
  int* live = (int*)(W + l.live);
  SSC_LAUNCH(build_active_rows_kernel, dim3(2), dim3(1024), 0, st, W + l.w, TB, B, act, live);
  SSC_CHECK_LAUNCH();
  c.act_count = act; c.act_rows = act + 4;
  c.live_count = live; c.live_rows = live + 4;
  SSC_TRY(ssc_feat_prep(bt->feats, B, R, F, W + l.mask, W + l.avg, st));
  SSC_TRY(gemm(c, true, true, {{bt->feats, F, p->wv, p->ld_wv, F}}, B * R, A, W + l.pv, A));
  SSC_TRY(ssc_embed_gather(p->emb, p->ld_emb, tok, TB, E, W + l.emb, l.Ep, st));
  SSC_TRY(gemm(c, true, true, {{W + l.emb, l.Ep, p->att_w_ih, p->ld_att_w_ih, E}}, TB, H4, W + l.ga_static, H4));
  SSC_TRY(gemm(c, true, true, {{W + l.avg, F, p->att_w_ih + E, p->ld_att_w_ih, F}}, B, H4, W + l.ga_avg, H4));
  // initial states (index 0 of every history), the KL accumulator, pad columns: one launch
  {
    FillList f;
    for (size_t off : {l.h1, l.c1, l.he, l.ce, l.hd, l.cd}) f.add(W + off, sH);
    f.add(kld, (size_t)B);
    if (l.Zp != Z) {  // keep pad columns of z finite (they are never read as K, but are memcpy'd in tests)
      f.add(W + l.z, (size_t)TB * l.Zp); f.add(W + l.mu, (size_t)TB * l.Zp); f.add(W + l.lv, (size_t)TB * l.Zp);
    }
    if (cfg->tied) f.add(W + l.proj, (size_t)TB * l.Ep);  // padded rows stay finite through the tanh
    SSC_TRY(fill_many(f, 0.f, st));
  }
  // (the rank-1 sentiment columns of W_ih^enc / W_ih^dec are made contiguous by prepare_weight_views)
  SSC_TRY(prepare_weight_views(l, p, W, st));
  const bool fc_adjacent = (p->fc_lv_w == p->fc_mean_w + (size_t)Z * p->ld_fc_mean_w) && p->ld_fc_lv_w == p->ld_fc_mean_w;

  // ---- time loop ----------------------------------------------------------------------------------
  if (g_loop.on) { (void)hipEventRecord(g_loop.e[0], st); }
  for (int t = 0; t < T; ++t) {
    float* h1p = W + l.h1 + t * sH; float* h1n = h1p + sH;
    float* c1p = W + l.c1 + t * sH; float* c1n = c1p + sH;
    float* hep = W + l.he + t * sH; float* hen = hep + sH;
    float* cep = W + l.ce + t * sH; float* cen = cep + sH;
    float* hdp = W + l.hd + t * sH; float* hdn = hdp + sH;
    float* cdp = W + l.cd + t * sH; float* cdn = cdp + sH;
    float* att = W + l.att + (size_t)t * B * l.Fp;
    float* qt = W + l.q + (size_t)t * B * l.Ap;
    float* zt = W + l.z + (size_t)t * B * l.Zp;
    float* poolt = l.D ? W + l.pool + (size_t)t * B * l.Dp : nullptr;   // c_t = sum_r alpha_tr obj_atts_r (SENTIMENT_VAE = 2)
    int ns = 0;
    const size_t sG = (size_t)B * H4;   // one gate slab
    int n_ga = 0, n_ge_r = 0, n_gd_r = 0, n_ge_a = 0, n_gd_a = 0;

    // (0) everything that only needs the states of step t-1, in ONE launch: the attention LSTM's gate product and the
    // recurrent K-segments of the encoder / decoder products (hd', he').  All recurrent inputs are zero at t = 0.
    if (t > 0) {
      const float* wr = p->att_w_ih + E + F;
      ssc_gemm_desc d3[3];
      // attention LSTM: x_a = [emb, avg, h1', hd'] (updown_cell.py:143-148); h1' meets the pre-summed W_ih[:,h1]+W_hh block
      fill_desc(d3[0], true, true, {{h1p, l.Hp, W + l.wsum_att, l.Hp, H}, {hdp, l.Hp, wr + H, p->ld_att_w_ih, H}}, B, H4);
      fill_desc(d3[1], true, true, {{hdp, l.Hp, p->enc_w_ih + F + H, p->ld_enc_w_ih, H}, {hep, l.Hp, p->enc_w_hh, p->ld_enc_w_hh, H}}, B, H4);
      fill_desc(d3[2], true, true, {{hdp, l.Hp, W + l.wsum_dec, l.Hp, H}}, B, H4);   // hd' meets W_ih^dec[:,hd] + W_hh^dec
      const ssc_gemm_desc* dp[3] = {&d3[0], &d3[1], &d3[2]};
      float* regions[3] = {W + l.sl_ga, W + l.sl_ge, W + l.sl_gd};
      const size_t caps[3] = {l.gate_floats, l.gate_floats / 2, l.gate_floats / 2};
      int ns3[3] = {0, 0, 0};
      SSC_TRY(ssc_gemm_slabs_group(dp, 3, regions, caps, ns3, st));
      n_ga = ns3[0]; n_ge_r = ns3[1]; n_gd_r = ns3[2];
    }
    // (i) attention LSTM cell
    {
      ssc_lstm_fwd_desc d{};
      d.B = B; d.H = H;
      if (n_ga > 0) { d.slabs = W + l.sl_ga; d.nslab = n_ga; d.slab_stride = sG; }
      d.add0 = W + l.ga_static + (size_t)t * B * H4; d.ld_add0 = H4;
      d.add1 = W + l.ga_avg; d.ld_add1 = H4; d.rows_per_add1 = 1;
      d.b_ih = p->att_b_ih; d.b_hh = p->att_b_hh;
      d.c_prev = c1p; d.ld_cprev = l.Hp;
      d.gates_out = W + l.gates_a + (size_t)t * B * H4;
      d.c_out = c1n; d.ld_cout = l.Hp; d.h_out = h1n; d.ld_hout = l.Hp;
      SSC_TRY(ssc_lstm_fwd(&d, st));
    }
    // (ii)+(iii) attention (attention.py:69-95, updown_cell.py:151-158); the q-projection's split-K slabs are summed
    // inside the logits kernel, which also stores q for the backward pass
    SSC_TRY(gemm_to_slabs(c, W + l.sl_q, l.small_floats, true, true, {{h1n, l.Hp, p->wq, p->ld_wq, H}}, B, A, &ns));
    SSC_TRY(ssc_attn_fwd_qslabs(W + l.sl_q, ns, (size_t)B * A, qt, l.Ap, W + l.pv, p->wa, W + l.mask, bt->feats, B, R, A, F, 1,
                                W + l.attn_logits, W + l.alpha + (size_t)t * B * R, att, l.Fp, st,
                                l.D ? bt->obj_atts : nullptr, l.D, poolt, l.Dp));
    // (iv) the attention-dependent K-segments [att, h1] of the encoder AND decoder products in one launch; their slabs follow
    // the recurrent ones.  x_e = [att, h1, hd', (s)] + he' (updown_cell.py:176-194); x_d = [att, h1, hd', (s), z] (:211-229)
    {
      ssc_gemm_desc d2[2];
      // (SENTIMENT_VAE = 2: + the attention-pooled conditioning block c, K = Dp against the aligned copies; a segment with K = 0 is dropped)
      fill_desc(d2[0], true, true, {{att, l.Fp, p->enc_w_ih, p->ld_enc_w_ih, F}, {h1n, l.Hp, p->enc_w_ih + F, p->ld_enc_w_ih, H},
                                    {poolt, l.Dp, W + l.wc_e, l.Dp, l.Dp}}, B, H4);
      fill_desc(d2[1], true, true, {{att, l.Fp, p->dec_w_ih, p->ld_dec_w_ih, F}, {h1n, l.Hp, p->dec_w_ih + F, p->ld_dec_w_ih, H},
                                    {poolt, l.Dp, W + l.wc_d, l.Dp, l.Dp}}, B, H4);
      const ssc_gemm_desc* dp[2] = {&d2[0], &d2[1]};
      float* regions[2] = {W + l.sl_ge + (size_t)n_ge_r * sG, W + l.sl_gd + (size_t)n_gd_r * sG};
      const size_t caps[2] = {l.gate_floats - (size_t)n_ge_r * sG, l.gate_floats - (size_t)n_gd_r * sG};
      int ns2[2] = {0, 0};
      SSC_TRY(ssc_gemm_slabs_group(dp, 2, regions, caps, ns2, st));
      n_ge_a = ns2[0]; n_gd_a = ns2[1];
    }
    // encoder LSTM cell.  With fc_mean / fc_log_var adjacent in the flat store (one (2Z, H) operand) the cell kernel also leaves
    // the partial products h_e . [W_mu ; W_lv]^T of its 16-unit slices (ssc_lstm_fwd_p) and the latent head sums them: no
    // product launch between the cell and the latent head
    const bool fused_fc = fc_adjacent && 2 * Z <= 256 && (size_t)ssc_cdiv(H, 16) * B * 2 * Z <= l.small_floats;
    {
      ssc_lstm_fwd_desc d{};
      d.B = B; d.H = H;
      d.slabs = W + l.sl_ge; d.nslab = n_ge_r + n_ge_a; d.slab_stride = sG;
      d.b_ih = p->enc_b_ih; d.b_hh = p->enc_b_hh;
      if (S == 1 && !l.D) { d.sent = bt->sentiment; d.wcol = W + l.wcol_e; d.ldwcol = 1; }
      d.c_prev = cep; d.ld_cprev = l.Hp;
      d.gates_out = W + l.gates_e + (size_t)t * B * H4;
      d.c_out = cen; d.ld_cout = l.Hp; d.h_out = hen; d.ld_hout = l.Hp;
      if (fused_fc) SSC_TRY(ssc_lstm_fwd_p(&d, p->fc_mean_w, p->ld_fc_mean_w, 2 * Z, W + l.sl_mulv, st));
      else SSC_TRY(ssc_lstm_fwd(&d, st));
    }
    // latent head: mean / log_var / z / KL (updown_cell.py:196-208, updown_captioner.py:295-303)
    {
      float* mulv = W + l.mulv;
      ssc_latent_fwd_desc d{};
      d.B = B; d.Z = Z;
      if (fused_fc) {
        d.mulv = W + l.sl_mulv; d.ldmulv = 2 * Z; d.nslab = ssc_cdiv(H, 16); d.slab_stride = (size_t)B * 2 * Z;
      } else if (fc_adjacent) {  // one (2Z x H) operand; the slabs go straight to the latent epilogue
        SSC_TRY(gemm_to_slabs(c, W + l.sl_mulv, l.small_floats, true, true, {{hen, l.Hp, p->fc_mean_w, p->ld_fc_mean_w, H}}, B,
                              2 * Z, &ns));
        d.mulv = W + l.sl_mulv; d.ldmulv = 2 * Z; d.nslab = ns; d.slab_stride = (size_t)B * 2 * Z;
      } else {
        SSC_TRY(gemm(c, true, true, {{hen, l.Hp, p->fc_mean_w, p->ld_fc_mean_w, H}}, B, Z, mulv, 2 * Z));
        SSC_TRY(gemm(c, true, true, {{hen, l.Hp, p->fc_lv_w, p->ld_fc_lv_w, H}}, B, Z, mulv + Z, 2 * Z));
        d.mulv = mulv; d.ldmulv = 2 * Z; d.nslab = 1; d.slab_stride = 0;
      }
      d.bmu = p->fc_mean_b; d.blv = p->fc_lv_b;
      d.eps = bt->eps + (size_t)t * B * Z; d.ldeps = Z;
      d.kld_mode = cfg->kld_mode; d.sent = cfg->pm_scale != 0.f ? bt->sentiment : nullptr;
      d.pm_scale = cfg->pm_scale; d.prior_var = cfg->prior_var;
      if (l.D) { d.pm = poolt; d.ldpm = l.Dp; }   // the prior mean of this step is the pooled c (updown_cell.py:160-163)
      d.w = W + l.w + (size_t)t * B;
      d.mu = W + l.mu + (size_t)t * B * l.Zp; d.lv = W + l.lv + (size_t)t * B * l.Zp; d.z = zt; d.ldz = l.Zp;
      d.kld_acc = kld;
      SSC_TRY(ssc_latent_fwd(&d, st));
    }
    // (vi) decoder LSTM cell; its z block (K = Z, the only operand that waits for the latent head) is formed inside the kernel
    {
      ssc_lstm_fwd_desc d{};
      d.B = B; d.H = H;
      d.slabs = W + l.sl_gd; d.nslab = n_gd_r + n_gd_a; d.slab_stride = sG;
      d.b_ih = p->dec_b_ih; d.b_hh = p->dec_b_hh;
      if (S == 1 && !l.D) { d.sent = bt->sentiment; d.wcol = W + l.wcol_d; d.ldwcol = 1; }
      d.c_prev = cdp; d.ld_cprev = l.Hp;
      d.gates_out = W + l.gates_d + (size_t)t * B * H4;
      d.c_out = cdn; d.ld_cout = l.Hp; d.h_out = hdn; d.ld_hout = l.Hp;
      SSC_TRY(ssc_lstm_fwd_z(&d, zt, l.Zp, W + l.wz, l.Zp, Z, st));
    }
  }

  if (g_loop.on) { (void)hipEventRecord(g_loop.e[1], st); g_loop.fwd = true; }

  // ---- vocabulary projection + CE over all steps (updown_captioner.py:444-445, 457-466) -----------
  const float* hd_all = W + l.hd + sH;  // rows t*B+b = h_dec after step t
  // only the rows with a real target are projected (ssc_ce_fwd skips the others)
  if (cfg->tied) {
    // (l.proj was zeroed with the initial states: padded rows stay finite through the tanh)
    SSC_TRY(gemm_rows(c, true, {{hd_all, l.Hp, p->proj_w, p->ld_proj_w, H}}, TB, E, W + l.proj, l.Ep));
    SSC_TRY(ssc_bias_tanh(W + l.proj, l.Ep, TB, E, p->proj_b, st));
    SSC_TRY(gemm_rows(c, true, {{W + l.proj, l.Ep, p->emb, p->ld_emb, E}}, TB, V, W + l.logits, l.Vp));
  } else {
    SSC_TRY(gemm_rows(c, true, {{hd_all, l.Hp, p->out_w, p->ld_out_w, H}}, TB, V, W + l.logits, l.Vp, p->out_b));
  }
  SSC_TRY(ssc_ce_fwd(W + l.logits, l.Vp, tok + B, W + l.w, W + l.nvalid, T, B, V, W + l.lse, loss, st));
  return SSC_OK;
}

namespace {
// ---- vocabulary sizes that are no multiple of 4 ------------------------------------------------------------------------------
// The two backward products of the vocabulary head that run over V as a k- or m-extent (d(hidden) = dlogits . W over k = V;
// dW = dlogits^T . hidden with V rows) need 16-byte rows for the MFMA kernels; with V % 4 != 0 they fell to the scalar kernel over
// all rows (V = 10001: +7.5 % step time - and real vocabularies are rarely a multiple of 4).  The products now run on the first
// V4 = V & ~3 entries and these two kernels add the 1-3 entries left (dlogits is defined - zero - on every row without a target).
__global__ __launch_bounds__(256) void head_tail_rows_kernel(const float* __restrict__ dlog, int ldl, const float* __restrict__ Wt,
                                                             int ldw, int v0, int nt, int N, float* __restrict__ out, int ldo) {
  // out[r, n] += sum_{j < nt} dlog[r, v0 + j] * Wt[v0 + j, n]
  const int r = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int j = 0; j < nt; ++j) acc += dlog[(size_t)r * ldl + v0 + j] * Wt[(size_t)(v0 + j) * ldw + n];
  out[(size_t)r * ldo + n] += acc;
}
__global__ __launch_bounds__(1024) void head_tail_dw_kernel(const float* __restrict__ dlog, int ldl, const float* __restrict__ X, int ldx,
                                                            int rows, int v0, int N, float* __restrict__ dW, int ldw) {
  // dW[v0 + j, n] = sum_r dlog[r, v0 + j] * X[r, n]   (j = blockIdx.y); 64 columns x 16 row lanes per workgroup, four rows in
  // flight per thread, lanes combined in a fixed order
  __shared__ float part[16][64];
  const int j = blockIdx.y, nl = threadIdx.x & 63, rl = threadIdx.x >> 6, n = min(blockIdx.x * 64 + nl, N - 1);
  float acc = 0.f;
  for (int r0 = rl; r0 < rows; r0 += 64) {
    float d4[4], x4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = min(r0 + 16 * u, rows - 1);
      d4[u] = dlog[(size_t)r * ldl + v0 + j];
      x4[u] = X[(size_t)r * ldx + n];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (r0 + 16 * u < rows) ? d4[u] * x4[u] : 0.f;
  }
  part[rl][nl] = acc;
  __syncthreads();
  if (rl == 0 && blockIdx.x * 64 + nl < N) {
    float t = part[0][nl];
    for (int i = 1; i < 16; ++i) t += part[i][nl];
    dW[(size_t)(v0 + j) * ldw + n] = t;
  }
}
int head_tail_rows(const float* dlog, int ldl, const float* Wt, int ldw, int v0, int V, int rows, int N, float* out, int ldo, hipStream_t st) {
  const int nt = V - v0;
  if (nt <= 0) return SSC_OK;
  SSC_LAUNCH(head_tail_rows_kernel, dim3(ssc_cdiv(N, 256), rows), dim3(256), 0, st, dlog, ldl, Wt, ldw, v0, nt, N, out, ldo);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}
int head_tail_dw(const float* dlog, int ldl, const float* X, int ldx, int v0, int V, int rows, int N, float* dW, int ldw, hipStream_t st) {
  const int nt = V - v0;
  if (nt <= 0) return SSC_OK;
  SSC_LAUNCH(head_tail_dw_kernel, dim3(ssc_cdiv(N, 64), nt), dim3(1024), 0, st, dlog, ldl, X, ldx, rows, v0, N, dW, ldw);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}
__global__ void repeat_kernel(const float* __restrict__ src, int n, int reps, float* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * reps) dst[i] = src[i % n];
}
}  // namespace

// phases (bit mask): 1 = vocabulary head + BPTT time loop, or its halves 16 = vocabulary head (output-head gradients are final
// after it) and 32 = BPTT time loop; 2 = embedding, attention-LSTM and attention-projection gradients, 4 = encoder-LSTM and
// latent-head gradients, 8 = decoder-LSTM gradients; 2 = 64 (the embedding gradient alone) + 128 (attention-LSTM and attention
// projections).  16 and 32 come first, in this order; 2 (or 64, 128), 4 and 8 are independent of each other (any order); one stream.  Splitting
// them lets the caller start the all-reduce of a finished gradient range while the next phase computes
// (ssc_runtime/engine.py): the output head's 48 MB travel under the whole time loop.
static int train_bwd_impl(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* bt, void* workspace,
                          size_t workspace_bytes, const float* gl, const float* gk, const ssc_params* g, void* stream,
                          unsigned phases) {
  SscGemmModeScope mode_scope(cfg);   // the numerics mode of this cfg, for every product the call issues
  SSC_TRY(check_cfg(cfg, p, bt));
  if (!workspace || !gl || !gk || !g) return SSC_EINVAL;
  const Layout l = make_layout(cfg, bt->B, bt->R, bt->L);
  if (workspace_bytes < l.total * sizeof(float)) return SSC_EWORKSPACE;
  float* W = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  Ctx c{st, W + l.slabs, l.slab_floats};
  const int B = l.B, R = l.R, T = l.T, E = l.E, H = l.H, A = l.A, F = l.F, Z = l.Z, S = l.S, V = l.V, H4 = l.H4;
  const int TB = T * B, XW = l.XW;
  int64_t* tok = (int64_t*)(W + l.tok);
  const size_t sH = (size_t)B * l.Hp;
  const float* hd_all = W + l.hd + sH;
  c.act_count = (const int*)(W + l.act); c.act_rows = c.act_count + 4;  // built by ssc_train_fwd of this minibatch
  c.live_count = (const int*)(W + l.live); c.live_rows = c.live_count + 4;

  if (phases & (1u | 16u)) {
  // ---- vocabulary head ------------------------------------------------------------------------------
  SSC_TRY(ssc_ce_bwd(W + l.logits, l.Vp, tok + B, W + l.w, W + l.nvalid, W + l.lse, gl, T, B, V, st));
  const float* dlog = W + l.logits;
  {  // every zero-initialised buffer of this phase in one launch
    FillList f;
    f.add(W + l.dhdv, (size_t)TB * l.Hp);   // the padded rows receive no gradient
    if (cfg->tied) f.add(W + l.dproj, (size_t)TB * l.Ep);
    // carried gradients start at zero.  g_c1 / g_ce / g_cd are plain (B,H) buffers.  The carried hidden-state gradients are NOT
    // reduced into buffers: the split-K slabs of the GEMMs that produce them (g_h1', g_hd', g_he', dq Wq, dmu Wmu + dlv Wlv) stay
    // in per-consumer slab regions and are summed, in a fixed order, inside the LSTM backward kernel that consumes them.
    for (size_t off : {l.g_c1, l.g_ce, l.g_cd}) f.add(W + off, sH);
    f.add(W + l.dx, (size_t)B * XW);
    f.add(W + l.dpv, (size_t)B * R * A);
    f.add(W + l.dwa, (size_t)B * A);
    f.add(W + l.dga_sum, (size_t)B * H4);
    SSC_TRY(fill_many(f, 0.f, st));
  }
  const int V4 = V >= 8 ? (V & ~3) : V;   // (see head_tail_rows_kernel)
  if (cfg->tied) {
    float* dP = W + l.dproj;
    SSC_TRY(gemm_rows(c, false, {{dlog, l.Vp, p->emb, p->ld_emb, V4}}, TB, E, dP, l.Ep));
    SSC_TRY(head_tail_rows(dlog, l.Vp, p->emb, p->ld_emb, V4, V, TB, E, dP, l.Ep, st));
    SSC_TRY(ssc_tanh_bwd(dP, l.Ep, W + l.proj, l.Ep, TB, E, st));
    SSC_TRY(gemm_rows(c, false, {{dP, l.Ep, p->proj_w, p->ld_proj_w, E}}, TB, H, W + l.dhdv, l.Hp));
    if (g->proj_w) SSC_TRY(gemm_dw(c, dP, l.Ep, hd_all, l.Hp, TB, E, H, g->proj_w, g->ld_proj_w));
    if (g->proj_b) SSC_TRY(ssc_colsum2(dP, l.Ep, TB, E, nullptr, g->proj_b, 1, nullptr, 0, c.slabs, st));
  } else {
    SSC_TRY(gemm_rows(c, false, {{dlog, l.Vp, p->out_w, p->ld_out_w, V4}}, TB, H, W + l.dhdv, l.Hp));
    SSC_TRY(head_tail_rows(dlog, l.Vp, p->out_w, p->ld_out_w, V4, V, TB, H, W + l.dhdv, l.Hp, st));
    if (g->out_w) {
      SSC_TRY(gemm_dw(c, dlog, l.Vp, hd_all, l.Hp, TB, V4, H, g->out_w, g->ld_out_w));
      SSC_TRY(head_tail_dw(dlog, l.Vp, hd_all, l.Hp, V4, V, TB, H, g->out_w, g->ld_out_w, st));
    }
    if (g->out_b) SSC_TRY(ssc_colsum2(dlog, l.Vp, TB, V, nullptr, g->out_b, 1, nullptr, 0, c.slabs, st));
  }
  }  // phase 16 (first half of phase 1)

  if (phases & (1u | 32u)) {
  // wsum_att / wz were prepared by ssc_train_fwd of the same minibatch (parameters are unchanged until the update)
  float* dx = W + l.dx;            // [datt (F) | dh1 (H) | dhd' (H)], ld XW
  int n_gh1 = 0, n_ghd = 0, n_ghd2 = 0, n_ghe = 0;  // slab counts carried from step t+1 (none at t = T-1)
  const bool fc_adjacent_b = (p->fc_lv_w == p->fc_mean_w + (size_t)Z * p->ld_fc_mean_w) && p->ld_fc_lv_w == p->ld_fc_mean_w;
  const size_t sBH = (size_t)B * H;
  const int NX = l.NX;             // F + 2H (+ the c-block, rounded up to 16-byte rows, with SENTIMENT_VAE = 2)
  const size_t sBX = (size_t)B * NX;
  // Launch order of one step (every product streams its weight block once; a product is issued as soon as its dG exists and
  // shares a launch with the latency-bound small product of the dependency chain that sits at the same place):
  //   dec cell -> {dz, dGd W_ih^dec[:, :F+2H], dGd W_hh^dec} -> latent -> dhe -> enc cell -> {dGe W_ih^enc[:, :F+2H]} -> sum
  //   (with dGe W_hh^enc) -> attention -> att cell (dq Wq formed inside) -> {dGa (W_ih^att[h1] + W_hh^att), dGa W_ih^att[hd]}

  if (g_loop.on) { (void)hipEventRecord(g_loop.e[2], st); }
  for (int t = T - 1; t >= 0; --t) {
    float* dgd = W + l.dgd + (size_t)t * B * H4;
    float* dge = W + l.dge + (size_t)t * B * H4;
    float* dga = W + l.dga + (size_t)t * B * H4;
    float* dmulv = W + l.dmulv + (size_t)t * B * 2 * Z;
    float* dq = W + l.dq + (size_t)t * B * l.Ap;
    int ns = 0, n_dxd = 0, n_dxe = 0;
    // 1. decoder LSTM: dh = dx[hd' block] (step t+1's dGd W_ih^dec[hd] + dGe W_ih^enc[hd]) + g_hd' slabs (step t+1's
    //    dGa W_ih^att[hd] and dGd W_hh^dec) + vocabulary path
    {
      ssc_lstm_bwd_desc d{};
      d.B = B; d.H = H;
      d.dh = dx + F + H; d.ld_dh = XW;
      d.dh2 = W + l.dhdv + (size_t)t * B * l.Hp; d.ld_dh2 = l.Hp;
      d.slabsA = W + l.sl_ghd; d.nA = n_ghd; d.strideA = sBH;
      d.slabsB = W + l.sl_ghd2; d.nB = n_ghd2; d.strideB = sBH;
      d.dc_in = W + l.g_cd; d.ld_dcin = l.Hp;
      d.gates = W + l.gates_d + (size_t)t * B * H4;
      d.c_prev = W + l.cd + t * sH; d.ld_cprev = l.Hp;
      d.c_new = W + l.cd + (t + 1) * sH; d.ld_cnew = l.Hp;
      d.dG = dgd; d.dc_prev = W + l.g_cd; d.ld_dcprev = l.Hp;
      SSC_TRY(ssc_lstm_bwd(&d, st));
    }
    // 2. everything dGd feeds, in one launch: dz = dGd W_ih^dec[:, z-block] (aligned copy; summed inside the latent backward),
    //    the decoder half of [datt | dh1 | dhd'] (slabs; the encoder half follows in 6), and dGd W_hh^dec for step t-1
    {
      ssc_gemm_desc d3[3];
      fill_desc(d3[0], true, false, {{dgd, H4, W + l.wz, l.Zp, H4}}, B, l.Zp);   // Zp columns (pad columns of wz are zero): 16-byte rows for any Z
      fill_desc(d3[1], true, false, {{dgd, H4, p->dec_w_ih, p->ld_dec_w_ih, H4}}, B, NX);
      fill_desc(d3[2], true, false, {{dgd, H4, p->dec_w_hh, p->ld_dec_w_hh, H4}}, B, H);
      const ssc_gemm_desc* dp[3] = {&d3[0], &d3[1], &d3[2]};
      float* regions[3] = {W + l.sl_dz, c.slabs, W + l.sl_ghd2};
      const size_t caps[3] = {l.small_floats, c.slab_floats / 2, l.small_floats};
      int ns3[3] = {0, 0, 0};
      SSC_TRY(ssc_gemm_slabs_group(dp, t > 0 ? 3 : 2, regions, caps, ns3, st));
      ns = ns3[0]; n_dxd = ns3[1]; n_ghd2 = ns3[2];
    }
    {
      ssc_latent_bwd_desc d{};
      d.B = B; d.Z = Z;
      d.dz = W + l.sl_dz; d.lddz = l.Zp; d.nslab = ns; d.slab_stride = (size_t)B * l.Zp;
      d.eps = bt->eps + (size_t)t * B * Z; d.ldeps = Z;
      d.mu = W + l.mu + (size_t)t * B * l.Zp; d.lv = W + l.lv + (size_t)t * B * l.Zp; d.ldz = l.Zp;
      d.kld_mode = cfg->kld_mode; d.sent = cfg->pm_scale != 0.f ? bt->sentiment : nullptr;
      d.pm_scale = cfg->pm_scale; d.prior_var = cfg->prior_var;
      if (l.D) { d.pm = W + l.pool + (size_t)t * B * l.Dp; d.ldpm = l.Dp; d.dpm = W + l.dpm; d.lddpm = l.Dp; }
      d.w = W + l.w + (size_t)t * B; d.gk = gk;
      d.dmulv = dmulv; d.lddmulv = 2 * Z;
      SSC_TRY(ssc_latent_bwd(&d, st));
    }
    // 4-5. encoder LSTM: dhe = g_he' slabs + (dmu | dlv) [W_mu ; W_lv].  With the two fc weights adjacent in the flat store the
    // K = 2Z product is formed inside the cell kernel (ssc_lstm_bwd_x): no launch of its own on the dependency chain
    {
      ssc_lstm_bwd_desc d{};
      d.B = B; d.H = H;
      d.slabsA = W + l.sl_ghe; d.nA = n_ghe; d.strideA = sBH;
      d.dc_in = W + l.g_ce; d.ld_dcin = l.Hp;
      d.gates = W + l.gates_e + (size_t)t * B * H4;
      d.c_prev = W + l.ce + t * sH; d.ld_cprev = l.Hp;
      d.c_new = W + l.ce + (t + 1) * sH; d.ld_cnew = l.Hp;
      d.dG = dge; d.dc_prev = W + l.g_ce; d.ld_dcprev = l.Hp;
      if (fc_adjacent_b && 2 * Z <= 768) {   // ssc_lstm_bwd_x holds a K <= 768 operand image in LDS; wider latents take the product below
        SSC_TRY(ssc_lstm_bwd_x(&d, dmulv, 2 * Z, p->fc_mean_w, p->ld_fc_mean_w, 2 * Z, st));
      } else {
        SSC_TRY(gemm_to_slabs(c, W + l.sl_dhe, l.small_floats, true, false,
                              {{dmulv, 2 * Z, p->fc_mean_w, p->ld_fc_mean_w, Z}, {dmulv + Z, 2 * Z, p->fc_lv_w, p->ld_fc_lv_w, Z}}, B, H,
                              &ns));
        d.slabsB = W + l.sl_dhe; d.nB = ns; d.strideB = sBH;
        SSC_TRY(ssc_lstm_bwd(&d, st));
      }
    }
    // 6. everything dGe feeds, in one launch: the encoder half of [datt | dh1 | dhd'] (appended to the decoder half's slabs of
    //    launch 2, one sum) and dGe W_hh^enc for step t-1
    {
      ssc_gemm_desc d2[2];
      fill_desc(d2[0], true, false, {{dge, H4, p->enc_w_ih, p->ld_enc_w_ih, H4}}, B, NX);
      fill_desc(d2[1], true, false, {{dge, H4, p->enc_w_hh, p->ld_enc_w_hh, H4}}, B, H);
      const ssc_gemm_desc* dp[2] = {&d2[0], &d2[1]};
      float* regions[2] = {c.slabs + (size_t)n_dxd * sBX, W + l.sl_ghe};
      const size_t caps[2] = {c.slab_floats - (size_t)n_dxd * sBX, l.small_floats};
      int ns2[2] = {0, 0};
      SSC_TRY(ssc_gemm_slabs_group(dp, t > 0 ? 2 : 1, regions, caps, ns2, st));
      n_dxe = ns2[0]; n_ghe = ns2[1];
    }
    SSC_TRY(ssc_reduce_slabs(c.slabs, n_dxd + n_dxe, sBX, B, NX, dx, XW, nullptr, 0, st));
    // 7. attention backward (SENTIMENT_VAE = 2: the weights also pooled obj_atts; dc = dx[c block] from both language LSTMs + the
    //    KL term's gradient on the prior mean)
    SSC_TRY(ssc_attn_bwd_pool(dx, XW, W + l.q + (size_t)t * B * l.Ap, l.Ap, W + l.pv, p->wa, W + l.alpha + (size_t)t * B * R,
                              bt->feats, B, R, A, F, dq, l.Ap, W + l.dpv, W + l.dwa, W + l.dalpha, l.D ? bt->obj_atts : nullptr, l.D,
                              dx + F + 2 * H, XW, l.SC, W + l.dpm, l.Dp, st));
    // 8-9. attention LSTM: dh1 = dx[h1 block] + g_h1' slabs + dq Wq; the K = A product is formed inside the cell kernel
    //      (ssc_lstm_bwd_x) while its LDS images fit, else it is a split-K product of its own
    {
      ssc_lstm_bwd_desc d{};
      d.B = B; d.H = H;
      d.dh = dx + F; d.ld_dh = XW;
      d.slabsA = W + l.sl_gh1; d.nA = n_gh1; d.strideA = sBH;
      d.dc_in = W + l.g_c1; d.ld_dcin = l.Hp;
      d.gates = W + l.gates_a + (size_t)t * B * H4;
      d.c_prev = W + l.c1 + t * sH; d.ld_cprev = l.Hp;
      d.c_new = W + l.c1 + (t + 1) * sH; d.ld_cnew = l.Hp;
      d.dG = dga; d.dc_prev = W + l.g_c1; d.ld_dcprev = l.Hp;
      d.dgsum = W + l.dga_sum;
      const bool fused_q = A <= 768;
      if (fused_q) {
        SSC_TRY(ssc_lstm_bwd_x(&d, dq, l.Ap, p->wq, p->ld_wq, A, st));
      } else {
        SSC_TRY(gemm_to_slabs(c, W + l.sl_dqw, l.small_floats, true, false, {{dq, l.Ap, p->wq, p->ld_wq, A}}, B, H, &ns));
        d.slabsB = W + l.sl_dqw; d.nB = ns; d.strideB = sBH;
        SSC_TRY(ssc_lstm_bwd(&d, st));
      }
    }
    // 10. what dGa carries to step t-1 (left as slabs for their consumers)
    if (t > 0) {
      const float* wr = p->att_w_ih + E + F;
      ssc_gemm_desc d2[2];
      fill_desc(d2[0], true, false, {{dga, H4, W + l.wsum_att, l.Hp, H4}}, B, H);
      fill_desc(d2[1], true, false, {{dga, H4, wr + H, p->ld_att_w_ih, H4}}, B, H);
      const ssc_gemm_desc* dp[2] = {&d2[0], &d2[1]};
      float* regions[2] = {W + l.sl_gh1, W + l.sl_ghd};
      const size_t caps[2] = {l.small_floats, l.small_floats};
      int ns2[2] = {0, 0};
      SSC_TRY(ssc_gemm_slabs_group(dp, 2, regions, caps, ns2, st));
      n_gh1 = ns2[0]; n_ghd = ns2[1];
    }
  }

  if (g_loop.on) { (void)hipEventRecord(g_loop.e[3], st); g_loop.bwd = true; }

  if (S == 1 && !l.D) {  // sentiment replicated over time (row = t*B+b) for the rank-1 column gradients
    SSC_LAUNCH(repeat_kernel, dim3(ssc_cdiv(TB, 256)), dim3(256), 0, st, bt->sentiment, B, T, W + l.sent_all);
    SSC_CHECK_LAUNCH();
  }
  }  // phase 32 (second half of phase 1)

  const int zcol = F + 2 * H + S;
  // ---- weight gradients: one K = T*B GEMM per block ----------------------------------------------------
  const float* dga = W + l.dga; const float* dge = W + l.dge; const float* dgd = W + l.dgd;
  const float* h1_prev = W + l.h1; const float* h1_new = W + l.h1 + sH;
  const float* hd_prev = W + l.hd; const float* he_prev = W + l.he; const float* he_new = W + l.he + sH;
  const float* att = W + l.att;
  if (phases & 64u) {   // phase 2a: the embedding gradient alone (its 40 MB range can travel while every other phase computes)
  if (g->emb && !cfg->tied) {
    SSC_TRY(ssc_fill(W + l.demb, (size_t)TB * l.Ep, 0.f, st));  // rows of the padding suffix add nothing to the embedding gradient
    SSC_TRY(gemm_rows(c, false, {{dga, H4, p->att_w_ih, p->ld_att_w_ih, H4}}, TB, E, W + l.demb, l.Ep, nullptr, /*live=*/true));
    // zero the table gradient, then scatter-add rows by token id (padding_idx row gets none)
    if (hipMemset2DAsync(g->emb, (size_t)g->ld_emb * sizeof(float), 0, (size_t)E * sizeof(float), V, st) != hipSuccess)
      return SSC_EHIP;
    SSC_TRY(ssc_embed_scatter_add(g->emb, g->ld_emb, tok, TB, E, W + l.demb, l.Ep, cfg->pad, st));
  }
  }  // phase 2a
  if (phases & 128u) {  // phase 2b: attention-LSTM and attention-projection gradients
  {  // all weight-gradient products of this phase at once (grouped launches)
    DwBatch q;
    if (g->att_w_ih) {
      float* gw = g->att_w_ih; int ld = g->ld_att_w_ih;
      SSC_TRY(queue_dw(c, q, dga, H4, W + l.emb, l.Ep, TB, H4, E, gw, ld));
      SSC_TRY(queue_dw(c, q, dga, H4, h1_prev, l.Hp, TB, H4, H, gw + E + F, ld));
      SSC_TRY(queue_dw(c, q, dga, H4, hd_prev, l.Hp, TB, H4, H, gw + E + F + H, ld));
    } else if (g->att_w_hh) {
      SSC_TRY(queue_dw(c, q, dga, H4, h1_prev, l.Hp, TB, H4, H, g->att_w_hh, g->ld_att_w_hh));
    }
    if (g->wq) SSC_TRY(queue_dw(c, q, W + l.dq, l.Ap, h1_new, l.Hp, TB, A, H, g->wq, g->ld_wq));
    SSC_TRY(flush_dw(c, q));
  }
  // attention LSTM
  if (g->att_w_ih) {
    float* gw = g->att_w_ih; int ld = g->ld_att_w_ih;
    SSC_TRY(gemm(c, false, false, {{W + l.dga_sum, H4, W + l.avg, F, B}}, H4, F, gw + E, ld));
  }
  if (g->att_w_hh) {
    // dW_hh^att = dGa^T H1_prev is the same product as the h1' block of dW_ih^att (both multiply h1'): copy, do not recompute
    if (g->att_w_ih) {
      if (hipMemcpy2DAsync(g->att_w_hh, (size_t)g->ld_att_w_hh * sizeof(float), g->att_w_ih + E + F,
                           (size_t)g->ld_att_w_ih * sizeof(float), (size_t)H * sizeof(float), H4, hipMemcpyDeviceToDevice,
                           st) != hipSuccess)
        return SSC_EHIP;
    }
  }
  if (g->att_b_ih && g->att_b_hh) {
    SSC_TRY(ssc_colsum2(dga, H4, TB, H4, nullptr, g->att_b_ih, 1, g->att_b_hh, 0, c.slabs, st));
  } else {
    if (g->att_b_ih) SSC_TRY(ssc_colsum2(dga, H4, TB, H4, nullptr, g->att_b_ih, 1, nullptr, 0, c.slabs, st));
    if (g->att_b_hh) SSC_TRY(ssc_colsum2(dga, H4, TB, H4, nullptr, g->att_b_hh, 1, nullptr, 0, c.slabs, st));
  }
  // attention projections
  if (g->wv) SSC_TRY(gemm(c, false, false, {{W + l.dpv, A, bt->feats, F, B * R}}, A, F, g->wv, g->ld_wv));
  if (g->wa) SSC_TRY(ssc_colsum(W + l.dwa, A, B, A, nullptr, g->wa, 1, 0, st));
  }  // phase 2b
  if (phases & 4u) {
  {
    DwBatch q;
    if (g->enc_w_ih) {
      float* gw = g->enc_w_ih; int ld = g->ld_enc_w_ih;
      SSC_TRY(queue_dw(c, q, dge, H4, att, l.Fp, TB, H4, F, gw, ld));
      SSC_TRY(queue_dw(c, q, dge, H4, h1_new, l.Hp, TB, H4, H, gw + F, ld));
      SSC_TRY(queue_dw(c, q, dge, H4, hd_prev, l.Hp, TB, H4, H, gw + F + H, ld));
    }
    if (g->enc_w_hh) SSC_TRY(queue_dw(c, q, dge, H4, he_prev, l.Hp, TB, H4, H, g->enc_w_hh, g->ld_enc_w_hh));
    // c-block (SENTIMENT_VAE = 2): dGe^T C over the padded Dp columns of the pooled history into the forward's aligned copy (free
    // since the BPTT loop ended); its D real columns are copied into place below
    if (g->enc_w_ih && l.D) SSC_TRY(queue_dw(c, q, dge, H4, W + l.pool, l.Dp, TB, H4, l.Dp, W + l.wc_e, l.Dp));
    if (g->fc_mean_w && g->fc_lv_w && g->fc_lv_w == g->fc_mean_w + (size_t)Z * g->ld_fc_mean_w && g->ld_fc_lv_w == g->ld_fc_mean_w) {
      // [dW_mu ; dW_lv] = (dmu | dlv)^T h_e as ONE (2Z x H) product: the two gradients are adjacent in the flat store, and 2Z keeps
      // 16-byte rows where Z alone does not (the shipped Z_SPACE = 150 sent the two Z-row products to the scalar kernel)
      SSC_TRY(queue_dw(c, q, W + l.dmulv, 2 * Z, he_new, l.Hp, TB, 2 * Z, H, g->fc_mean_w, g->ld_fc_mean_w));
    } else {
      if (g->fc_mean_w) SSC_TRY(queue_dw(c, q, W + l.dmulv, 2 * Z, he_new, l.Hp, TB, Z, H, g->fc_mean_w, g->ld_fc_mean_w));
      if (g->fc_lv_w) SSC_TRY(queue_dw(c, q, W + l.dmulv + Z, 2 * Z, he_new, l.Hp, TB, Z, H, g->fc_lv_w, g->ld_fc_lv_w));
    }
    SSC_TRY(flush_dw(c, q));
  }
  // encoder LSTM
  if (g->enc_w_ih) {
    float* gw = g->enc_w_ih; int ld = g->ld_enc_w_ih;
    if (S == 1 && !l.D) SSC_TRY(ssc_colsum2(dge, H4, TB, H4, W + l.sent_all, gw + F + 2 * H, ld, nullptr, 0, c.slabs, st));
    if (l.D && hipMemcpy2DAsync(gw + F + 2 * H, (size_t)ld * sizeof(float), W + l.wc_e, (size_t)l.Dp * sizeof(float),
                                (size_t)l.SC * sizeof(float), H4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return SSC_EHIP;
  }
  if (g->enc_b_ih && g->enc_b_hh) {
    SSC_TRY(ssc_colsum2(dge, H4, TB, H4, nullptr, g->enc_b_ih, 1, g->enc_b_hh, 0, c.slabs, st));
  } else {
    if (g->enc_b_ih) SSC_TRY(ssc_colsum2(dge, H4, TB, H4, nullptr, g->enc_b_ih, 1, nullptr, 0, c.slabs, st));
    if (g->enc_b_hh) SSC_TRY(ssc_colsum2(dge, H4, TB, H4, nullptr, g->enc_b_hh, 1, nullptr, 0, c.slabs, st));
  }
  // latent heads
  const float* dmulv = W + l.dmulv;
  if (g->fc_mean_b) SSC_TRY(ssc_colsum2(dmulv, 2 * Z, TB, Z, nullptr, g->fc_mean_b, 1, nullptr, 0, c.slabs, st));
  if (g->fc_lv_b) SSC_TRY(ssc_colsum2(dmulv + Z, 2 * Z, TB, Z, nullptr, g->fc_lv_b, 1, nullptr, 0, c.slabs, st));
  }  // phase 4
  if (phases & 8u) {
  {
    DwBatch q;
    if (g->dec_w_ih) {
      float* gw = g->dec_w_ih; int ld = g->ld_dec_w_ih;
      SSC_TRY(queue_dw(c, q, dgd, H4, att, l.Fp, TB, H4, F, gw, ld));
      SSC_TRY(queue_dw(c, q, dgd, H4, h1_new, l.Hp, TB, H4, H, gw + F, ld));
      SSC_TRY(queue_dw(c, q, dgd, H4, hd_prev, l.Hp, TB, H4, H, gw + F + H, ld));
      if (l.Zp != Z) {
        // Z no multiple of 4: the z-block product runs on the padded Zp columns of z (zero pads) into the forward's aligned
        // z-block copy - free since the BPTT loop ended - and its Z real columns are copied into place below
        SSC_TRY(queue_dw(c, q, dgd, H4, W + l.z, l.Zp, TB, H4, l.Zp, W + l.wz, l.Zp));
      } else {
        SSC_TRY(queue_dw(c, q, dgd, H4, W + l.z, l.Zp, TB, H4, Z, gw + zcol, ld));
      }
      if (l.D) SSC_TRY(queue_dw(c, q, dgd, H4, W + l.pool, l.Dp, TB, H4, l.Dp, W + l.wc_d, l.Dp));   // c-block, as for the encoder
    } else if (g->dec_w_hh) {
      SSC_TRY(queue_dw(c, q, dgd, H4, hd_prev, l.Hp, TB, H4, H, g->dec_w_hh, g->ld_dec_w_hh));
    }
    SSC_TRY(flush_dw(c, q));
  }
  // decoder LSTM (skipped while frozen: train.py:156-161)
  if (g->dec_w_ih && l.Zp != Z) {
    if (hipMemcpy2DAsync(g->dec_w_ih + zcol, (size_t)g->ld_dec_w_ih * sizeof(float), W + l.wz, (size_t)l.Zp * sizeof(float),
                         (size_t)Z * sizeof(float), H4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return SSC_EHIP;
  }
  if (g->dec_w_ih) {
    float* gw = g->dec_w_ih; int ld = g->ld_dec_w_ih;
    if (S == 1 && !l.D) SSC_TRY(ssc_colsum2(dgd, H4, TB, H4, W + l.sent_all, gw + F + 2 * H, ld, nullptr, 0, c.slabs, st));
    if (l.D && hipMemcpy2DAsync(gw + F + 2 * H, (size_t)ld * sizeof(float), W + l.wc_d, (size_t)l.Dp * sizeof(float),
                                (size_t)l.SC * sizeof(float), H4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return SSC_EHIP;
  }
  if (g->dec_w_hh) {
    // dW_hh^dec = dGd^T HD_prev is the same product as the hd' block of dW_ih^dec
    if (g->dec_w_ih) {
      if (hipMemcpy2DAsync(g->dec_w_hh, (size_t)g->ld_dec_w_hh * sizeof(float), g->dec_w_ih + F + H,
                           (size_t)g->ld_dec_w_ih * sizeof(float), (size_t)H * sizeof(float), H4, hipMemcpyDeviceToDevice,
                           st) != hipSuccess)
        return SSC_EHIP;
    }
  }
  if (g->dec_b_ih && g->dec_b_hh) {
    SSC_TRY(ssc_colsum2(dgd, H4, TB, H4, nullptr, g->dec_b_ih, 1, g->dec_b_hh, 0, c.slabs, st));
  } else {
    if (g->dec_b_ih) SSC_TRY(ssc_colsum2(dgd, H4, TB, H4, nullptr, g->dec_b_ih, 1, nullptr, 0, c.slabs, st));
    if (g->dec_b_hh) SSC_TRY(ssc_colsum2(dgd, H4, TB, H4, nullptr, g->dec_b_hh, 1, nullptr, 0, c.slabs, st));
  }
  }  // phase 8
  return SSC_OK;
}

extern "C" int ssc_train_bwd(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* bt, void* workspace,
                             size_t workspace_bytes, const float* gl, const float* gk, const ssc_params* g, void* stream) {
  return train_bwd_impl(cfg, p, bt, workspace, workspace_bytes, gl, gk, g, stream, 15u | 64u | 128u);
}

extern "C" int ssc_train_bwd_phases(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* bt, void* workspace,
                                    size_t workspace_bytes, const float* gl, const float* gk, const ssc_params* g,
                                    unsigned phases, void* stream) {
  if (phases == 0 || phases > 255u) return SSC_EINVAL;
  if (phases & 2u) phases |= 64u | 128u;   // 2 = both halves of the embedding / attention-LSTM / attention phase
  return train_bwd_impl(cfg, p, bt, workspace, workspace_bytes, gl, gk, g, stream, phases);
}

// Elementwise / small-reduction kernels of the var_updown hot path (gfx950, wave64).
// All of them are HBM-bound byte movers; they are written for coalesced access along the contiguous
// axis and 64-lane shuffle reductions.  Reference citations are on the C entry points in ssc.h.
#include <type_traits>

#include <algorithm>

#include "ssc_common.h"

thread_local int ssc_tls_hip_error = 0;

extern "C" int ssc_version(void) { return 4; }
extern "C" int ssc_last_hip_error(void) { return ssc_tls_hip_error; }
extern "C" const char* ssc_arch(void) { return "gfx950"; }

namespace {

// ---------------------------------------------------------------------------------------------
// feat_prep: mask[b,r] = (sum_f |v| > 0) ; avg[b,f] = sum_r mask*v / max(sum_r mask, 1e-8)
// ---------------------------------------------------------------------------------------------
__global__ void feat_mask_kernel(const float* __restrict__ feats, int BR, int F, float* __restrict__ mask) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= BR) return;
  const float* p = feats + (size_t)row * F;
  float s = 0.f;
  for (int f = lane; f < F; f += 64) s += fabsf(p[f]);
  s = ssc_wave_sum(s);
  if (lane == 0) mask[row] = s > 0.f ? 1.f : 0.f;
}

__global__ void feat_avg_kernel(const float* __restrict__ feats, const float* __restrict__ mask, int R, int F,
                                float* __restrict__ avg) {
  int b = blockIdx.y;
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const float* p = feats + (size_t)b * R * F + f;
  const float* m = mask + (size_t)b * R;
  float s = 0.f, n = 0.f;
  for (int r = 0; r < R; ++r) {
    float mr = m[r];
    s += mr * p[(size_t)r * F];
    n += mr;
  }
  avg[(size_t)b * F + f] = s / fmaxf(n, 1e-8f);
}

// ---------------------------------------------------------------------------------------------
__global__ void prep_tokens_kernel(const int64_t* __restrict__ caps, int B, int L, int pad, int boundary,
                                   int64_t* __restrict__ tok, float* __restrict__ w, float* __restrict__ nvalid) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0;
  for (int j = 0; j < L; ++j) n += caps[(size_t)b * L + j] != pad;
  tok[b] = boundary;
  for (int j = 0; j < L; ++j) tok[(size_t)(1 + j) * B + b] = caps[(size_t)b * L + j];
  tok[(size_t)(L + 1) * B + b] = 0;
  tok[(size_t)(n + 1) * B + b] = boundary;
  float nv = 0.f;
  for (int t = 0; t <= L; ++t) {
    float wt = tok[(size_t)(t + 1) * B + b] != pad ? 1.f : 0.f;
    w[(size_t)t * B + b] = wt;
    nv += wt;
  }
  nvalid[b] = nv;
}

__global__ void embed_gather_kernel(const float* __restrict__ table, int ldt, const int64_t* __restrict__ ids, int n, int E,
                                    float* __restrict__ out, int ldo) {
  int i = blockIdx.y;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  out[(size_t)i * ldo + e] = table[(size_t)ids[i] * ldt + e];
}

__global__ void embed_scatter_kernel(float* __restrict__ dt, int ldt, const int64_t* __restrict__ ids, int n, int E,
                                     const float* __restrict__ d, int ldd, int pad) {
  int i = blockIdx.y;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t id = ids[i];
  if (id == pad) return;
  atomicAdd(&dt[(size_t)id * ldt + e], d[(size_t)i * ldd + e]);
}

// ---------------------------------------------------------------------------------------------
// LSTM pointwise forward / backward
// ---------------------------------------------------------------------------------------------
__global__ void lstm_fwd_kernel(const ssc_lstm_fwd_desc d) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  const bool pad = j >= d.H;   // (only with h_planes: the planes' padding columns H .. roundup(H, 32) are zeroed here)
  if (pad && (!d.h_planes || j >= (d.H + 31) / 32 * 32)) return;
  if (d.rows) {   // only the listed rows (decode: the rows that are read at all)
    if (b >= *d.row_count) return;
    b = d.rows[b];
  }
  if (pad) {
    unsigned short* hp = reinterpret_cast<unsigned short*>(d.h_planes) + ((size_t)b * d.ld_hplanes + (j >> 5) * 32) * 2 + (j & 31);
    hp[0] = 0; hp[32] = 0;
    return;
  }
  const int H = d.H, H4 = 4 * d.H;
  // The kernel moves ~20 MB: it is bound by memory latency, not bandwidth.  Every operand that does not depend on the
  // slab sums is requested first and all slabs (up to 16) in one batch, so about one latency is exposed in total.
  float a0[4], a1[4], bi[4], bh[4], sw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = g * H + j;
    a0[g] = d.add0 ? d.add0[(size_t)(d.add0_rows ? d.add0_rows[b] : (int64_t)b) * d.ld_add0 + n] : 0.f;
    a1[g] = d.add1 ? d.add1[(size_t)(b / d.rows_per_add1) * d.ld_add1 + n] : 0.f;
    bi[g] = d.b_ih ? d.b_ih[n] : 0.f;
    bh[g] = d.b_hh ? d.b_hh[n] : 0.f;
    sw[g] = d.sent ? d.wcol[(size_t)n * d.ldwcol] : 0.f;
  }
  const float sv = d.sent ? d.sent[b] : 0.f;
  const float cp = d.c_prev ? d.c_prev[(size_t)(d.c_prev_rows ? d.c_prev_rows[b] : b) * d.ld_cprev + j] : 0.f;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  // split-K slabs: summed in index order per gate (a `v += load` loop with a dynamic trip count would serialise one
  // memory latency per slab), U loads per gate in flight.  U follows the slab count: the decode step hands ONE slab (the gate
  // product of 5000 rows is not split) and a fixed batch of 16 made it issue 64 loads per cell for the 4 it needs.
  const size_t srow = d.slab_rows ? (size_t)d.slab_rows[b] : (size_t)b;   // (decode: the row of this beam's parent in a product over distinct parents)
  auto add_slabs = [&](auto uc) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    for (int s0 = 0; s0 < d.nslab; s0 += U) {
      float t[4][U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* sp = d.slabs + (size_t)min(s0 + u, d.nslab - 1) * d.slab_stride + srow * H4 + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) t[g][u] = sp[g * H];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] += (s0 + u < d.nslab) ? t[g][u] : 0.f;
    }
  };
  if (d.nslab <= 1) add_slabs(std::integral_constant<int, 1>{});
  else if (d.nslab <= 4) add_slabs(std::integral_constant<int, 4>{});
  else if (d.nslab <= 8) add_slabs(std::integral_constant<int, 8>{});
  else add_slabs(std::integral_constant<int, 16>{});
  if (d.nslab2 > 0) {
    const size_t r2 = d.slab2_rows ? (size_t)d.slab2_rows[b] : (size_t)b;
    for (int sl = 0; sl < d.nslab2; ++sl)
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] += d.slabs2[(size_t)sl * d.slab2_stride + r2 * H4 + g * H + j];
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // absent terms add +0.f, which leaves every value unchanged
    float v = pre[g];
    v += a0[g];
    v += a1[g];
    v += bi[g];
    v += bh[g];
    if (d.sent) v += sv * sw[g];
    pre[g] = v;
  }
  float ig = ssc_sigmoid(pre[0]), fg = ssc_sigmoid(pre[1]), gg = tanhf(pre[2]), og = ssc_sigmoid(pre[3]);
  float c = fg * cp + ig * gg;
  float h = og * tanhf(c);
  if (d.gates_out) {
    float* go = d.gates_out + (size_t)b * H4 + j;
    go[0] = ig; go[H] = fg; go[2 * H] = gg; go[3 * H] = og;
  }
  d.c_out[(size_t)b * d.ld_cout + j] = c;
  d.h_out[(size_t)b * d.ld_hout + j] = h;
  if (d.h_planes) {   // h also as the two fp16 pieces of h * scale: halfs j of the row's k-block (hi: 32 halfs, then lo: 32 halfs)
    unsigned short hi, lo;
    ssc_split1_f16(h * (d.planes_scale ? *d.planes_scale : 1.f), hi, lo);
    unsigned short* hp = reinterpret_cast<unsigned short*>(d.h_planes) + ((size_t)b * d.ld_hplanes + (j >> 5) * 32) * 2 + (j & 31);
    hp[0] = hi; hp[32] = lo;
  }
}
// lstm_fwd_kernel plus one more addend of the gate pre-activations computed IN the kernel: pre[b,n] += z[b,:] . wz[n,:]
// (K = Z: the latent block of the decoder LSTM's input, updown_cell.py:211-229).  z only exists after the latent head of the
// same step, so as a K-segment of the gate product it would tie the whole 88 MB product to the end of the step's dependency
// chain; here the product's other segments are issued earlier (grouped with the encoder product) and the small z block costs
// no launch of its own.  One 512-thread workgroup per (32 batch rows x 16 hidden units) = one cell per thread; every load of
// the kernel (the z rows and the 64 wz rows of the tile, cell operands, slab values) is requested up front, so one memory
// latency is exposed as in lstm_fwd_kernel; wave (rt, g) then forms the 16 x 16 block of row tile rt and gate g on the
// exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) from LDS images and hands it to the cell threads through LDS.  150 workgroups at
// C2: one round of the chip (a first form with 256-thread workgroups of 16 x 16 cells needed 254 VGPRs, one workgroup per CU,
// and its 300 workgroups took two rounds: 16.8 us).
typedef float ssc_f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void lstm_fwd_z_kernel(const ssc_lstm_fwd_desc d, const float* __restrict__ z, int ldz,
                                                            const float* __restrict__ wz, int ldwz, int Z) {
  constexpr int TB = 32, TJ = 16, KT = 128, LD = KT + 4, NT = 512;   // k-tile of 128 (one pass for Z <= 128); rows padded by 4 floats
  __shared__ __attribute__((aligned(16))) float sz[TB * LD];       // z[b0 + r, k]
  __shared__ __attribute__((aligned(16))) float sw[4 * TJ * LD];   // wz[g*H + j0 + jj, k], row = g*16 + jj
  __shared__ float st[TB * 65];                                    // product tile [row b][column g*16 + jj]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = d.H, H4 = 4 * d.H;
  const int j0 = blockIdx.x * TJ, b0 = blockIdx.y * TB;
  const int bb = tid >> 4, jj = tid & 15;
  const int b = b0 + bb, j = j0 + jj;
  const bool live = b < d.B && j < H;
  const int bc = live ? b : 0, jc = live ? j : 0;   // clamped: every thread runs the same loads
  // ---- operand tiles of the product: requested first (they are needed first) ---------------------------------------------------
  float zr[TB * KT / NT], wr[4 * TJ * KT / NT];
  auto request_tiles = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < TB * KT / NT; ++u) {   // 8 floats per thread, coalesced along k
      const int idx = tid + NT * u, row = idx / KT, kk = idx % KT, k = k0 + kk, zb = b0 + row;
      zr[u] = (zb < d.B && k < Z) ? z[(size_t)zb * ldz + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4 * TJ * KT / NT; ++u) {   // 16 floats per thread
      const int idx = tid + NT * u, row = idx / KT, kk = idx % KT, k = k0 + kk, wj = j0 + (row & 15);
      wr[u] = (wj < H && k < Z) ? wz[(size_t)((row >> 4) * H + wj) * ldwz + k] : 0.f;
    }
  };
  request_tiles(0);
  // ---- this thread's cell: operands that do not depend on the product ----------------------------------------------------------
  float a0[4], a1[4], bi[4], bh[4], sw4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = g * H + jc;
    a0[g] = d.add0 ? d.add0[(size_t)(d.add0_rows ? d.add0_rows[bc] : (int64_t)bc) * d.ld_add0 + n] : 0.f;
    a1[g] = d.add1 ? d.add1[(size_t)(bc / d.rows_per_add1) * d.ld_add1 + n] : 0.f;
    bi[g] = d.b_ih ? d.b_ih[n] : 0.f;
    bh[g] = d.b_hh ? d.b_hh[n] : 0.f;
    sw4[g] = d.sent ? d.wcol[(size_t)n * d.ldwcol] : 0.f;
  }
  const float sv = d.sent ? d.sent[bc] : 0.f;
  const float cp = d.c_prev ? d.c_prev[(size_t)bc * d.ld_cprev + jc] : 0.f;
  float t0[4][16];   // first batch of slab values (later batches, if any, after the product)
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const float* sp = d.slabs + (size_t)min(u, max(d.nslab - 1, 0)) * d.slab_stride + (size_t)bc * H4 + jc;
#pragma unroll
    for (int g = 0; g < 4; ++g) t0[g][u] = d.nslab > 0 ? sp[g * H] : 0.f;
  }
  // ---- z . wz^T for the workgroup's 32 rows x (4 gates x 16 units) ----------------------------------------------------------
  // fragment convention: lane (r = lane & 15, q = lane >> 4) reads 4 consecutive k at 16 c + 4 q of row r; MFMA i of chunk c
  // takes element i of every lane, i.e. contracts k in {16 c + 4 q + i : q = 0..3} - the same for both operands
  ssc_f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, q4 = lane >> 4;
  const int rt = wave >> 2, gw = wave & 3;   // 8 waves: row tile (0, 1) x gate
  for (int k0 = 0; k0 < Z; k0 += KT) {
    if (k0 > 0) {
      __syncthreads();   // the previous k-tile has been consumed
      request_tiles(k0);
    }
#pragma unroll
    for (int u = 0; u < TB * KT / NT; ++u) { const int idx = tid + NT * u; sz[(idx / KT) * LD + idx % KT] = zr[u]; }
#pragma unroll
    for (int u = 0; u < 4 * TJ * KT / NT; ++u) { const int idx = tid + NT * u; sw[(idx / KT) * LD + idx % KT] = wr[u]; }
    __syncthreads();
    const int kend = min(KT, (Z - k0 + 15) / 16 * 16);   // whole 16-wide chunks; the tail is zero-filled
    for (int c = 0; c < kend; c += 16) {
      const float4 av = *reinterpret_cast<const float4*>(&sz[(rt * 16 + r16) * LD + c + 4 * q4]);
      const float4 bv = *reinterpret_cast<const float4*>(&sw[(gw * 16 + r16) * LD + c + 4 * q4]);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
  }
  // output layout of the 16x16 MFMA: lane holds rows 4 q + i (i = 0..3) of column r
#pragma unroll
  for (int i = 0; i < 4; ++i) st[(rt * 16 + 4 * q4 + i) * 65 + gw * 16 + r16] = acc[i];
  __syncthreads();
  // ---- cell update (same arithmetic and summation order as lstm_fwd_kernel, the z term added last) ------------------------------
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] += (u < d.nslab) ? t0[g][u] : 0.f;
  for (int s0 = 16; s0 < d.nslab; s0 += 16) {
    float t[4][16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float* sp = d.slabs + (size_t)min(s0 + u, d.nslab - 1) * d.slab_stride + (size_t)bc * H4 + jc;
#pragma unroll
      for (int g = 0; g < 4; ++g) t[g][u] = sp[g * H];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] += (s0 + u < d.nslab) ? t[g][u] : 0.f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = pre[g];
    v += a0[g];
    v += a1[g];
    v += bi[g];
    v += bh[g];
    if (d.sent) v += sv * sw4[g];
    v += st[bb * 65 + g * 16 + jj];
    pre[g] = v;
  }
  if (!live) return;
  float ig = ssc_sigmoid(pre[0]), fg = ssc_sigmoid(pre[1]), gg = tanhf(pre[2]), og = ssc_sigmoid(pre[3]);
  float c = fg * cp + ig * gg;
  float h = og * tanhf(c);
  if (d.gates_out) {
    float* go = d.gates_out + (size_t)b * H4 + j;
    go[0] = ig; go[H] = fg; go[2 * H] = gg; go[3 * H] = og;
  }
  d.c_out[(size_t)b * d.ld_cout + j] = c;
  d.h_out[(size_t)b * d.ld_hout + j] = h;
}
// lstm_fwd_kernel for rows that share per-IMAGE operands (decode: rows_per_image beam rows per image), with one more addend of the
// gate pre-activations formed in the kernel from a per-image table:
//   pre[b, n] += sum_r alpha[b, r] * P[(img(b) R + r) 4H + n],   img(b) = b / rows_per_image
// P[img, r, :] = W_ih^dec[:, :F] v_{img,r} is the decoder-gate contribution of region r, formed ONCE per image
// (ssc_decode_prepare); since the attended feature vector is sum_r alpha_r v_r (updown_cell.py:156-158) and the gate product is
// linear in it, sum_r alpha_r P_r IS the att segment of the decoder gate product (updown_cell.py:211-229) - K = R = 36 against a
// table the image's 100 rows share, instead of K = F = 2048 against the weights in every step: the largest product of a decode
// step loses 45 % of its k-steps (same value up to fp32 reassociation; SURVEY Appendix A.5 / B: per-image terms are computed
// once per image).  One 256-thread workgroup per (image, 16 hidden units): the (R x 4 gates x 16 units) table tile goes to LDS
// once, then the image's rows are taken 16 at a time, one cell per thread, alpha rows through LDS; R <= 128.
constexpr int IMG_MAXR = 128;
// (cpw: 16-row chunks per workgroup - a whole image per workgroup at C4's 50 x 100 rows, one chunk each for a single image, so that
// the grid fills the chip either way; blockIdx.y = image * ceil(chunks / cpw) + chunk group)
// 512 threads = 16 rows x 32 hidden units: a wave reads two full 128-byte lines per row-gate access (the first form had 16 units
// per workgroup: 64-byte half lines whose other half a neighbouring workgroup fetched again later - 198 us for ~300 MB).
__global__ __launch_bounds__(512) void lstm_fwd_img_kernel(const ssc_lstm_fwd_desc d, const float* __restrict__ alpha, int ldalpha,
                                                           const float* __restrict__ P, int R, int rpi, int cpw) {
  constexpr int TJ = 32, TR = 16, PW = 4 * TJ;   // units, rows per chunk, table tile width (4 gates x TJ)
  // dynamic LDS sized by R (R = 36: 20.7 KB)
  extern __shared__ float img_lds[];
  float* sP = img_lds;                     // [R][PW]
  float* sA = img_lds + (size_t)R * PW;    // [TR][R + 1]
  const int LDA = R + 1;
  const int tid = threadIdx.x, rr = tid >> 5, jj = tid & 31;
  const int H = d.H, H4 = 4 * d.H;
  const int chunks = (rpi + TR - 1) / TR, groups = (chunks + cpw - 1) / cpw;
  const int img = blockIdx.y / groups, grp = blockIdx.y - img * groups;
  const int j0 = blockIdx.x * TJ, j = j0 + jj;
  const int jc = j < H ? j : 0;
  for (int idx = tid; idx < R * PW; idx += 512) {
    const int r = idx / PW, c = idx - r * PW, g = c / TJ, ju = j0 + (c - g * TJ);
    sP[idx] = ju < H ? P[((size_t)img * R + r) * H4 + (size_t)g * H + ju] : 0.f;
  }
  float bi[4], bh[4], sw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = g * H + jc;
    bi[g] = d.b_ih ? d.b_ih[n] : 0.f;
    bh[g] = d.b_hh ? d.b_hh[n] : 0.f;
    sw[g] = d.sent ? d.wcol[(size_t)n * d.ldwcol] : 0.f;
  }
  const int row_end = min(d.B, (img + 1) * rpi);
  for (int c0 = grp * cpw * TR; c0 < min(rpi, (grp + 1) * cpw * TR); c0 += TR) {
    const int b = img * rpi + c0 + rr;
    const bool live = c0 + rr < rpi && b < row_end && j < H;
    const int bc = (c0 + rr < rpi && b < row_end) ? b : min(img * rpi, d.B - 1);
    // this thread's cell operands first (independent of the table term)
    float a0[4], a1[4], t0[4];
    const size_t r1 = d.slab_rows ? (size_t)d.slab_rows[bc] : (size_t)bc;
    const size_t r2 = d.slab2_rows ? (size_t)d.slab2_rows[bc] : (size_t)bc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = g * H + jc;
      a0[g] = d.add0 ? d.add0[(size_t)(d.add0_rows ? d.add0_rows[bc] : (int64_t)bc) * d.ld_add0 + n] : 0.f;
      a1[g] = d.add1 ? d.add1[(size_t)(bc / d.rows_per_add1) * d.ld_add1 + n] : 0.f;
      t0[g] = d.nslab > 0 ? d.slabs[r1 * H4 + n] : 0.f;
    }
    float t2[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.nslab2 > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) t2[g] = d.slabs2[r2 * H4 + g * H + jc];
    }
    const float sv = d.sent ? d.sent[bc] : 0.f;
    const float cp = d.c_prev ? d.c_prev[(size_t)(d.c_prev_rows ? d.c_prev_rows[bc] : bc) * d.ld_cprev + jc] : 0.f;
    __syncthreads();   // (the previous chunk's alpha rows have been consumed; first pass: sP's writers)
    for (int idx = tid; idx < TR * R; idx += 512) {
      const int row = idx / R, r = idx - row * R, ab = img * rpi + c0 + row;
      sA[row * LDA + r] = (c0 + row < rpi && ab < row_end) ? alpha[(size_t)ab * ldalpha + r] : 0.f;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* ar = sA + rr * LDA;
    for (int r = 0; r < R; ++r) {   // region order: fixed summation order
      const float a = ar[r];
      const float* pr = sP + r * PW + jj;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] += a * pr[g * TJ];
    }
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = t0[g];
      for (int sl = 1; sl < d.nslab; ++sl) v += d.slabs[(size_t)sl * d.slab_stride + r1 * H4 + g * H + jc];
      v += t2[g];
      for (int sl = 1; sl < d.nslab2; ++sl) v += d.slabs2[(size_t)sl * d.slab2_stride + r2 * H4 + g * H + jc];
      v += a0[g];
      v += a1[g];
      v += bi[g];
      v += bh[g];
      if (d.sent) v += sv * sw[g];
      v += acc[g];
      pre[g] = v;
    }
    if (live) {
      const float ig = ssc_sigmoid(pre[0]), fg = ssc_sigmoid(pre[1]), gg = tanhf(pre[2]), og = ssc_sigmoid(pre[3]);
      const float c = fg * cp + ig * gg;
      const float h = og * tanhf(c);
      if (d.gates_out) {
        float* go = d.gates_out + (size_t)b * H4 + j;
        go[0] = ig; go[H] = fg; go[2 * H] = gg; go[3 * H] = og;
      }
      d.c_out[(size_t)b * d.ld_cout + j] = c;
      d.h_out[(size_t)b * d.ld_hout + j] = h;
    }
  }
}

// The same cell with the table contraction on the fp32 matrix cores (v_mfma_f32_16x16x4_f32; fp32 products and sums, fixed order).
// One workgroup of eight waves per (16 hidden units, image[, row group]); the waves take the 16-row chunks round robin (one each at
// C4's 100 rows per image) and issue their chunk's operand loads before the tile is staged:
//   A operand = the image's table tile transposed, P[img][r][gate*H + u] for 16 units x 4 gates x R regions, staged once per
//               workgroup in LDS (row stride 80 floats: the four k-rows of a k-step fall on disjoint banks) - one ds_read per MFMA;
//               rows R and R + 1 of the tile hold b_ih + b_hh and the sentiment column, contracted with 1 and the row's sentiment;
//   B operand = alpha^T of the 16-row chunk (KS = ceil((R + 2) / 4) values per lane, read straight from global - alpha is L2-resident);
//   D[unit][row]: lane l holds units 4 * (l / 16) .. + 3 of row l % 16 for each gate -> the lane's 16 accumulators are exactly the
//               four gate pre-activations of four adjacent units of one row, so the cell update follows in registers and every
//               slab / state access is a float4 (H % 4 == 0, 16-byte aligned rows - checked by the caller).
// The VALU form above spent ~100 of its 190 us at C4 (5000 rows, R = 36) on LDS reads for the contraction (5 reads per 4 FMAs).
// (A first matrix-core form held the table tile in 36 registers per lane, one wave per image: ~200 registers, two waves per SIMD,
// 96 us; with the next chunk's operands prefetched by hand 114 us.)
// MODE 0: one slab read by row index, no second slab; MODE 1: + a second slab read through slab2_rows and the previous cell state
// through c_prev_rows (the sibling-dedup decode step); MODE 2: every option of the descriptor.  Modes 0 and 1 have NO conditional loads: hipcc turns `p ? *p : 0` into a branch
// with s_waitcnt vmcnt(0) at the join, which serialises the loads of a lane (the first forms of this kernel: 120-135 us).
template <int KS, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 2 : KS <= 10 ? 4 : KS <= 17 ? 3 : 2, 8))) void lstm_fwd_img_mfma_kernel(
    const ssc_lstm_fwd_desc d, const float* __restrict__ alpha, int ldalpha, const float* __restrict__ P, int R, int rpi, int cpw) {
  constexpr int LDP = 80, NW = 8;
  constexpr bool RARE = MODE == 2;
  extern __shared__ float img_lds[];   // [4 * KS][LDP], rows >= R + 2 zero
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int H = d.H, H4 = 4 * d.H;
  const int ub = blockIdx.x;
  const int chunks = (rpi + 15) / 16, groups = (chunks + cpw - 1) / cpw;
  const int img = blockIdx.y / groups, grp = blockIdx.y - img * groups;
  const int u = ub * 16 + 4 * lk;   // this lane's four units
  const bool uok = u < H;            // (H % 4 == 0: all four or none)
  const int uc = uok ? u : 0;
  const int row_end = min(d.B, (img + 1) * rpi);
  const int c_end = min(rpi, (grp + 1) * cpw * 16);
  const float* sentp = d.sent ? d.sent : alpha;   // (always a readable address: the value is dropped when there is no sentiment)
  const float sflag = d.sent ? 1.f : 0.f;
  struct Ops { ssc_f32x4v pre[4], t2[4], cp; float al[KS], sv; int b, bc; bool rok; size_t r1, r2; };
  // a chunk's operands: issued before anything that waits (the tile staging and its barrier for a wave's first chunk)
  auto issue = [&](int c0, Ops& o) {
    o.b = img * rpi + c0 + li;
    o.rok = c0 + li < rpi && o.b < row_end;
    o.bc = o.rok ? o.b : img * rpi;
    if (MODE == 2) {
      o.r1 = d.slab_rows ? (size_t)d.slab_rows[o.bc] : (size_t)o.bc;
      o.r2 = d.slab2_rows ? (size_t)d.slab2_rows[o.bc] : (size_t)o.bc;
    } else {
      o.r1 = (size_t)o.bc;
      o.r2 = MODE == 1 ? (size_t)d.slab2_rows[o.bc] : 0;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = g * H + uc;
      if (MODE == 2) o.pre[g] = d.nslab > 0 ? *reinterpret_cast<const ssc_f32x4v*>(d.slabs + o.r1 * H4 + n) : ssc_f32x4v{0.f, 0.f, 0.f, 0.f};
      else o.pre[g] = *reinterpret_cast<const ssc_f32x4v*>(d.slabs + o.r1 * H4 + n);
    }
    if (MODE == 2)
      o.cp = d.c_prev ? *reinterpret_cast<const ssc_f32x4v*>(d.c_prev + (size_t)(d.c_prev_rows ? d.c_prev_rows[o.bc] : o.bc) * d.ld_cprev + uc)
                      : ssc_f32x4v{0.f, 0.f, 0.f, 0.f};
    else if (MODE == 1) o.cp = *reinterpret_cast<const ssc_f32x4v*>(d.c_prev + (size_t)d.c_prev_rows[o.bc] * d.ld_cprev + uc);
    else o.cp = *reinterpret_cast<const ssc_f32x4v*>(d.c_prev + (size_t)o.bc * d.ld_cprev + uc);
    const float* ap = alpha + (size_t)o.bc * ldalpha;
    o.sv = sentp[o.bc];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int r = ks * 4 + lk;
      o.al[ks] = ap[min(r, R - 1)];   // (raw: masked in `finish`, after the waits of the tile staging)
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = g * H + uc;
      if (MODE == 2) o.t2[g] = d.nslab2 > 0 ? *reinterpret_cast<const ssc_f32x4v*>(d.slabs2 + o.r2 * H4 + n) : ssc_f32x4v{0.f, 0.f, 0.f, 0.f};
      else if (MODE == 1) o.t2[g] = *reinterpret_cast<const ssc_f32x4v*>(d.slabs2 + o.r2 * H4 + n);
    }
  };
  Ops o;
  int c0 = (grp * cpw + wave) * 16;
  if (c0 < c_end) issue(c0, o);
  // the tile: rows < R from the table, row R the bias sum, row R + 1 the sentiment column, the rest zero
  {
    const int c = lane, g = c >> 4, ua = ub * 16 + (c & 15);
    const bool cok = ua < H;
    const size_t n = (size_t)g * H + (cok ? ua : 0);
    const float* pi = P + (size_t)img * R * H4 + n;
    constexpr int NIT = (4 * KS + NW - 1) / NW;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = pi[(size_t)min(wave + it * NW, R - 1) * H4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = wave + it * NW;
      if (r < 4 * KS && r != R && r != R + 1) img_lds[r * LDP + c] = (r < R && cok) ? v[it] : 0.f;
    }
    if (wave == 0) {
      const float bs = (d.b_ih ? d.b_ih[n] : 0.f) + (d.b_hh ? d.b_hh[n] : 0.f);
      img_lds[R * LDP + c] = cok ? bs : 0.f;
    } else if (wave == 1) {
      img_lds[(R + 1) * LDP + c] = (d.sent && cok) ? d.wcol[n * d.ldwcol] : 0.f;
    }
  }
  __syncthreads();
  const float* sp = img_lds + lk * LDP + li;
  while (c0 < c_end) {
    // alpha^T with the rows beyond R: 1 for the bias row, the sentiment for its column, 0 elsewhere (arithmetic, not a select
    // around the load: the compiler sinks a load whose value is used on one side only into a branch)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int r = ks * 4 + lk;
      const float m = (r < R && o.rok) ? 1.f : 0.f;
      o.al[ks] = o.al[ks] * m + (r == R ? 1.f : r == R + 1 ? o.sv * sflag : 0.f);
    }
    ssc_f32x4v acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = ssc_f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(sp[ks * 4 * LDP + g * 16], o.al[ks], acc[g], 0, 0, 0);
      if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // (keeps the tile reads two k-steps ahead at most: registers)
    }
    const int b = o.b, bc = o.bc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = g * H + uc;
      if (MODE >= 1) o.pre[g] += o.t2[g];
      if (RARE) {   // split-K slabs beyond the first, the per-token and per-image rows
        for (int sl = 1; sl < d.nslab; ++sl) o.pre[g] += *reinterpret_cast<const ssc_f32x4v*>(d.slabs + (size_t)sl * d.slab_stride + o.r1 * H4 + n);
        for (int sl = 1; sl < d.nslab2; ++sl) o.pre[g] += *reinterpret_cast<const ssc_f32x4v*>(d.slabs2 + (size_t)sl * d.slab2_stride + o.r2 * H4 + n);
        if (d.add0) o.pre[g] += *reinterpret_cast<const ssc_f32x4v*>(d.add0 + (size_t)(d.add0_rows ? d.add0_rows[bc] : (int64_t)bc) * d.ld_add0 + n);
        if (d.add1) o.pre[g] += *reinterpret_cast<const ssc_f32x4v*>(d.add1 + (size_t)(bc / d.rows_per_add1) * d.ld_add1 + n);
      }
      o.pre[g] += acc[g];
    }
    if (o.rok && uok) {
      ssc_f32x4v ig, fg, gg, og, c, h;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        ig[v] = ssc_sigmoid(o.pre[0][v]); fg[v] = ssc_sigmoid(o.pre[1][v]); gg[v] = tanhf(o.pre[2][v]); og[v] = ssc_sigmoid(o.pre[3][v]);
        c[v] = fg[v] * o.cp[v] + ig[v] * gg[v];
        h[v] = og[v] * tanhf(c[v]);
      }
      if (RARE && d.gates_out) {
        float* go = d.gates_out + (size_t)b * H4 + u;
        *reinterpret_cast<ssc_f32x4v*>(go) = ig;
        *reinterpret_cast<ssc_f32x4v*>(go + H) = fg;
        *reinterpret_cast<ssc_f32x4v*>(go + 2 * H) = gg;
        *reinterpret_cast<ssc_f32x4v*>(go + 3 * H) = og;
      }
      *reinterpret_cast<ssc_f32x4v*>(d.c_out + (size_t)b * d.ld_cout + u) = c;
      *reinterpret_cast<ssc_f32x4v*>(d.h_out + (size_t)b * d.ld_hout + u) = h;
      if (d.h_planes) {   // (uniform) h also as its two fp16 pieces (ssc_lstm_fwd_desc.h_planes)
        ssc_u32x2 hi, lo;
        ssc_split4_f16(h * (d.planes_scale ? *d.planes_scale : 1.f), hi, lo);
        unsigned* hp = reinterpret_cast<unsigned*>(d.h_planes) + (size_t)b * d.ld_hplanes + ssc_plane_word(u);
        *reinterpret_cast<ssc_u32x2*>(hp) = hi;
        *reinterpret_cast<ssc_u32x2*>(hp + 16) = lo;
      }
    } else if (o.rok && d.h_planes && u < (H + 31) / 32 * 32) {   // the planes' padding columns (the grid then covers roundup(H, 32) units)
      unsigned* hp = reinterpret_cast<unsigned*>(d.h_planes) + (size_t)b * d.ld_hplanes + ssc_plane_word(u);
      *reinterpret_cast<ssc_u32x2*>(hp) = ssc_u32x2{0u, 0u};
      *reinterpret_cast<ssc_u32x2*>(hp + 16) = ssc_u32x2{0u, 0u};
    }
    c0 += NW * 16;
    if (c0 < c_end) issue(c0, o);
  }
}

// lstm_fwd_kernel that also leaves partial products of its OUTPUT: pout[blockIdx.x][b, n] = sum_{j in the workgroup's 16 units}
// h[b,j] wp[n,j]  (wp (NP,H) ld ldwp: an nn.Linear weight; NP <= 256).  The encoder LSTM's h feeds fc_mean | fc_log_var
// (updown_cell.py:196-197) in the same step: as a product of its own that was a 10 us launch on the dependency chain for 1.2 MB
// of weights; here every (32 rows x 16 units) workgroup multiplies the h tile it has just computed with its 16 weight columns
// (exact-fp32 16x16x4 MFMA, K = 16) and the consumer (latent head) sums the cdiv(H,16) partial slabs in index order.
__global__ __launch_bounds__(512, 2) void lstm_fwd_p_kernel(const ssc_lstm_fwd_desc d, const float* __restrict__ wp, int ldwp,
                                                            int NP, float* __restrict__ pout) {
  constexpr int TB = 32, TJ = 16, LD = TJ + 4, NT = 512, NPMAX = 256;
  __shared__ __attribute__((aligned(16))) float sh[TB * LD];      // h[b0 + r, j0 + k]
  __shared__ __attribute__((aligned(16))) float sw[NPMAX * LD];   // wp[n, j0 + k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = d.H, H4 = 4 * d.H;
  const int j0 = blockIdx.x * TJ, b0 = blockIdx.y * TB;
  const int bb = tid >> 4, jj = tid & 15;
  const int b = b0 + bb, j = j0 + jj;
  const bool live = b < d.B && j < H;
  const int bc = live ? b : 0, jc = live ? j : 0;   // clamped: every thread runs the same loads
  // the weight slice is requested first (independent of the cell)
  float wr[NPMAX * TJ / NT];
#pragma unroll
  for (int u = 0; u < NPMAX * TJ / NT; ++u) {   // 8 floats per thread: 16 consecutive threads read 64 contiguous bytes of a row
    const int idx = tid + NT * u, n = idx / TJ, wj = j0 + idx % TJ;
    wr[u] = (n < NP && wj < H) ? wp[(size_t)n * ldwp + wj] : 0.f;
  }
  float a0[4], a1[4], bi[4], bh[4], sw4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = g * H + jc;
    a0[g] = d.add0 ? d.add0[(size_t)(d.add0_rows ? d.add0_rows[bc] : (int64_t)bc) * d.ld_add0 + n] : 0.f;
    a1[g] = d.add1 ? d.add1[(size_t)(bc / d.rows_per_add1) * d.ld_add1 + n] : 0.f;
    bi[g] = d.b_ih ? d.b_ih[n] : 0.f;
    bh[g] = d.b_hh ? d.b_hh[n] : 0.f;
    sw4[g] = d.sent ? d.wcol[(size_t)n * d.ldwcol] : 0.f;
  }
  const float sv = d.sent ? d.sent[bc] : 0.f;
  const float cp = d.c_prev ? d.c_prev[(size_t)bc * d.ld_cprev + jc] : 0.f;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < d.nslab; s0 += 16) {
    float t[4][16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float* sp = d.slabs + (size_t)min(s0 + u, d.nslab - 1) * d.slab_stride + (size_t)bc * H4 + jc;
#pragma unroll
      for (int g = 0; g < 4; ++g) t[g][u] = sp[g * H];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] += (s0 + u < d.nslab) ? t[g][u] : 0.f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = pre[g];
    v += a0[g];
    v += a1[g];
    v += bi[g];
    v += bh[g];
    if (d.sent) v += sv * sw4[g];
    pre[g] = v;
  }
  const float ig = ssc_sigmoid(pre[0]), fg = ssc_sigmoid(pre[1]), gg = tanhf(pre[2]), og = ssc_sigmoid(pre[3]);
  const float c = fg * cp + ig * gg;
  const float h = og * tanhf(c);
  if (live) {
    if (d.gates_out) {
      float* go = d.gates_out + (size_t)b * H4 + j;
      go[0] = ig; go[H] = fg; go[2 * H] = gg; go[3 * H] = og;
    }
    d.c_out[(size_t)b * d.ld_cout + j] = c;
    d.h_out[(size_t)b * d.ld_hout + j] = h;
  }
  // ---- partial product of the h tile with the workgroup's 16 weight columns ------------------------------------------------------
  sh[bb * LD + jj] = live ? h : 0.f;
#pragma unroll
  for (int u = 0; u < NPMAX * TJ / NT; ++u) { const int idx = tid + NT * u; sw[(idx / TJ) * LD + idx % TJ] = wr[u]; }
  __syncthreads();
  const int r16 = lane & 15, q4 = lane >> 4;
  float* po = pout + (size_t)blockIdx.x * d.B * NP;
  // 2 row tiles x NP/16 column tiles; wave w takes column tiles 2w, 2w+1 (K = 16: one chunk, 4 MFMAs per tile)
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int ct = 2 * wave + ci;
    if (ct * 16 >= NP) break;
    const float4 bv = *reinterpret_cast<const float4*>(&sw[(ct * 16 + r16) * LD + 4 * q4]);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float4 av = *reinterpret_cast<const float4*>(&sh[(rt * 16 + r16) * LD + 4 * q4]);
      ssc_f32x4v acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
      const int n = ct * 16 + r16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ob = b0 + rt * 16 + 4 * q4 + i;
        if (ob < d.B && n < NP) po[(size_t)ob * NP + n] = acc[i];
      }
    }
  }
}

__global__ void lstm_bwd_kernel(const ssc_lstm_bwd_desc d) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  if (j >= d.H) return;
  const int H = d.H, H4 = 4 * d.H;
  // latency-bound like lstm_fwd_kernel: the saved activations are requested before the slab sums are consumed
  const float dcin = d.dc_in ? d.dc_in[(size_t)b * d.ld_dcin + j] : 0.f;
  const float* g = d.gates + (size_t)b * H4 + j;
  const float ig = g[0], fg = g[H], gg = g[2 * H], og = g[3 * H];
  const float cp = d.c_prev[(size_t)b * d.ld_cprev + j];
  const float cn = d.c_new[(size_t)b * d.ld_cnew + j];
  float dh = d.dh ? d.dh[(size_t)b * d.ld_dh + j] : 0.f;
  if (d.dh2) dh += d.dh2[(size_t)b * d.ld_dh2 + j];
  // split-K slabs of the producing GEMMs, fixed order, up to 16 loads in flight (the batch follows the slab count)
  auto add_slabs = [&](const float* slabs, int n, size_t stride, auto uc) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    for (int s0 = 0; s0 < n; s0 += U) {
      float t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = slabs[(size_t)min(s0 + u, n - 1) * stride + (size_t)b * H + j];
#pragma unroll
      for (int u = 0; u < U; ++u) dh += (s0 + u < n) ? t[u] : 0.f;
    }
  };
  if (d.nA > 8) add_slabs(d.slabsA, d.nA, d.strideA, std::integral_constant<int, 16>{});
  else if (d.nA > 0) add_slabs(d.slabsA, d.nA, d.strideA, std::integral_constant<int, 8>{});
  if (d.nB > 8) add_slabs(d.slabsB, d.nB, d.strideB, std::integral_constant<int, 16>{});
  else if (d.nB > 0) add_slabs(d.slabsB, d.nB, d.strideB, std::integral_constant<int, 8>{});
  float tc = tanhf(cn);
  float d_o = dh * tc;
  float dc = dcin + dh * og * (1.f - tc * tc);
  float dgi = dc * gg * ig * (1.f - ig);
  float dgf = dc * cp * fg * (1.f - fg);
  float dgg = dc * ig * (1.f - gg * gg);
  float dgo = d_o * og * (1.f - og);
  float* o = d.dG + (size_t)b * H4 + j;
  o[0] = dgi; o[H] = dgf; o[2 * H] = dgg; o[3 * H] = dgo;
  d.dc_prev[(size_t)b * d.ld_dcprev + j] = dc * fg;
  if (d.dgsum) {
    float* s = d.dgsum + (size_t)b * H4 + j;
    s[0] += dgi; s[H] += dgf; s[2 * H] += dgg; s[3 * H] += dgo;
  }
}

// lstm_bwd_kernel plus one more addend of dh formed IN the kernel: dh[b,j] += sum_k x[b,k] w[k,j]  (x (B,K) ld ldx; w (K,H) ld ldw,
// j-contiguous; K <= 1024).  Used in BPTT for the encoder LSTM - dh = carried g_he' + (dmu | dlv) . [W_mu ; W_lv], K = 2Z = 256
// (updown_cell.py:196-197 backward) - and for the attention LSTM - dh += dq . Wq, K = A = 768 (attention.py:69 backward): as
// split-K products of their own these were 10-18 us launches on the step's dependency chain for 1-4 MB of weights.  Same shape
// as lstm_fwd_z_kernel: one 512-thread workgroup per (32 batch rows x 16 hidden units), one cell per thread; the x rows and the
// (K x 16) weight slice go through LDS images to the exact-fp32 16x16x4 MFMA: wave w takes row tile w & 1 and the quarter
// w >> 1 of the K range, the four partial tiles are added in quarter order.
template <int KMAX>   // 256 | 768: bounds the staged registers (KMAX / 16 + KMAX / 32 floats per thread)
__global__ __launch_bounds__(512, 2) void lstm_bwd_x_kernel(const ssc_lstm_bwd_desc d, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, int ldw, int K) {
  constexpr int TB = 32, TJ = 16, NT = 512;
  extern __shared__ __attribute__((aligned(16))) float bwdx_lds[];
  const int KP = (K + 63) & ~63;       // four quarters of whole 16-wide chunks
  const int LD = KP + 4;
  float* sx = bwdx_lds;                // x[b0 + r, k]: TB * LD
  float* sw = sx + TB * LD;            // w[k, j0 + jj] stored [jj][k]: TJ * LD
  float* st = sw + TJ * LD;            // partial product tiles [quarter][row b][jj]: 4 * TB * 17
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = d.H, H4 = 4 * d.H;
  const int j0 = blockIdx.x * TJ, b0 = blockIdx.y * TB;
  const int bb = tid >> 4, jj = tid & 15;
  const int b = b0 + bb, j = j0 + jj;
  const bool live = b < d.B && j < H;
  const int bc = live ? b : 0, jc = live ? j : 0;   // clamped: every thread runs the same loads
  // staging without integer division by the run-time K (it cost ~35 instructions per element: 24 us per launch at K = 768):
  // x: thread (row = tid >> 4, t = tid & 15) takes k = t + 16 u of its row; w: thread (k = (tid >> 4) + 32 u, unit tid & 15).
  // Either way 16 consecutive threads read 64 contiguous bytes.
  float xr[KMAX / 16], wr[KMAX / 32];
  {
    const int xb = b0 + (tid >> 4);
    const float* xp = x + (size_t)min(xb, d.B - 1) * ldx;
#pragma unroll
    for (int u = 0; u < KMAX / 16; ++u) {
      const int k = (tid & 15) + 16 * u;
      xr[u] = (k < K && xb < d.B) ? xp[k] : 0.f;
    }
    const int wj = j0 + (tid & 15);
#pragma unroll
    for (int u = 0; u < KMAX / 32; ++u) {
      const int k = (tid >> 4) + 32 * u;
      wr[u] = (k < K && wj < H) ? w[(size_t)k * ldw + wj] : 0.f;
    }
  }
  // ---- the cell's own operands (as in lstm_bwd_kernel) -------------------------------------------------------------------------
  const float dcin = d.dc_in ? d.dc_in[(size_t)bc * d.ld_dcin + jc] : 0.f;
  const float* g = d.gates + (size_t)bc * H4 + jc;
  const float ig = g[0], fg = g[H], gg = g[2 * H], og = g[3 * H];
  const float cp = d.c_prev[(size_t)bc * d.ld_cprev + jc];
  const float cn = d.c_new[(size_t)bc * d.ld_cnew + jc];
  float dh = d.dh ? d.dh[(size_t)bc * d.ld_dh + jc] : 0.f;
  if (d.dh2) dh += d.dh2[(size_t)bc * d.ld_dh2 + jc];
  auto add_slabs = [&](const float* slabs, int n, size_t stride, auto uc) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    for (int s0 = 0; s0 < n; s0 += U) {
      float t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = slabs[(size_t)min(s0 + u, n - 1) * stride + (size_t)bc * H + jc];
#pragma unroll
      for (int u = 0; u < U; ++u) dh += (s0 + u < n) ? t[u] : 0.f;
    }
  };
  if (d.nA > 8) add_slabs(d.slabsA, d.nA, d.strideA, std::integral_constant<int, 16>{});
  else if (d.nA > 0) add_slabs(d.slabsA, d.nA, d.strideA, std::integral_constant<int, 8>{});
  if (d.nB > 8) add_slabs(d.slabsB, d.nB, d.strideB, std::integral_constant<int, 16>{});
  else if (d.nB > 0) add_slabs(d.slabsB, d.nB, d.strideB, std::integral_constant<int, 8>{});
  // ---- x . w for the workgroup's 32 rows x 16 units (fragment convention of lstm_fwd_z_kernel) -----------------------------------
#pragma unroll
  for (int u = 0; u < KMAX / 16; ++u) {
    const int k = (tid & 15) + 16 * u;
    if (k < KP) sx[(tid >> 4) * LD + k] = xr[u];
  }
#pragma unroll
  for (int u = 0; u < KMAX / 32; ++u) {
    const int k = (tid >> 4) + 32 * u;
    if (k < KP) sw[(tid & 15) * LD + k] = wr[u];
  }
  __syncthreads();
  {
    ssc_f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, q4 = lane >> 4;
    const int rt = wave & 1, kq = wave >> 1;
    const int kspan = KP >> 2;   // a multiple of 16
    for (int c = kq * kspan; c < (kq + 1) * kspan; c += 16) {
      const float4 av = *reinterpret_cast<const float4*>(&sx[(rt * 16 + r16) * LD + c + 4 * q4]);
      const float4 bv = *reinterpret_cast<const float4*>(&sw[r16 * LD + c + 4 * q4]);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st[(kq * TB + rt * 16 + 4 * q4 + i) * 17 + r16] = acc[i];
  }
  __syncthreads();
  dh += ((st[bb * 17 + jj] + st[(TB + bb) * 17 + jj]) + st[(2 * TB + bb) * 17 + jj]) + st[(3 * TB + bb) * 17 + jj];
  if (!live) return;
  float tc = tanhf(cn);
  float d_o = dh * tc;
  float dc = dcin + dh * og * (1.f - tc * tc);
  float dgi = dc * gg * ig * (1.f - ig);
  float dgf = dc * cp * fg * (1.f - fg);
  float dgg = dc * ig * (1.f - gg * gg);
  float dgo = d_o * og * (1.f - og);
  float* o = d.dG + (size_t)b * H4 + j;
  o[0] = dgi; o[H] = dgf; o[2 * H] = dgg; o[3 * H] = dgo;
  d.dc_prev[(size_t)b * d.ld_dcprev + j] = dc * fg;
  if (d.dgsum) {
    float* sp = d.dgsum + (size_t)b * H4 + j;
    sp[0] += dgi; sp[H] += dgf; sp[2 * H] += dgg; sp[3 * H] += dgo;
  }
}

// ---------------------------------------------------------------------------------------------
// latent head
// ---------------------------------------------------------------------------------------------
__global__ void latent_fwd_kernel(const ssc_latent_fwd_desc d) {
  int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (b >= d.B) return;
  const int Z = d.Z;
  const float pm_row = d.sent ? d.pm_scale * d.sent[b] : 0.f;
  float pvar = d.prior_var;
  float lpv = logf(pvar);
  float acc = 0.f;
  for (int z = lane; z < Z; z += 64) {
    float m = 0.f, l = 0.f;
    for (int s0 = 0; s0 < d.nslab; s0 += 8) {
      float tm[8], tl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* row = d.mulv + (size_t)min(s0 + u, d.nslab - 1) * d.slab_stride + (size_t)b * d.ldmulv;
        tm[u] = row[z];
        tl[u] = row[Z + z];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        m += (s0 + u < d.nslab) ? tm[u] : 0.f;
        l += (s0 + u < d.nslab) ? tl[u] : 0.f;
      }
    }
    m += d.bmu[z];
    l += d.blv[z];
    float var = expf(l);
    float zz = d.eps[(size_t)b * d.ldeps + z] * sqrtf(var) + m;
    d.mu[(size_t)b * d.ldz + z] = m;
    d.lv[(size_t)b * d.ldz + z] = l;
    d.z[(size_t)b * d.ldz + z] = zz;
    if (d.kld_mode == 0) {
      acc += 1.f + l - m * m - var;
    } else {
      const float pm = d.pm ? d.pm[(size_t)b * d.ldpm + z] : pm_row;
      float dm = m - pm;
      acc += 1.f + l - lpv - (dm * dm + var) / (pvar + 0.00001f);
    }
  }
  acc = ssc_wave_sum(acc);
  if (lane == 0) d.kld_acc[b] += d.w[b] * (-0.5f * acc);
}

__global__ void latent_prior_sample_kernel(const float* __restrict__ eps, int ldeps, const float* __restrict__ sent,
                                           float pm_scale, float sd, int G, int Z, float* __restrict__ z, int ldz) {
  int g = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Z) {   // pad columns of a 16-byte padded row are zeroed (the row is a K-segment of r4(Z) columns in the decode step)
    if (i < ((Z + 3) & ~3) && i < ldz) z[(size_t)g * ldz + i] = 0.f;
    return;
  }
  float pm = sent ? pm_scale * sent[g] : 0.f;
  z[(size_t)g * ldz + i] = eps[(size_t)g * ldeps + i] * sd + pm;
}

// the same with a per-row, per-dimension prior mean (SENTIMENT_VAE = 2: the attention-pooled attribute means, updown_cell.py:160-163)
// ... and, optionally, a per-element prior variance (a caller of _decode_step that hands its own prior, updown_captioner.py:371-381)
__global__ void latent_prior_sample_pm_kernel(const float* __restrict__ eps, int ldeps, const float* __restrict__ pm, int ldpm,
                                              const float* __restrict__ pv, int ldpv, const float* __restrict__ sent, float pm_scale,
                                              float sd, int G, int Z, float* __restrict__ z, int ldz) {
  int g = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Z) {
    if (i < ((Z + 3) & ~3) && i < ldz) z[(size_t)g * ldz + i] = 0.f;
    return;
  }
  const float m = pm ? pm[(size_t)g * ldpm + i] : (sent ? pm_scale * sent[g] : 0.f);
  const float s = pv ? sqrtf(pv[(size_t)g * ldpv + i]) : sd;
  z[(size_t)g * ldz + i] = eps[(size_t)g * ldeps + i] * s + m;
}

// latent_fwd_kernel for MANY partial slabs (the cdiv(H,16) partial products of lstm_fwd_p_kernel): one 256-thread workgroup per
// row; the slab list is split into 256 / (64 ceil(Z/64)) contiguous parts that are summed in parallel (each in index order) and
// combined in part order - fixed summation order, four times the loads in flight of the wave-per-row form.  Z <= 256.
__global__ __launch_bounds__(256) void latent_fwd_wide_kernel(const ssc_latent_fwd_desc d) {
  __shared__ float pm_[4][64], pl_[4][64], kl_[4];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int Z = d.Z;
  const int ZW = (Z + 63) / 64;            // waves per full z range (1, 2 or 4)
  const int SP = 4 / ZW;                   // slab parts
  const int zw = wave % ZW, part = wave / ZW;
  const int z = zw * 64 + lane;
  const int per = (d.nslab + SP - 1) / SP;
  const int s_lo = part * per, s_hi = min(d.nslab, s_lo + per);
  float m = 0.f, l = 0.f;
  if (z < Z && ZW * SP == 4) {
    for (int s0 = s_lo; s0 < s_hi; s0 += 16) {
      float tm[16], tl[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* row = d.mulv + (size_t)min(s0 + u, s_hi - 1) * d.slab_stride + (size_t)b * d.ldmulv;
        tm[u] = row[z];
        tl[u] = row[Z + z];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        m += (s0 + u < s_hi) ? tm[u] : 0.f;
        l += (s0 + u < s_hi) ? tl[u] : 0.f;
      }
    }
  }
  pm_[wave][lane] = m;
  pl_[wave][lane] = l;
  __syncthreads();
  float acc = 0.f;
  if (part == 0 && z < Z) {
    for (int p = 1; p < SP; ++p) { m += pm_[p * ZW + zw][lane]; l += pl_[p * ZW + zw][lane]; }
    const float pm = d.pm ? d.pm[(size_t)b * d.ldpm + z] : (d.sent ? d.pm_scale * d.sent[b] : 0.f);
    const float pvar = d.prior_var, lpv = logf(pvar);
    m += d.bmu[z];
    l += d.blv[z];
    const float var = expf(l);
    const float zz = d.eps[(size_t)b * d.ldeps + z] * sqrtf(var) + m;
    d.mu[(size_t)b * d.ldz + z] = m;
    d.lv[(size_t)b * d.ldz + z] = l;
    d.z[(size_t)b * d.ldz + z] = zz;
    if (d.kld_mode == 0) {
      acc = 1.f + l - m * m - var;
    } else {
      const float dm = m - pm;
      acc = 1.f + l - lpv - (dm * dm + var) / (pvar + 0.00001f);
    }
  }
  acc = ssc_wave_sum(acc);
  if (lane == 0) kl_[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int w2 = 0; w2 < ZW; ++w2) a += kl_[w2];   // waves of part 0, in z order
    d.kld_acc[b] += d.w[b] * (-0.5f * a);
  }
}

__global__ void latent_bwd_kernel(const ssc_latent_bwd_desc d) {
  int b = blockIdx.y;
  int z = blockIdx.x * blockDim.x + threadIdx.x;
  if (z >= d.Z) return;
  float k = d.gk[b] * d.w[b];
  float m = d.mu[(size_t)b * d.ldz + z], l = d.lv[(size_t)b * d.ldz + z];
  float dz = 0.f;
  {
    const int ns = d.nslab > 1 ? d.nslab : 1;
    for (int s0 = 0; s0 < ns; s0 += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = d.dz[(size_t)min(s0 + u, ns - 1) * d.slab_stride + (size_t)b * d.lddz + z];
#pragma unroll
      for (int u = 0; u < 8; ++u) dz += (s0 + u < ns) ? t[u] : 0.f;
    }
  }
  float e = d.eps[(size_t)b * d.ldeps + z];
  float var = expf(l);
  float dmu, dlv;
  if (d.kld_mode == 0) {
    dmu = dz + k * m;
    dlv = dz * e * 0.5f * sqrtf(var) - 0.5f * k * (1.f - var);
  } else {
    float pm = d.pm ? d.pm[(size_t)b * d.ldpm + z] : (d.sent ? d.pm_scale * d.sent[b] : 0.f);
    float den = d.prior_var + 0.00001f;
    dmu = dz + k * (m - pm) / den;
    if (d.dpm) d.dpm[(size_t)b * d.lddpm + z] = -k * (m - pm) / den;
    dlv = dz * e * 0.5f * sqrtf(var) - 0.5f * k * (1.f - var / den);
  }
  d.dmulv[(size_t)b * d.lddmulv + z] = dmu;
  d.dmulv[(size_t)b * d.lddmulv + d.Z + z] = dlv;
}

// ---------------------------------------------------------------------------------------------
// cross entropy over the vocabulary
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = is_max ? ssc_wave_max(v) : ssc_wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

__global__ void ce_fwd_kernel(const float* __restrict__ logits, int ldl, const int64_t* __restrict__ targets,
                              const float* __restrict__ w, int V, float* __restrict__ lse, float* __restrict__ nllw) {
  __shared__ float sh[16];
  int row = blockIdx.x;
  if (w[row] == 0.f) {  // padded target: no loss, and its logits row may not have been computed at all
    if (threadIdx.x == 0) { lse[row] = 0.f; nllw[row] = 0.f; }
    return;
  }
  const float* p = logits + (size_t)row * ldl;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, p[v]);
  mx = block_reduce(mx, sh, true);
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += expf(p[v] - mx);
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) {
    float l = mx + logf(s);
    lse[row] = l;
    nllw[row] = w[row] * (l - p[targets[row]]);
  }
}

__global__ void ce_loss_kernel(const float* __restrict__ nllw, const float* __restrict__ nvalid, int T, int B,
                               float* __restrict__ loss) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += nllw[(size_t)t * B + b];
  float n = nvalid[b];
  loss[b] = n * (s / (n + 1e-13f));
}

__global__ void ce_bwd_kernel(float* __restrict__ logits, int ldl, const int64_t* __restrict__ targets,
                              const float* __restrict__ w, const float* __restrict__ nvalid, const float* __restrict__ lse,
                              const float* __restrict__ gl, int B, int V) {
  int row = blockIdx.x;
  int b = row % B;
  float n = nvalid[b];
  float coef = gl[b] * w[row] * (n / (n + 1e-13f));
  float* p = logits + (size_t)row * ldl;
  float l = lse[row];
  int tgt = (int)targets[row];
  if (coef == 0.f) {
    for (int v = threadIdx.x; v < V; v += blockDim.x) p[v] = 0.f;
    return;
  }
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float sm = expf(p[v] - l);
    p[v] = (sm - (v == tgt ? 1.f : 0.f)) * coef;
  }
}

__global__ void log_softmax_kernel(const float* __restrict__ logits, int ldl, int V, float* __restrict__ out, int ldo) {
  __shared__ float sh[16];
  int row = blockIdx.x;
  const float* p = logits + (size_t)row * ldl;
  float* o = out + (size_t)row * ldo;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, p[v]);
  mx = block_reduce(mx, sh, true);
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += expf(p[v] - mx);
  s = block_reduce(s, sh, false);
  float l = mx + logf(s);
  for (int v = threadIdx.x; v < V; v += blockDim.x) o[v] = p[v] - l;
}

// ---------------------------------------------------------------------------------------------
// column sums, tanh, fill
// ---------------------------------------------------------------------------------------------
// column sums in two deterministic stages: stage 1 - workgroup (column slice of 256, row chunk) -> partials
// [chunk][N] (thread per column: 1 KiB coalesced row reads); stage 2 - sum the chunks, write out (and out2).
constexpr int COLSUM_CHUNKS = 64;   // 19 column slices x 64 row chunks = 1216 workgroups at N = 4H (latency-bound otherwise)
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ X, int ldx, int rows, int N,
                                                            const float* __restrict__ wrow, float* __restrict__ part) {
  int n = blockIdx.x * 256 + threadIdx.x;
  int chunk = blockIdx.y;
  int per = (rows + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
  int r0 = chunk * per, r1 = r0 + per < rows ? r0 + per : rows;
  if (n >= N) return;
  float s = 0.f;
  for (int r = r0; r < r1; r += 8) {  // 8 row loads in flight, summed in row order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int rr = min(r + u, r1 - 1);
      x[u] = X[(size_t)rr * ldx + n];
      if (wrow) x[u] *= wrow[rr];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (r + u < r1) ? x[u] : 0.f;
  }
  part[(size_t)chunk * N + n] = s;
}
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ part, int N, float* __restrict__ out,
                                                            int out_stride, float* __restrict__ out2, int accumulate) {
  // 64 columns per workgroup; wave q sums chunks 16q .. 16q+15 (16 loads in flight), wave 0 adds the four partial sums
  __shared__ float sh[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float t = 0.f;
  if (n < N) {
    float x[COLSUM_CHUNKS / 4];
#pragma unroll
    for (int u = 0; u < COLSUM_CHUNKS / 4; ++u) x[u] = part[(size_t)(q * (COLSUM_CHUNKS / 4) + u) * N + n];
#pragma unroll
    for (int u = 0; u < COLSUM_CHUNKS / 4; ++u) t += x[u];
  }
  sh[q][c] = t;
  __syncthreads();
  if (q != 0 || n >= N) return;
  t = ((sh[0][c] + sh[1][c]) + sh[2][c]) + sh[3][c];
  float* o = out + (size_t)n * out_stride;
  *o = accumulate ? *o + t : t;
  if (out2) out2[n] = accumulate ? out2[n] + t : t;
}

// small-row fallback (rows < 64): one pass
__global__ void colsum_kernel(const float* __restrict__ X, int ldx, int rows, int N, const float* __restrict__ wrow,
                              float* __restrict__ out, int out_stride, float* __restrict__ out2, int accumulate) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) {
    float x = X[(size_t)r * ldx + n];
    s += wrow ? wrow[r] * x : x;
  }
  float* o = out + (size_t)n * out_stride;
  *o = accumulate ? *o + s : s;
  if (out2) out2[n] = accumulate ? out2[n] + s : s;
}

__global__ void copy_strided_kernel(const float* __restrict__ src, size_t stride, int n, float* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(size_t)i * stride];
}

__global__ void bias_tanh_kernel(float* __restrict__ x, int ldx, int N, const float* __restrict__ bias) {
  int r = blockIdx.y;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float* p = x + (size_t)r * ldx + n;
  *p = tanhf(*p + (bias ? bias[n] : 0.f));
}

__global__ void tanh_bwd_kernel(float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int N) {
  int r = blockIdx.y;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float yy = y[(size_t)r * ldy + n];
  dy[(size_t)r * lddy + n] *= (1.f - yy * yy);
}

__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// clip + SGD on flat buffers
// ---------------------------------------------------------------------------------------------
__global__ void sq_norm_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ scratch) {
  __shared__ float sh[16];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  if (ssc_aligned16_dev(g)) {  // 16 B/lane stream over the bulk, scalar tail
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t k = i; k < n4; k += stride) {
      float4 v = g4[k];
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (size_t k = (n4 << 2) + i; k < n; k += stride) s += g[k] * g[k];
  } else {
    for (; i < n; i += stride) {
      float v = g[i];
      s += v * v;
    }
  }
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}

__global__ void sq_norm_final_kernel(const float* __restrict__ scratch, int nb, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += scratch[i];
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) *out = s;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n,
                           const float* __restrict__ sqnorm, float gscale, float max_norm, float lr, float momentum,
                           float wd, int first) {
  float norm = sqrtf(*sqnorm) * gscale;
  float coef = fminf(1.f, max_norm / (norm + 1e-6f)) * gscale;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float pv = p[i];
    float dd = g[i] * coef + wd * pv;
    float b = first ? dd : momentum * buf[i] + dd;
    buf[i] = b;
    p[i] = pv - lr * b;
  }
}

inline hipStream_t S(void* s) { return (hipStream_t)s; }

}  // namespace

// =============================================================================================
extern "C" int ssc_feat_prep(const float* feats, int B, int R, int F, float* mask, float* avg, void* stream) {
  if (!feats || !mask || !avg || B <= 0 || R <= 0 || F <= 0) return SSC_EINVAL;
  int BR = B * R;
  SSC_LAUNCH(feat_mask_kernel, dim3(ssc_cdiv(BR, 4)), dim3(256), 0, S(stream), feats, BR, F, mask);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(feat_avg_kernel, dim3(ssc_cdiv(F, 256), B), dim3(256), 0, S(stream), feats, mask, R, F, avg);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_prep_tokens(const int64_t* caps, int B, int L, int pad, int boundary, int64_t* tokens_tm, float* w_tm,
                               float* nvalid, void* stream) {
  if (!caps || !tokens_tm || !w_tm || !nvalid || B <= 0 || L <= 0) return SSC_EINVAL;
  SSC_LAUNCH(prep_tokens_kernel, dim3(ssc_cdiv(B, 64)), dim3(64), 0, S(stream), caps, B, L, pad, boundary,
                     tokens_tm, w_tm, nvalid);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_embed_gather(const float* table, int ldt, const int64_t* ids, int n, int E, float* out, int ldo,
                                void* stream) {
  if (!table || !ids || !out || n <= 0 || E <= 0 || ldt < E || ldo < E) return SSC_EINVAL;
  SSC_LAUNCH(embed_gather_kernel, dim3(ssc_cdiv(E, 256), n), dim3(256), 0, S(stream), table, ldt, ids, n, E, out,
                     ldo);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_embed_scatter_add(float* dtable, int ldt, const int64_t* ids, int n, int E, const float* d, int ldd,
                                     int pad, void* stream) {
  if (!dtable || !ids || !d || n <= 0 || E <= 0 || ldt < E || ldd < E) return SSC_EINVAL;
  SSC_LAUNCH(embed_scatter_kernel, dim3(ssc_cdiv(E, 256), n), dim3(256), 0, S(stream), dtable, ldt, ids, n, E, d,
                     ldd, pad);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_lstm_fwd(const ssc_lstm_fwd_desc* d, void* stream) {
  if (!d || d->B <= 0 || d->H <= 0 || !d->c_out || !d->h_out) return SSC_EINVAL;
  if (d->nslab > 0 && !d->slabs) return SSC_EINVAL;
  if (d->sent && !d->wcol) return SSC_EINVAL;
  if (d->add1 && d->rows_per_add1 <= 0) return SSC_EINVAL;
  if ((d->rows != nullptr) != (d->row_count != nullptr)) return SSC_EINVAL;
  if (d->h_planes && (d->ld_hplanes < (d->H + 31) / 32 * 32 || (d->ld_hplanes & 3) || (reinterpret_cast<uintptr_t>(d->h_planes) & 15))) return SSC_EINVAL;
  SSC_LAUNCH(lstm_fwd_kernel, dim3(ssc_cdiv(d->H, 128), d->B), dim3(128), 0, S(stream), *d);   // (128 threads per block: the grid covers roundup(H, 32))
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_lstm_fwd_z(const ssc_lstm_fwd_desc* d, const float* z, int ldz, const float* wz, int ldwz, int Z, void* stream) {
  if (!d || d->B <= 0 || d->H <= 0 || !d->c_out || !d->h_out) return SSC_EINVAL;
  if (d->nslab < 0 || (d->nslab > 0 && !d->slabs)) return SSC_EINVAL;
  if (d->sent && !d->wcol) return SSC_EINVAL;
  if (d->add1 && d->rows_per_add1 <= 0) return SSC_EINVAL;
  if (!z || !wz || Z <= 0 || ldz < Z || ldwz < Z) return SSC_EINVAL;
  if (d->h_planes) return SSC_EINVAL;   // (the fp16 pieces are written by ssc_lstm_fwd / ssc_lstm_fwd_img only)
  SSC_LAUNCH(lstm_fwd_z_kernel, dim3(ssc_cdiv(d->H, 16), ssc_cdiv(d->B, 32)), dim3(512), 0, S(stream), *d, z, ldz, wz, ldwz, Z);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

int ssc_g_img_mfma = ssc_env_int("SSC_IMG_MFMA", 1);   // 0: the VALU form of lstm_fwd_img_kernel (the fallback for H % 4 != 0)
int ssc_g_img_cpw = ssc_env_int("SSC_IMG_CPW", 0);   // tuning: 16-row chunks per workgroup of lstm_fwd_img_kernel (0 = by grid size)
extern "C" int ssc_lstm_fwd_img(const ssc_lstm_fwd_desc* d, const float* alpha, int ldalpha, const float* P, int R,
                                int rows_per_image, void* stream) {
  if (!d || d->B <= 0 || d->H <= 0 || !d->c_out || !d->h_out) return SSC_EINVAL;
  if (d->nslab < 0 || (d->nslab > 0 && !d->slabs)) return SSC_EINVAL;
  if (d->sent && !d->wcol) return SSC_EINVAL;
  if (d->add1 && d->rows_per_add1 <= 0) return SSC_EINVAL;
  if (!alpha || !P || R <= 0 || R > IMG_MAXR || ldalpha < R || rows_per_image <= 0 || d->B % rows_per_image != 0) return SSC_EINVAL;
  const int nimg = d->B / rows_per_image, chunks = ssc_cdiv(rows_per_image, 16);
  // matrix-core form: every row access is a float4
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = d->H % 4 == 0 && al16(d->slabs) && al16(d->slabs2) && al16(d->c_prev) && al16(d->c_out) && al16(d->h_out) &&
                   al16(d->gates_out) && al16(d->add0) && al16(d->add1) && al16(d->b_ih) && al16(d->b_hh) && d->ld_cprev % 4 == 0 &&
                   d->ld_cout % 4 == 0 && d->ld_hout % 4 == 0 && d->ld_add0 % 4 == 0 && d->ld_add1 % 4 == 0 &&
                   d->slab_stride % 4 == 0 && d->slab2_stride % 4 == 0;
  if (d->h_planes && (!(vec && ssc_g_img_mfma) || d->ld_hplanes < (d->H + 31) / 32 * 32 || (d->ld_hplanes & 3) || !al16(d->h_planes)))
    return SSC_EINVAL;   // (the pieces are written by the matrix-core form only)
  if (vec && ssc_g_img_mfma) {
    const int gx = ssc_cdiv(d->h_planes ? (d->H + 31) / 32 * 32 : d->H, 16);   // (+ the planes' padding columns)
    int cpw = (int)(((long)gx * nimg * chunks) / 4096);   // >= ~4096 workgroups when the rows allow it (the table tile is staged per row group)
    if (ssc_g_img_cpw > 0) cpw = ssc_g_img_cpw;
    cpw = std::min(std::max(cpw, 8), std::max(chunks, 8));   // (eight waves, a chunk each)
    const dim3 grid(gx, nimg * ssc_cdiv(chunks, cpw));
    // (mode 2: split-K slabs beyond the first, per-token / per-image rows, saved gates, a missing state or slab - the decode step
    // uses none of them)
    const bool rare = d->nslab != 1 || d->nslab2 > 1 || d->add0 || d->add1 || d->gates_out || d->slab_rows || !d->c_prev ||
                      (d->nslab2 == 1 && !(d->slab2_rows && d->c_prev_rows)) || (d->nslab2 == 0 && d->c_prev_rows);
    const int mode = rare ? 2 : d->nslab2 == 1 ? 1 : 0;
#define SSC_IMG_LAUNCH(KS)                                                                                                           \
  do {                                                                                                                             \
    const size_t lds = (size_t)4 * KS * 80 * sizeof(float);                                                                        \
    const void* fn = mode == 2 ? (const void*)lstm_fwd_img_mfma_kernel<KS, 2>                                                      \
                               : mode == 1 ? (const void*)lstm_fwd_img_mfma_kernel<KS, 1> : (const void*)lstm_fwd_img_mfma_kernel<KS, 0>; \
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SSC_EHIP; \
    if (mode == 2) SSC_LAUNCH((lstm_fwd_img_mfma_kernel<KS, 2>), grid, dim3(512), lds, S(stream), *d, alpha, ldalpha, P, R, rows_per_image, cpw); \
    else if (mode == 1) SSC_LAUNCH((lstm_fwd_img_mfma_kernel<KS, 1>), grid, dim3(512), lds, S(stream), *d, alpha, ldalpha, P, R, rows_per_image, cpw); \
    else SSC_LAUNCH((lstm_fwd_img_mfma_kernel<KS, 0>), grid, dim3(512), lds, S(stream), *d, alpha, ldalpha, P, R, rows_per_image, cpw); \
  } while (0)
    if (R <= 38) SSC_IMG_LAUNCH(10);   // (KS k-steps of 4 cover the R regions + the bias and sentiment rows)
    else if (R <= 66) SSC_IMG_LAUNCH(17);
    else SSC_IMG_LAUNCH(33);
#undef SSC_IMG_LAUNCH
    SSC_CHECK_LAUNCH();
    return SSC_OK;
  }
  const int gx = ssc_cdiv(d->H, 32);
  int cpw = (int)(((long)gx * nimg * chunks) / 2048);   // ~2048 workgroups when the rows allow it
  if (ssc_g_img_cpw > 0) cpw = ssc_g_img_cpw;
  if (cpw < 1) cpw = 1;
  if (cpw > chunks) cpw = chunks;
  const size_t lds = ((size_t)R * 128 + 16 * (size_t)(R + 1)) * sizeof(float);   // <= 64 KB + 8 KB at R = 128
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)lstm_fwd_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return SSC_EHIP;
  }
  SSC_LAUNCH(lstm_fwd_img_kernel, dim3(gx, nimg * ssc_cdiv(chunks, cpw)), dim3(512), lds, S(stream), *d, alpha, ldalpha, P, R,
             rows_per_image, cpw);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_lstm_bwd(const ssc_lstm_bwd_desc* d, void* stream) {
  if (!d || d->B <= 0 || d->H <= 0 || !d->gates || !d->c_prev || !d->c_new || !d->dG || !d->dc_prev) return SSC_EINVAL;
  if ((d->nA > 0 && !d->slabsA) || (d->nB > 0 && !d->slabsB) || d->nA < 0 || d->nB < 0) return SSC_EINVAL;
  SSC_LAUNCH(lstm_bwd_kernel, dim3(ssc_cdiv(d->H, 128), d->B), dim3(128), 0, S(stream), *d);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_lstm_fwd_p(const ssc_lstm_fwd_desc* d, const float* wp, int ldwp, int NP, float* pout, void* stream) {
  if (!d || d->B <= 0 || d->H <= 0 || !d->c_out || !d->h_out) return SSC_EINVAL;
  if (d->nslab < 0 || (d->nslab > 0 && !d->slabs)) return SSC_EINVAL;
  if (d->sent && !d->wcol) return SSC_EINVAL;
  if (d->add1 && d->rows_per_add1 <= 0) return SSC_EINVAL;
  if (!wp || !pout || NP <= 0 || NP > 256 || ldwp < d->H) return SSC_EINVAL;
  if (d->h_planes) return SSC_EINVAL;   // (the fp16 pieces are written by ssc_lstm_fwd / ssc_lstm_fwd_img only)
  SSC_LAUNCH(lstm_fwd_p_kernel, dim3(ssc_cdiv(d->H, 16), ssc_cdiv(d->B, 32)), dim3(512), 0, S(stream), *d, wp, ldwp, NP, pout);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_lstm_bwd_x(const ssc_lstm_bwd_desc* d, const float* x, int ldx, const float* w, int ldw, int K, void* stream) {
  if (!d || d->B <= 0 || d->H <= 0 || !d->gates || !d->c_prev || !d->c_new || !d->dG || !d->dc_prev) return SSC_EINVAL;
  if ((d->nA > 0 && !d->slabsA) || (d->nB > 0 && !d->slabsB) || d->nA < 0 || d->nB < 0) return SSC_EINVAL;
  if (!x || !w || K <= 0 || K > 768 || ldx < K || ldw < d->H) return SSC_EINVAL;
  const int KP = (K + 63) & ~63;
  const size_t lds = ((size_t)(32 + 16) * (KP + 4) + 4 * 32 * 17) * sizeof(float);
  const dim3 grid(ssc_cdiv(d->H, 16), ssc_cdiv(d->B, 32));
  if (K <= 256) {
    SSC_LAUNCH(lstm_bwd_x_kernel<256>, grid, dim3(512), lds, S(stream), *d, x, ldx, w, ldw, K);
  } else {
    // up to 157 KB of dynamic LDS at K = 768 (one workgroup per CU).  The attribute is per device: set on every call (cheap)
    if (hipFuncSetAttribute((const void*)lstm_bwd_x_kernel<768>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SSC_EHIP;
    if (lds > 160 * 1024) return SSC_EINVAL;
    SSC_LAUNCH(lstm_bwd_x_kernel<768>, grid, dim3(512), lds, S(stream), *d, x, ldx, w, ldw, K);
  }
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_latent_fwd(const ssc_latent_fwd_desc* d, void* stream) {
  if (!d || d->B <= 0 || d->Z <= 0 || !d->mulv || d->nslab < 1 || !d->bmu || !d->blv || !d->eps || !d->w || !d->mu ||
      !d->lv || !d->z || !d->kld_acc)
    return SSC_EINVAL;
  const int zw = (d->Z + 63) / 64;
  if (d->nslab > 16 && (zw == 1 || zw == 2 || zw == 4))
    SSC_LAUNCH(latent_fwd_wide_kernel, dim3(d->B), dim3(256), 0, S(stream), *d);
  else
    SSC_LAUNCH(latent_fwd_kernel, dim3(ssc_cdiv(d->B, 4)), dim3(256), 0, S(stream), *d);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_latent_prior_sample(const float* eps, int ldeps, const float* sent, float pm_scale, float prior_var,
                                       int G, int Z, float* z, int ldz, void* stream) {
  if (!eps || !z || G <= 0 || Z <= 0) return SSC_EINVAL;
  SSC_LAUNCH(latent_prior_sample_kernel, dim3(ssc_cdiv((Z + 3) & ~3, 64), G), dim3(64), 0, S(stream), eps, ldeps, sent,
                     pm_scale, sqrtf(prior_var), G, Z, z, ldz);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_latent_prior_sample_pm(const float* eps, int ldeps, const float* pm, int ldpm, const float* pv, int ldpv,
                                          const float* sent, float pm_scale, float prior_var, int G, int Z, float* z, int ldz,
                                          void* stream) {
  if (!eps || !z || G <= 0 || Z <= 0 || ldeps < Z || (pm && ldpm < Z) || (pv && ldpv < Z) || ldz < Z) return SSC_EINVAL;
  SSC_LAUNCH(latent_prior_sample_pm_kernel, dim3(ssc_cdiv((Z + 3) & ~3, 64), G), dim3(64), 0, S(stream), eps, ldeps, pm, ldpm, pv, ldpv, sent,
             pm_scale, sqrtf(prior_var), G, Z, z, ldz);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_latent_bwd(const ssc_latent_bwd_desc* d, void* stream) {
  if (!d || d->B <= 0 || d->Z <= 0 || !d->dz || !d->eps || !d->mu || !d->lv || !d->w || !d->gk || !d->dmulv)
    return SSC_EINVAL;
  SSC_LAUNCH(latent_bwd_kernel, dim3(ssc_cdiv(d->Z, 64), d->B), dim3(64), 0, S(stream), *d);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// lse must hold 2*T*B floats: [lse | w*nll]
extern "C" int ssc_ce_fwd(const float* logits, int ldl, const int64_t* targets, const float* w, const float* nvalid, int T,
                          int B, int V, float* lse, float* loss, void* stream) {
  if (!logits || !targets || !w || !nvalid || !lse || !loss || T <= 0 || B <= 0 || V <= 0 || ldl < V) return SSC_EINVAL;
  int rows = T * B;
  SSC_LAUNCH(ce_fwd_kernel, dim3(rows), dim3(256), 0, S(stream), logits, ldl, targets, w, V, lse, lse + rows);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(ce_loss_kernel, dim3(ssc_cdiv(B, 64)), dim3(64), 0, S(stream), lse + rows, nvalid, T, B, loss);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_ce_bwd(float* logits, int ldl, const int64_t* targets, const float* w, const float* nvalid,
                          const float* lse, const float* gl, int T, int B, int V, void* stream) {
  if (!logits || !targets || !w || !nvalid || !lse || !gl || T <= 0 || B <= 0 || V <= 0 || ldl < V) return SSC_EINVAL;
  SSC_LAUNCH(ce_bwd_kernel, dim3(T * B), dim3(256), 0, S(stream), logits, ldl, targets, w, nvalid, lse, gl, B, V);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_log_softmax(const float* logits, int ldl, int rows, int V, float* out, int ldo, void* stream) {
  if (!logits || !out || rows <= 0 || V <= 0 || ldl < V || ldo < V) return SSC_EINVAL;
  SSC_LAUNCH(log_softmax_kernel, dim3(rows), dim3(256), 0, S(stream), logits, ldl, V, out, ldo);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// scratch: COLSUM_CHUNKS*N floats when rows >= 64 (may be null for fewer rows); out2: optional second copy (N, stride 1)
extern "C" int ssc_colsum2(const float* X, int ldx, int rows, int N, const float* wrow, float* out, int out_stride,
                           float* out2, int accumulate, float* scratch, void* stream) {
  if (!X || !out || rows <= 0 || N <= 0 || ldx < N || out_stride < 1) return SSC_EINVAL;
  if (rows >= 64 && scratch) {
    SSC_LAUNCH(colsum_stage1_kernel, dim3(ssc_cdiv(N, 256), COLSUM_CHUNKS), dim3(256), 0, S(stream), X, ldx, rows, N,
                       wrow, scratch);
    SSC_CHECK_LAUNCH();
    SSC_LAUNCH(colsum_stage2_kernel, dim3(ssc_cdiv(N, 64)), dim3(256), 0, S(stream), scratch, N, out, out_stride,
                       out2, accumulate);
    SSC_CHECK_LAUNCH();
    return SSC_OK;
  }
  SSC_LAUNCH(colsum_kernel, dim3(ssc_cdiv(N, 256)), dim3(256), 0, S(stream), X, ldx, rows, N, wrow, out, out_stride,
                     out2, accumulate);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_colsum(const float* X, int ldx, int rows, int N, const float* wrow, float* out, int out_stride,
                          int accumulate, void* stream) {
  return ssc_colsum2(X, ldx, rows, N, wrow, out, out_stride, nullptr, accumulate, nullptr, stream);
}

namespace {
// |x| maxima as unsigned bit patterns (non-negative floats order like their bit patterns): one atomicMax per wave
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t rows, int cols, size_t ld,
                                                     unsigned* __restrict__ out) {
  float m = 0.f;
  const size_t n = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols;
    m = fmaxf(m, fabsf(x[r * ld + (i - r * cols)]));
  }
  m = ssc_wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
__global__ void pow2_scale_kernel(const unsigned* __restrict__ mx, int target_log2, float* __restrict__ out, int combine) {
  const float m = __uint_as_float(*mx);
  float sc = 1.f;
  if (m > 0.f && m < INFINITY) {
    int e;
    (void)frexpf(m, &e);           // m = f 2^e, f in [0.5, 1): m <= 2^e
    sc = ldexpf(1.f, target_log2 - e);
  }
  *out = combine ? fminf(*out, sc) : sc;
}
}  // namespace

extern "C" int ssc_pow2_scale(const float* x, size_t rows, int cols, size_t ld, int target_log2, float* out, int combine,
                              float* scratch, void* stream) {
  if (!x || !out || !scratch || rows == 0 || cols <= 0 || ld < (size_t)cols) return SSC_EINVAL;
  if (hipMemsetAsync(scratch, 0, sizeof(float), S(stream)) != hipSuccess) return SSC_EHIP;
  const size_t n = rows * (size_t)cols;
  const int grid = (int)std::min<size_t>((n + 1023) / 1024, 2048);
  SSC_LAUNCH(absmax_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, S(stream), x, rows, cols, ld, reinterpret_cast<unsigned*>(scratch));
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(pow2_scale_kernel, dim3(1), dim3(1), 0, S(stream), reinterpret_cast<const unsigned*>(scratch), target_log2, out, combine);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_copy_strided(const float* src, size_t stride, int n, float* dst, void* stream) {
  if (!src || !dst || n <= 0) return SSC_EINVAL;
  SSC_LAUNCH(copy_strided_kernel, dim3(ssc_cdiv(n, 256)), dim3(256), 0, S(stream), src, stride, n, dst);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_bias_tanh(float* x, int ldx, int rows, int N, const float* bias, void* stream) {
  if (!x || rows <= 0 || N <= 0 || ldx < N) return SSC_EINVAL;
  SSC_LAUNCH(bias_tanh_kernel, dim3(ssc_cdiv(N, 256), rows), dim3(256), 0, S(stream), x, ldx, N, bias);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_tanh_bwd(float* dy, int lddy, const float* y, int ldy, int rows, int N, void* stream) {
  if (!dy || !y || rows <= 0 || N <= 0) return SSC_EINVAL;
  SSC_LAUNCH(tanh_bwd_kernel, dim3(ssc_cdiv(N, 256), rows), dim3(256), 0, S(stream), dy, lddy, y, ldy, N);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_fill(float* p, size_t n, float v, void* stream) {
  if (!p) return SSC_EINVAL;
  if (n == 0) return SSC_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  SSC_LAUNCH(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, S(stream), p, n, v);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_sq_norm(const float* g, size_t n, float* scratch, float* out, void* stream) {
  if (!g || !scratch || !out) return SSC_EINVAL;
  const int nb = 1024;
  SSC_LAUNCH(sq_norm_partial_kernel, dim3(nb), dim3(256), 0, S(stream), g, n, scratch);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(sq_norm_final_kernel, dim3(1), dim3(256), 0, S(stream), scratch, nb, out);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_sgd_step(float* p, const float* g, float* buf, size_t n, const float* sqnorm, float gscale,
                            float max_norm, float lr, float momentum, float weight_decay, int first, void* stream) {
  if (!p || !g || !buf || !sqnorm) return SSC_EINVAL;
  if (n == 0) return SSC_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  SSC_LAUNCH(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, S(stream), p, g, buf, n, sqnorm, gscale, max_norm,
                     lr, momentum, weight_decay, first);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// Bottom-up top-down attention step for gfx950: additive attention logits, allennlp-style masked
// softmax over R regions and the weighted sum of region features, plus the backward.
// Reference: updown-baseline/updown/modules/attention.py:69-95 (logits + masked_softmax),
// var_updown/var_updown/modules/updown_cell.py:151-158 (weighted sum).  The reference's
// q.repeat (attention.py:78-80) is never materialised.
//
// HBM-bound (AI ~ 0.55 flop/B): per step it must read pv (G,R,A) and feats (G,R,F) once.
//   logits : one 64-lane wave per (row, region) - 16 B/lane reads of pv, shuffle reduction of the
//            A-dot; grid = G*R waves (2304 at B=64,R=36) fills the 256 CUs.
//   apply  : one wave per (row, 256-float feature chunk): the wave re-derives the row's softmax with
//            shuffles (R <= a few hundred, negligible) and streams its feature columns once with
//            R independent 1 KiB loads in flight; grid = G * F/256 waves.
#include "ssc_common.h"

namespace {

// q may arrive as `nslab` split-K partial slabs of the q-projection GEMM ((G,A), ld A each).  One 256-thread workgroup per
// (row g, group of RG = 8 regions): the workgroup sums the slabs ONCE into LDS (fixed order; round 1 had every (row, region)
// wave re-sum them: 36 x 24 KB of L2 reads per row at C2), the first group of each row writes the reduced q for the backward
// pass, and every wave takes two of the group's regions - their pv rows are requested before the q sum is waited for.
constexpr int ATTN_RG = 8;
constexpr int ATTN_MAXA = 4096;   // floats of q kept in LDS
__global__ __launch_bounds__(256) void attn_logits_kernel(const float* __restrict__ q, int ldq, int nslab, size_t slab_stride,
                                                          float* __restrict__ q_out, int ldqo,
                                                          const float* __restrict__ pv, const float* __restrict__ wa,
                                                          int G, int R, int A, int rows_per_image,
                                                          float* __restrict__ logits, const int* __restrict__ rows,
                                                          const int* __restrict__ row_count) {
  __shared__ __attribute__((aligned(16))) float sq[ATTN_MAXA];
  const int groups = (R + ATTN_RG - 1) / ATTN_RG;
  int g = blockIdx.x / groups;
  const int grp = blockIdx.x - g * groups;
  if (rows) {   // decode: only the listed rows (a device-side list of the rows that are read at all: ssc_decode_step_desc.row_lp)
    if (g >= *row_count) return;
    g = rows[g];
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int img = g / rows_per_image;
  const float* qp = q + (size_t)g * ldq;
  const bool vec = ((A & 3) == 0) && ((ldq & 3) == 0) && ((ldqo & 3) == 0) && ((slab_stride & 3) == 0) && ssc_aligned16_dev(qp) &&
                   ssc_aligned16_dev(pv) && ssc_aligned16_dev(wa) && (!q_out || ssc_aligned16_dev(q_out));
  // ---- this wave's pv rows are requested first (A <= 1024: 4 float4 per lane and region, 2 regions per wave) ----------------
  const bool pre_ok = vec && A <= 1024;
  float4 pre[ATTN_RG / 4][4];
  if (pre_ok) {
#pragma unroll
    for (int j = 0; j < ATTN_RG / 4; ++j) {
      const int r = min(grp * ATTN_RG + wave + 4 * j, R - 1);
      const float* pp = pv + ((size_t)img * R + r) * A;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = min(lane * 4 + 256 * u, A - 4);
        pre[j][u] = *reinterpret_cast<const float4*>(pp + a);
      }
    }
  }
  // ---- q: slab sum into LDS -------------------------------------------------------------------------------------------
  if (vec) {
    for (int a = tid * 4; a < A; a += 1024) {
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s0 = 0; s0 < nslab; s0 += 8) {  // fixed order, 8 slab loads in flight
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(qp + (size_t)min(s0 + u, nslab - 1) * slab_stride + a);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < nslab) { qv.x += t[u].x; qv.y += t[u].y; qv.z += t[u].z; qv.w += t[u].w; }
      }
      if (q_out && grp == 0) *reinterpret_cast<float4*>(q_out + (size_t)g * ldqo + a) = qv;
      *reinterpret_cast<float4*>(&sq[a]) = qv;
    }
  } else {
    for (int a = tid; a < A; a += 256) {
      float qv = qp[a];
      for (int sl = 1; sl < nslab; ++sl) qv += qp[sl * slab_stride + a];
      if (q_out && grp == 0) q_out[(size_t)g * ldqo + a] = qv;
      sq[a] = qv;
    }
  }
  __syncthreads();
  // ---- this wave's regions -----------------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < ATTN_RG / 4; ++j) {
    const int r = grp * ATTN_RG + wave + 4 * j;
    if (r >= R) break;
    const float* pp = pv + ((size_t)img * R + r) * A;
    float s = 0.f;
    if (pre_ok) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = lane * 4 + 256 * u;
        if (a < A) {
          const float4 qv = *reinterpret_cast<const float4*>(&sq[a]);
          const float4 p4 = pre[j][u];
          const float4 w4 = *reinterpret_cast<const float4*>(wa + a);
          s += w4.x * ssc_tanh_fast(qv.x + p4.x) + w4.y * ssc_tanh_fast(qv.y + p4.y) + w4.z * ssc_tanh_fast(qv.z + p4.z) + w4.w * ssc_tanh_fast(qv.w + p4.w);
        }
      }
    } else if (vec) {
      for (int a = lane * 4; a < A; a += 256) {
        const float4 qv = *reinterpret_cast<const float4*>(&sq[a]);
        const float4 p4 = *reinterpret_cast<const float4*>(pp + a);
        const float4 w4 = *reinterpret_cast<const float4*>(wa + a);
        s += w4.x * ssc_tanh_fast(qv.x + p4.x) + w4.y * ssc_tanh_fast(qv.y + p4.y) + w4.z * ssc_tanh_fast(qv.z + p4.z) + w4.w * ssc_tanh_fast(qv.w + p4.w);
      }
    } else {
      for (int a = lane; a < A; a += 64) s += wa[a] * ssc_tanh_fast(sq[a] + pp[a]);
    }
    s = ssc_wave_sum(s);
    if (lane == 0) logits[(size_t)g * R + r] = s;
  }
}

// masked softmax of one row held across the wave's lanes (R strided by 64), allennlp semantics:
// x = l*m ; p = softmax(x)*m ; alpha = p / (sum p + 1e-13)
template <int MAXR_PER_LANE>
__device__ __forceinline__ void wave_masked_softmax(const float* __restrict__ l, const float* __restrict__ m, int R, int lane,
                                                    float (&alpha)[MAXR_PER_LANE]) {
  float x[MAXR_PER_LANE], mk[MAXR_PER_LANE];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXR_PER_LANE; ++i) {
    int r = lane + 64 * i;
    if (r < R) {
      mk[i] = m[r];
      x[i] = l[r] * mk[i];
      mx = fmaxf(mx, x[i]);
    } else {
      mk[i] = 0.f;
      x[i] = -INFINITY;
    }
  }
  mx = ssc_wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXR_PER_LANE; ++i) {
    x[i] = (lane + 64 * i < R) ? expf(x[i] - mx) : 0.f;
    s += x[i];
  }
  s = ssc_wave_sum(s);
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < MAXR_PER_LANE; ++i) {
    x[i] = x[i] / s * mk[i];
    ps += x[i];
  }
  ps = ssc_wave_sum(ps);
#pragma unroll
  for (int i = 0; i < MAXR_PER_LANE; ++i) alpha[i] = x[i] / (ps + 1e-13f);
}

constexpr int MAXR_LANE = 4;  // R <= 256

// One wave per (row, 256-float feature chunk).  (A four-wave form - regions split over the waves of a 256-thread workgroup,
// partial sums through LDS - was measured in round 2: no gain at the train shape, 6.4 us either way, and 93 vs 60 us at the
// decode shape, where the kernel is bound by L2 bandwidth, not latency.)
// Chunks past the F/256 feature chunks (present only when `obj` is given) pool a second per-region tensor obj (nimg,R,D) with the
// same weights: pool[g,:D] = sum_r alpha_r obj_r, the grounded style prior / LSTM conditioning of SENTIMENT_VAE = 2
// (updown_cell.py:160-163); its pad columns D..ldpool-1 are written as zeros (they are K columns of a gate product).
__global__ __launch_bounds__(64) void attn_apply_kernel(const float* __restrict__ logits, const float* __restrict__ mask,
                                                        const float* __restrict__ feats, int G, int R, int F,
                                                        int rows_per_image, float* __restrict__ alpha_out,
                                                        float* __restrict__ att, int ldatt, const float* __restrict__ obj, int D,
                                                        float* __restrict__ pool, int ldpool) {
  __shared__ float sa[64 * MAXR_LANE];
  int g = blockIdx.y, chunk = blockIdx.x;
  int lane = threadIdx.x;
  int img = g / rows_per_image;
  const int fchunks = (F + 255) / 256;
  float al[MAXR_LANE];
  wave_masked_softmax<MAXR_LANE>(logits + (size_t)g * R, mask + (size_t)img * R, R, lane, al);
#pragma unroll
  for (int i = 0; i < MAXR_LANE; ++i) {
    int r = lane + 64 * i;
    if (r < R) {
      sa[r] = al[i];
      if (chunk == 0) alpha_out[(size_t)g * R + r] = al[i];
    }
  }
  __syncthreads();
  if (!att) return;   // weights only (ssc_attn_weights)
  if (chunk >= fchunks) {   // pooled obj columns (rows of D floats: no 16-byte alignment to rely on)
    const float* op = obj + (size_t)img * R * D;
    for (int k = 0; k < 4; ++k) {
      const int dd = (chunk - fchunks) * 256 + k * 64 + lane;
      if (dd < ldpool) {
        float acc = 0.f;
        if (dd < D)
          for (int r = 0; r < R; ++r) acc += sa[r] * op[(size_t)r * D + dd];
        pool[(size_t)g * ldpool + dd] = acc;
      }
    }
    return;
  }
  const float* fp = feats + (size_t)img * R * F;
  int f = chunk * 256 + lane * 4;
  if (((F & 3) == 0) && ssc_aligned16_dev(fp) && ((ldatt & 3) == 0) && ssc_aligned16_dev(att)) {
    if (f < F) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r0 = 0; r0 < R; r0 += 12) {  // 12 region rows in flight (the kernel is latency-bound), summed in region order
        float4 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = *reinterpret_cast<const float4*>(fp + (size_t)min(r0 + u, R - 1) * F + f);
#pragma unroll
        for (int u = 0; u < 12; ++u) {
          const float a = (r0 + u < R) ? sa[r0 + u] : 0.f;
          acc.x += a * v[u].x; acc.y += a * v[u].y; acc.z += a * v[u].z; acc.w += a * v[u].w;
        }
      }
      *reinterpret_cast<float4*>(att + (size_t)g * ldatt + f) = acc;
    }
  } else {
    for (int k = 0; k < 4; ++k) {
      int ff = chunk * 256 + k * 64 + lane;
      if (ff < F) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += sa[r] * fp[(size_t)r * F + ff];
        att[(size_t)g * ldatt + ff] = acc;
      }
    }
  }
}

// out[g, d] = sum_r alpha[g, r] x[img(g), r, d]: the attention-pooled per-region attribute means that SENTIMENT_VAE = 2 uses as
// prior mean and LSTM conditioning (var_updown/var_updown/modules/updown_cell.py:160-163).  D is small (150): thread per (g, d).
__global__ void attn_pool_kernel(const float* __restrict__ alpha, const float* __restrict__ x, int G, int R, int D,
                                 int rows_per_image, float* __restrict__ out, int ldo) {
  const int g = blockIdx.y, d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) {   // the pad columns of a 16-byte padded row (ldo >= r4(D)) are zeroed: the row can then be a K-segment of r4(D) columns
    if (d < ((D + 3) & ~3) && d < ldo) out[(size_t)g * ldo + d] = 0.f;
    return;
  }
  const float* xp = x + (size_t)(g / rows_per_image) * R * D + d;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) acc += alpha[(size_t)g * R + r] * xp[(size_t)r * D];
  out[(size_t)g * ldo + d] = acc;
}

// ---- backward ---------------------------------------------------------------------------------
// dalpha[g,r] = datt[g,:] . feats[g,r,:]
// (+ with obj: dalpha[g,r] += (dpa[g,:D] + dpb[g,:D]) . obj[g,r,:D] - the weights also pooled obj, see attn_apply_kernel)
__global__ __launch_bounds__(256) void attn_dalpha_kernel(const float* __restrict__ datt, int lddatt,
                                                          const float* __restrict__ feats, int G, int R, int F,
                                                          float* __restrict__ dalpha, const float* __restrict__ obj, int D,
                                                          const float* __restrict__ dpa, int lddpa, int Da,
                                                          const float* __restrict__ dpb, int lddpb) {
  int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (wid >= G * R) return;
  int g = wid / R;
  const float* dp = datt + (size_t)g * lddatt;
  const float* fp = feats + (size_t)wid * F;
  float s = 0.f;
  if (((F & 3) == 0) && ((lddatt & 3) == 0) && ssc_aligned16_dev(dp) && ssc_aligned16_dev(fp)) {
    for (int f = lane * 4; f < F; f += 256) {
      float4 a = *reinterpret_cast<const float4*>(dp + f);
      float4 b = *reinterpret_cast<const float4*>(fp + f);
      s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (int f = lane; f < F; f += 64) s += dp[f] * fp[f];
  }
  if (obj) {
    const float* op = obj + (size_t)wid * D;
    // (dpa holds Da <= D columns: the conditioning block of the language LSTMs is the whole pooled vector - "glove" - or its
    // first entry only - "senti_word_net", updown_cell.py:169-172)
    for (int k = lane; k < D; k += 64)
      s += ((k < Da ? dpa[(size_t)g * lddpa + k] : 0.f) + (dpb ? dpb[(size_t)g * lddpb + k] : 0.f)) * op[k];
  }
  s = ssc_wave_sum(s);
  if (lane == 0) dalpha[wid] = s;
}

// dl = alpha * (dalpha - sum alpha*dalpha)   (masked rows have alpha == 0; the 1e-13 term is inert in fp32,
// see DESIGN.md "attention backward").  Then per (g, r, a): u = tanh(q+pv); dpre = dl*wa*(1-u^2);
// dpv_acc += dpre; dq[g,a] = sum_r dpre; dwa_acc[g,a] += sum_r dl*u.
// One 256-thread workgroup per (row g, 256-wide slice of A): wave w takes regions r = w, w+4, ... with its 64 lanes
// on 4 consecutive a each (16 B/lane, coalesced), the four waves' partial sums meet in LDS.
__global__ __launch_bounds__(256) void attn_bwd_apply_kernel(const float* __restrict__ q, int ldq,
                                                             const float* __restrict__ pv, const float* __restrict__ wa,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ dalpha, int G, int R, int A,
                                                             float* __restrict__ dq, int lddq, float* __restrict__ dpv_acc,
                                                             float* __restrict__ dwa_acc) {
  __shared__ float sdl[64 * MAXR_LANE];
  __shared__ float red[2][4][256];
  const int g = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every wave derives c = sum_r alpha*dalpha redundantly (R is small), wave 0 publishes dl
  float c = 0.f;
  for (int r = lane; r < R; r += 64) c += alpha[(size_t)g * R + r] * dalpha[(size_t)g * R + r];
  c = ssc_wave_sum(c);
  if (wave == 0)
    for (int r = lane; r < R; r += 64) sdl[r] = alpha[(size_t)g * R + r] * (dalpha[(size_t)g * R + r] - c);
  __syncthreads();
  const int a0 = chunk * 256 + lane * 4;
  float dqa[4] = {0.f, 0.f, 0.f, 0.f}, dw[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = ((A & 3) == 0) && ((ldq & 3) == 0) && ssc_aligned16_dev(q) && ssc_aligned16_dev(pv) &&
                   ssc_aligned16_dev(wa) && ssc_aligned16_dev(dpv_acc);
  if (vec) {
    if (a0 < A) {
      const float4 qv = *reinterpret_cast<const float4*>(q + (size_t)g * ldq + a0);
      const float4 wv = *reinterpret_cast<const float4*>(wa + a0);
      const float qa[4] = {qv.x, qv.y, qv.z, qv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
      for (int rb = wave; rb < R; rb += 4 * 5) {  // 5 of this wave's regions per batch: 10 loads in flight, region order kept
        float4 p4[5], ac[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const size_t off = ((size_t)g * R + min(rb + 4 * i, R - 1)) * A + a0;
          p4[i] = *reinterpret_cast<const float4*>(pv + off);
          ac[i] = *reinterpret_cast<const float4*>(dpv_acc + off);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int r = rb + 4 * i;
          if (r < R) {
            const float pp[4] = {p4[i].x, p4[i].y, p4[i].z, p4[i].w};
            float dp[4];
            const float dl = sdl[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float u = ssc_tanh_fast(qa[k] + pp[k]);
              dp[k] = dl * ww[k] * (1.f - u * u);
              dqa[k] += dp[k];
              dw[k] += dl * u;
            }
            float4 acc = ac[i];
            acc.x += dp[0]; acc.y += dp[1]; acc.z += dp[2]; acc.w += dp[3];
            *reinterpret_cast<float4*>(dpv_acc + ((size_t)g * R + r) * A + a0) = acc;
          }
        }
      }
    }
  } else {
    for (int k = 0; k < 4; ++k) {
      int a = chunk * 256 + k * 64 + lane;  // scalar path: lane-contiguous a
      if (a >= A) continue;
      float qa = q[(size_t)g * ldq + a], w = wa[a];
      for (int r = wave; r < R; r += 4) {
        size_t off = ((size_t)g * R + r) * A + a;
        float u = ssc_tanh_fast(qa + pv[off]);
        float dl = sdl[r];
        float dpre = dl * w * (1.f - u * u);
        dpv_acc[off] += dpre;
        dqa[k] += dpre;
        dw[k] += dl * u;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][wave][lane * 4 + k] = dqa[k];
    red[1][wave][lane * 4 + k] = dw[k];
  }
  __syncthreads();
  // thread tid finalises slot tid of the 256-wide slice; slot -> a depends on the path's lane mapping
  {
    float s0 = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
    float s1 = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
    int l = tid >> 2, k = tid & 3;
    int a = vec ? (chunk * 256 + l * 4 + k) : (chunk * 256 + k * 64 + l);
    if (a < A) {
      dq[(size_t)g * lddq + a] = s0;
      dwa_acc[(size_t)g * A + a] += s1;
    }
  }
}

}  // namespace

extern "C" int ssc_attn_logits(const float* q, int ldq, const float* pv, const float* wa, int G, int R, int A,
                               int rows_per_image, float* logits, void* stream) {
  if (!q || !pv || !wa || !logits || G <= 0 || R <= 0 || A <= 0 || A > ATTN_MAXA || rows_per_image <= 0 || ldq < A) return SSC_EINVAL;
  SSC_LAUNCH(attn_logits_kernel, dim3(G * ssc_cdiv(R, ATTN_RG)), dim3(256), 0, (hipStream_t)stream, q, ldq, 1, (size_t)0,
                     (float*)nullptr, 0, pv, wa, G, R, A, rows_per_image, logits, (const int*)nullptr, (const int*)nullptr);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// ssc_attn_weights over a device-side row list (internal: the decode step's live rows); rows outside the list keep stale logits / weights
int ssc_attn_weights_rows(const float* q, int ldq, const float* pv, const float* wa, const float* mask, int G, int R, int A,
                          int rows_per_image, float* logits, float* alpha, const int* rows, const int* row_count, hipStream_t st) {
  if (!q || !pv || !wa || !mask || !alpha || !logits || !rows || !row_count || G <= 0 || R <= 0 || A <= 0 || A > ATTN_MAXA ||
      rows_per_image <= 0 || ldq < A || R > 64 * MAXR_LANE)
    return SSC_EINVAL;
  SSC_LAUNCH(attn_logits_kernel, dim3(G * ssc_cdiv(R, ATTN_RG)), dim3(256), 0, st, q, ldq, 1, (size_t)0, (float*)nullptr, 0, pv, wa,
             G, R, A, rows_per_image, logits, rows, row_count);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(attn_apply_kernel, dim3(1, G), dim3(64), 0, st, logits, mask, (const float*)nullptr, G, R, 4, rows_per_image, alpha,
             (float*)nullptr, 0, (const float*)nullptr, 0, (float*)nullptr, 0);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_attn_fwd(const float* q, int ldq, const float* pv, const float* wa, const float* mask,
                            const float* feats, int G, int R, int A, int F, int rows_per_image, float* logits,
                            float* alpha, float* att, int ldatt, void* stream) {
  if (!mask || !feats || !alpha || !att || !logits || F <= 0 || ldatt < F) return SSC_EINVAL;
  if (R > 64 * MAXR_LANE) return SSC_EINVAL;
  SSC_TRY(ssc_attn_logits(q, ldq, pv, wa, G, R, A, rows_per_image, logits, stream));
  SSC_LAUNCH(attn_apply_kernel, dim3(ssc_cdiv(F, 256), G), dim3(64), 0, (hipStream_t)stream, logits, mask, feats,
                     G, R, F, rows_per_image, alpha, att, ldatt, (const float*)nullptr, 0, (float*)nullptr, 0);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_attn_weights(const float* q, int ldq, const float* pv, const float* wa, const float* mask, int G, int R, int A,
                                int rows_per_image, float* logits, float* alpha, void* stream) {
  if (!mask || !alpha || !logits || R > 64 * MAXR_LANE) return SSC_EINVAL;
  SSC_TRY(ssc_attn_logits(q, ldq, pv, wa, G, R, A, rows_per_image, logits, stream));
  SSC_LAUNCH(attn_apply_kernel, dim3(1, G), dim3(64), 0, (hipStream_t)stream, logits, mask, (const float*)nullptr, G, R, 4,
             rows_per_image, alpha, (float*)nullptr, 0, (const float*)nullptr, 0, (float*)nullptr, 0);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_attn_fwd_pool(const float* q, int ldq, const float* pv, const float* wa, const float* mask,
                                 const float* feats, int G, int R, int A, int F, int rows_per_image, float* logits,
                                 float* alpha, float* att, int ldatt, const float* obj, int D, float* pool, int ldpool,
                                 void* stream) {
  if (!mask || !feats || !alpha || !att || !logits || F <= 0 || ldatt < F) return SSC_EINVAL;
  if (!obj || !pool || D <= 0 || ldpool < D || R > 64 * MAXR_LANE) return SSC_EINVAL;
  SSC_TRY(ssc_attn_logits(q, ldq, pv, wa, G, R, A, rows_per_image, logits, stream));
  SSC_LAUNCH(attn_apply_kernel, dim3(ssc_cdiv(F, 256) + ssc_cdiv(ldpool, 256), G), dim3(64), 0, (hipStream_t)stream, logits, mask,
             feats, G, R, F, rows_per_image, alpha, att, ldatt, obj, D, pool, ldpool);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_attn_pool(const float* alpha, const float* x, int G, int R, int D, int rows_per_image, float* out, int ldo,
                             void* stream) {
  if (!alpha || !x || !out || G <= 0 || R <= 0 || D <= 0 || rows_per_image <= 0 || ldo < D) return SSC_EINVAL;
  SSC_LAUNCH(attn_pool_kernel, dim3(ssc_cdiv((D + 3) & ~3, 128), G), dim3(128), 0, (hipStream_t)stream, alpha, x, G, R, D, rows_per_image,
             out, ldo);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// internal: attention forward with q given as split-K slabs ((G,A) each, ld A); writes the reduced q to q_out (ld ldqo)
int ssc_attn_fwd_qslabs(const float* qslabs, int nslab, size_t slab_stride, float* q_out, int ldqo, const float* pv,
                        const float* wa, const float* mask, const float* feats, int G, int R, int A, int F,
                        int rows_per_image, float* logits, float* alpha, float* att, int ldatt, hipStream_t st,
                        const float* obj, int D, float* pool, int ldpool) {
  if (!qslabs || nslab < 1 || !q_out || !pv || !wa || !mask || !feats || !alpha || !att || !logits) return SSC_EINVAL;
  if (obj && (!pool || D <= 0 || ldpool < D)) return SSC_EINVAL;
  if (G <= 0 || R <= 0 || A <= 0 || A > ATTN_MAXA || F <= 0 || ldatt < F || ldqo < A || R > 64 * MAXR_LANE) return SSC_EINVAL;
  SSC_LAUNCH(attn_logits_kernel, dim3(G * ssc_cdiv(R, ATTN_RG)), dim3(256), 0, st, qslabs, A, nslab, slab_stride, q_out,
                     ldqo, pv, wa, G, R, A, rows_per_image, logits, (const int*)nullptr, (const int*)nullptr);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(attn_apply_kernel, dim3(ssc_cdiv(F, 256) + (obj ? ssc_cdiv(ldpool, 256) : 0), G), dim3(64), 0, st, logits, mask, feats,
             G, R, F, rows_per_image, alpha, att, ldatt, obj, D, pool, ldpool);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_attn_bwd(const float* datt, int lddatt, const float* q, int ldq, const float* pv, const float* wa,
                            const float* alpha, const float* feats, int G, int R, int A, int F, float* dq, int lddq,
                            float* dpv_acc, float* dwa_acc, float* scratch_dalpha, void* stream) {
  return ssc_attn_bwd_pool(datt, lddatt, q, ldq, pv, wa, alpha, feats, G, R, A, F, dq, lddq, dpv_acc, dwa_acc, scratch_dalpha,
                           nullptr, 0, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int ssc_attn_bwd_pool(const float* datt, int lddatt, const float* q, int ldq, const float* pv, const float* wa,
                                 const float* alpha, const float* feats, int G, int R, int A, int F, float* dq, int lddq,
                                 float* dpv_acc, float* dwa_acc, float* scratch_dalpha, const float* obj, int D,
                                 const float* dpool_a, int lddpa, int Da, const float* dpool_b, int lddpb, void* stream) {
  if (!datt || !q || !pv || !wa || !alpha || !feats || !dq || !dpv_acc || !dwa_acc || !scratch_dalpha) return SSC_EINVAL;
  if (G <= 0 || R <= 0 || A <= 0 || F <= 0 || R > 64 * MAXR_LANE || lddatt < F || ldq < A || lddq < A) return SSC_EINVAL;
  if (obj && (!dpool_a || D <= 0 || Da < 0 || Da > D || lddpa < Da || (dpool_b && lddpb < D))) return SSC_EINVAL;
  SSC_LAUNCH(attn_dalpha_kernel, dim3(ssc_cdiv(G * R, 4)), dim3(256), 0, (hipStream_t)stream, datt, lddatt, feats,
                     G, R, F, scratch_dalpha, obj, D, dpool_a, lddpa, Da, dpool_b, lddpb);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(attn_bwd_apply_kernel, dim3(ssc_cdiv(A, 256), G), dim3(256), 0, (hipStream_t)stream, q, ldq, pv, wa,
                     alpha, scratch_dalpha, G, R, A, dq, lddq, dpv_acc, dwa_acc);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

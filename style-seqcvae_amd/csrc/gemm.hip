// Exact-fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), segmented over K.
//
//   C[M,N] (+)= sum_s op(A_s)[M,K_s] * op(B_s)[K_s,N] (+ bias)
//
// Replaces aten::mm/addmm under nn.LSTMCell / nn.Linear and their backward on the var_updown hot
// path (reference: var_updown/var_updown/modules/updown_cell.py:146,192,196-197,227;
// updown-baseline/updown/modules/attention.py:69,125; updown_captioner.py:444-445).  Segments let
// the torch.cat inputs of the three LSTM cells (updown_cell.py:143,178,211) stay un-materialised.
//
// Design (CDNA4): 64x64 output tile per 256-thread workgroup = 2x2 waves, one 32x32 MFMA tile each
// (16 accumulator VGPRs); BK = 32 per stage, double-buffered LDS, register prefetch of the next
// stage while the current one is consumed.  Operands are staged in one of two LDS images:
//   KC (operand is k-contiguous in HBM): [row][BK+4] floats, fragments by ds_read_b128 - a lane
//       takes 4 consecutive k, the lane half (l>>5) picks k-quad 0/1 of an 8-wide chunk, so MFMA j
//       of a chunk contracts k = {j, 4+j}; the +4 pad makes the 16-lane b128 groups conflict-free.
//   MC (operand is m/n-contiguous in HBM): [BK][64] floats, fragments by ds_read_b32 with the
//       same k assignment, 32 consecutive floats per half-wave (conflict-free).
// so NT / NN / TN products need no transposing LDS writes.  Global loads are 16 B/lane when base
// and leading dimension allow it, else 4 B/lane (still coalesced along the contiguous axis).
// Split-K: grid.z workgroups per tile write partial slabs that a second kernel (or the fused LSTM /
// latent epilogue kernels) sums in a fixed order -> deterministic, no float atomics.
#include "ssc_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int KC_LD = BK + 4;  // 36 floats: 144 B rows, 16-B aligned, conflict-free b128 reads
constexpr int MC_LD = 64;
constexpr int TILE_FLOATS = 64 * KC_LD;  // 2304 >= 32*64

struct KSeg {
  const float* A;
  const float* B;
  int lda, ldb, K, nsteps;
  int avec, bvec;  // 16 B/lane global loads allowed for this segment's A / B operand
};

struct KArgs {
  KSeg seg[SSC_MAX_SEG];
  int nseg;
  int M, N;
  float* out;  // C (splits==1 && direct) or slab base
  int ldo;
  size_t slab_stride;  // floats between slabs (0 when direct)
  const float* bias;
  int accumulate;
  int steps_total, steps_per_split;
};

// ---- global -> register staging of one 64 x 32 operand tile (8 floats per thread) ------------------
template <bool KC>
__device__ __forceinline__ void load_tile(float (&r)[8], const float* __restrict__ base, int ld, int rows, int K,
                                          int row0, int k0, int tid, bool VEC) {
  if (KC && VEC) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + 256 * u;
      int row = idx >> 3, kq = idx & 7;
      int grow = row0 + row, k = k0 + 4 * kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < rows) {
        const float* p = base + (size_t)grow * ld + k;
        if (k + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
        }
      }
      r[4 * u] = v.x; r[4 * u + 1] = v.y; r[4 * u + 2] = v.z; r[4 * u + 3] = v.w;
    }
  } else if (KC && !VEC) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int idx = tid + 256 * u;
      int row = idx >> 5, kk = idx & 31;
      int grow = row0 + row, k = k0 + kk;
      r[u] = (grow < rows && k < K) ? base[(size_t)grow * ld + k] : 0.f;
    }
  } else if (!KC && VEC) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + 256 * u;
      int kk = idx >> 4, mq = idx & 15;
      int gk = k0 + kk, gm = row0 + 4 * mq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < K) {
        const float* p = base + (size_t)gk * ld + gm;
        if (gm + 3 < rows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gm < rows) v.x = p[0];
          if (gm + 1 < rows) v.y = p[1];
          if (gm + 2 < rows) v.z = p[2];
        }
      }
      r[4 * u] = v.x; r[4 * u + 1] = v.y; r[4 * u + 2] = v.z; r[4 * u + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int idx = tid + 256 * u;
      int kk = idx >> 6, m = idx & 63;
      int gk = k0 + kk, gm = row0 + m;
      r[u] = (gk < K && gm < rows) ? base[(size_t)gk * ld + gm] : 0.f;
    }
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float (&r)[8], int tid, bool VEC) {
  if (KC && VEC) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + 256 * u;
      int row = idx >> 3, kq = idx & 7;
      *reinterpret_cast<float4*>(&s[row * KC_LD + 4 * kq]) = make_float4(r[4 * u], r[4 * u + 1], r[4 * u + 2], r[4 * u + 3]);
    }
  } else if (KC && !VEC) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int idx = tid + 256 * u;
      s[(idx >> 5) * KC_LD + (idx & 31)] = r[u];
    }
  } else if (!KC && VEC) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + 256 * u;
      int kk = idx >> 4, mq = idx & 15;
      *reinterpret_cast<float4*>(&s[kk * MC_LD + 4 * mq]) = make_float4(r[4 * u], r[4 * u + 1], r[4 * u + 2], r[4 * u + 3]);
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int idx = tid + 256 * u;
      s[(idx >> 6) * MC_LD + (idx & 63)] = r[u];
    }
  }
}

// fragment of 4 k-values (k = chunk*8 + 4*half + j) for tile row/col `rc`
template <bool KC>
__device__ __forceinline__ void read_frag(float (&f)[4], const float* __restrict__ s, int rc, int chunk, int half) {
  if constexpr (KC) {
    float4 v = *reinterpret_cast<const float4*>(&s[rc * KC_LD + chunk * 8 + 4 * half]);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = s[(chunk * 8 + 4 * half + j) * MC_LD + rc];
  }
}

__device__ __forceinline__ void locate(const KArgs& a, int step, int& seg, int& k0) {
  int rem = step;
  seg = 0;
#pragma unroll 1
  for (int i = 0; i < a.nseg; ++i) {
    if (rem < a.seg[i].nsteps) { seg = i; break; }
    rem -= a.seg[i].nsteps;
  }
  k0 = rem * BK;
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(const KArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * TILE_FLOATS];  // As[2], Bs[2]
  float* As = lds;
  float* Bs = lds + 2 * TILE_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, z = blockIdx.z;

  const int s_lo = z * a.steps_per_split;
  int s_hi = s_lo + a.steps_per_split;
  if (s_hi > a.steps_total) s_hi = a.steps_total;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float ra[8], rb[8];
  bool va = false, vb = false;
  if (s_lo < s_hi) {
    int sg, k0;
    locate(a, s_lo, sg, k0);
    va = a.seg[sg].avec != 0;
    vb = a.seg[sg].bvec != 0;
    load_tile<A_KC>(ra, a.seg[sg].A, a.seg[sg].lda, a.M, a.seg[sg].K, m0, k0, tid, va);
    load_tile<B_KC>(rb, a.seg[sg].B, a.seg[sg].ldb, a.N, a.seg[sg].K, n0, k0, tid, vb);
    store_tile<A_KC>(As, ra, tid, va);
    store_tile<B_KC>(Bs, rb, tid, vb);
  }
  __syncthreads();

  for (int s = s_lo; s < s_hi; ++s) {
    const int cur = (s - s_lo) & 1;
    const bool more = (s + 1) < s_hi;
    if (more) {
      int sg, k0;
      locate(a, s + 1, sg, k0);
      va = a.seg[sg].avec != 0;
      vb = a.seg[sg].bvec != 0;
      load_tile<A_KC>(ra, a.seg[sg].A, a.seg[sg].lda, a.M, a.seg[sg].K, m0, k0, tid, va);
      load_tile<B_KC>(rb, a.seg[sg].B, a.seg[sg].ldb, a.N, a.seg[sg].K, n0, k0, tid, vb);
    }
    const float* as = As + cur * TILE_FLOATS;
    const float* bs = Bs + cur * TILE_FLOATS;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float fa[4], fb[4];
      read_frag<A_KC>(fa, as, wm * 32 + l31, c, half);
      read_frag<B_KC>(fb, bs, wn * 32 + l31, c, half);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    }
    if (more) {
      store_tile<A_KC>(As + (cur ^ 1) * TILE_FLOATS, ra, tid, va);
      store_tile<B_KC>(Bs + (cur ^ 1) * TILE_FLOATS, rb, tid, vb);
    }
    __syncthreads();
  }

  // epilogue: acc[r] -> row (r&3) + 8*(r>>2) + 4*half, col l31 of the wave's 32x32 tile
  float* out = a.out + (size_t)z * a.slab_stride;
  const int col = n0 + wn * 32 + l31;
  if (col < a.N) {
    const float bv = (a.bias != nullptr) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < a.M) {
        float* p = out + (size_t)row * a.ldo + col;
        float v = acc[r] + bv;
        if (a.accumulate) v += *p;
        *p = v;
      }
    }
  }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N,
                                    float* __restrict__ C, int ldc, const float* __restrict__ bias, int accumulate) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)M * N;
  if (i >= total) return;
  int row = (int)(i / N), col = (int)(i % N);
  float v = 0.f;
  for (int s = 0; s < nslab; ++s) v += slabs[(size_t)s * slab_stride + i];
  if (bias) v += bias[col];
  float* p = C + (size_t)row * ldc + col;
  if (accumulate) v += *p;
  *p = v;
}

typedef void (*gemm_fn)(const KArgs);

// ---- optional in-situ profiling (bench.py roofline leg): hipEvent pair around every GEMM launch ------
struct ProfRec {
  hipEvent_t e0, e1;
  int kind, M, N, K, splits;
};
constexpr int PROF_MAX = 4096;
ProfRec* g_prof = nullptr;
int g_prof_n = 0;
bool g_prof_on = false;

int build_args(const ssc_gemm_desc* d, KArgs& k) {
  if (!d || d->nseg < 1 || d->nseg > SSC_MAX_SEG || d->M <= 0 || d->N <= 0) return SSC_EINVAL;
  k.nseg = d->nseg;
  k.M = d->M;
  k.N = d->N;
  k.steps_total = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const ssc_gemm_seg& s = d->seg[i];
    if (!s.A || !s.B || s.K <= 0) return SSC_EINVAL;
    int min_lda = d->a_kc ? s.K : d->M, min_ldb = d->b_kc ? s.K : d->N;
    if (s.lda < min_lda || s.ldb < min_ldb) return SSC_EINVAL;
    if ((((uintptr_t)s.A) & 3u) || (((uintptr_t)s.B) & 3u)) return SSC_EALIGN;
    k.seg[i].A = s.A;
    k.seg[i].B = s.B;
    k.seg[i].lda = s.lda;
    k.seg[i].ldb = s.ldb;
    k.seg[i].K = s.K;
    k.seg[i].nsteps = ssc_cdiv(s.K, BK);
    k.steps_total += k.seg[i].nsteps;
    k.seg[i].avec = (ssc_aligned16(s.A) && !(s.lda & 3)) ? 1 : 0;
    k.seg[i].bvec = (ssc_aligned16(s.B) && !(s.ldb & 3)) ? 1 : 0;
  }
  return SSC_OK;
}

int launch(const ssc_gemm_desc* d, KArgs& k, int splits, hipStream_t st) {
  k.steps_per_split = ssc_cdiv(k.steps_total, splits);
  dim3 grid(ssc_cdiv(d->N, BN), ssc_cdiv(d->M, BM), splits);
  gemm_fn fn;
  if (d->a_kc && d->b_kc) fn = gemm_kernel<true, true>;
  else if (d->a_kc && !d->b_kc) fn = gemm_kernel<true, false>;
  else if (!d->a_kc && !d->b_kc) fn = gemm_kernel<false, false>;
  else fn = gemm_kernel<false, true>;
  ProfRec* rec = nullptr;
  if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {
    rec = &g_prof[g_prof_n++];
    rec->kind = (d->a_kc ? 0 : 2) + (d->b_kc ? 0 : 1);  // 0 NT, 1 NN, 3 TN
    rec->M = d->M; rec->N = d->N; rec->splits = splits;
    rec->K = 0;
    for (int i = 0; i < d->nseg; ++i) rec->K += d->seg[i].K;
    (void)hipEventRecord(rec->e0, st);
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), 0, st, k);
  if (rec) (void)hipEventRecord(rec->e1, st);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

}  // namespace

extern "C" int ssc_gemm_auto_splits(int M, int N, int ksteps) {
  long tiles = (long)ssc_cdiv(M, BM) * ssc_cdiv(N, BN);
  if (tiles >= 384) return 1;
  int s = (int)((640 + tiles - 1) / tiles);
  int maxs = ksteps / 6;  // keep >= 6 k-steps (192 of K) per workgroup
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  // no empty trailing split
  int per = ssc_cdiv(ksteps, s);
  s = ssc_cdiv(ksteps, per);
  return s;
}

// partial slabs only: slabs[z] is (M,N) with ld N.  Used by the fused epilogue kernels.
int ssc_gemm_slabs(const ssc_gemm_desc* d, int splits, float* slabs, hipStream_t st) {
  KArgs k;
  SSC_TRY(build_args(d, k));
  if (splits < 1 || !slabs) return SSC_EINVAL;
  int per = ssc_cdiv(k.steps_total, splits);
  if (ssc_cdiv(k.steps_total, per) != splits) return SSC_EINVAL;
  k.out = slabs;
  k.ldo = d->N;
  k.slab_stride = (size_t)d->M * d->N;
  k.bias = nullptr;
  k.accumulate = 0;
  return launch(d, k, splits, st);
}

extern "C" int ssc_gemm(const ssc_gemm_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  KArgs k;
  SSC_TRY(build_args(d, k));
  if (!d->C || d->ldc < d->N) return SSC_EINVAL;
  int splits = d->splits;
  if (splits <= 0) splits = ssc_gemm_auto_splits(d->M, d->N, k.steps_total);
  if (splits > k.steps_total) splits = k.steps_total;
  {
    int per = ssc_cdiv(k.steps_total, splits);
    splits = ssc_cdiv(k.steps_total, per);
  }
  if (splits > 1 && (!d->workspace || d->workspace_floats < (size_t)splits * d->M * d->N)) {
    if (d->splits > 1) return SSC_EWORKSPACE;
    splits = 1;  // auto mode without (enough) workspace: fall back to one pass
  }
  if (splits == 1) {
    k.out = d->C;
    k.ldo = d->ldc;
    k.slab_stride = 0;
    k.bias = d->bias;
    k.accumulate = d->accumulate;
    return launch(d, k, 1, st);
  }
  k.out = d->workspace;
  k.ldo = d->N;
  k.slab_stride = (size_t)d->M * d->N;
  k.bias = nullptr;
  k.accumulate = 0;
  SSC_TRY(launch(d, k, splits, st));
  size_t total = (size_t)d->M * d->N;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d->workspace, splits,
                     k.slab_stride, d->M, d->N, d->C, d->ldc, d->bias, d->accumulate);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// ---- profiling control (process-global, not thread-safe; used by bench.py only) ------------------------
extern "C" int ssc_prof_enable(int on) {
  if (on && !g_prof) {
    g_prof = new ProfRec[PROF_MAX];
    for (int i = 0; i < PROF_MAX; ++i) {
      if (hipEventCreate(&g_prof[i].e0) != hipSuccess || hipEventCreate(&g_prof[i].e1) != hipSuccess) return SSC_EHIP;
    }
  }
  g_prof_on = on != 0;
  if (on) g_prof_n = 0;
  return SSC_OK;
}

// out: n records x 6 floats {kind, M, N, K, splits, milliseconds}; returns the record count (<= max_records)
extern "C" int ssc_prof_collect(float* out, int max_records) {
  if (!out || !g_prof) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return SSC_EHIP;
  int n = g_prof_n < max_records ? g_prof_n : max_records;
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_prof[i].e0, g_prof[i].e1);
    float* o = out + (size_t)i * 6;
    o[0] = (float)g_prof[i].kind; o[1] = (float)g_prof[i].M; o[2] = (float)g_prof[i].N; o[3] = (float)g_prof[i].K;
    o[4] = (float)g_prof[i].splits; o[5] = ms;
  }
  g_prof_n = 0;
  return n;
}

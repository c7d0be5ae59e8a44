// Internal helpers shared by the HIP translation units of libssc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ssc.h"

extern thread_local int ssc_tls_hip_error;

// Every kernel launch goes through SSC_LAUNCH: the thread's sticky HIP error is cleared first, so that SSC_CHECK_LAUNCH
// reports THIS launch and not an error some earlier, unrelated HIP call of the process left behind (seen: hipError 100
// from the host framework's device probing made the first launch of a process "fail").
#define SSC_LAUNCH(...)              \
  do {                               \
    (void)hipGetLastError();         \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define SSC_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) {                       \
      ssc_tls_hip_error = (int)e__;                \
      return SSC_EHIP;                             \
    }                                              \
  } while (0)

#define SSC_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != SSC_OK) return rc__; \
  } while (0)

// Tuning / diagnostic defaults may be overridden from the environment ONLY in a process that opts in with SSC_DEBUG=1: a stray
// SSC_* variable must not change which kernels the product path runs.  Tools (tools/*.py) set SSC_DEBUG=1.
#include <stdlib.h>
#include <string.h>
static inline bool ssc_env_debug() { const char* d = getenv("SSC_DEBUG"); return d && d[0] == '1'; }
static inline int ssc_env_int(const char* name, int dflt) {
  if (!ssc_env_debug()) return dflt;
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

static inline bool ssc_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int ssc_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t ssc_round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

#define SSC_WAVE 64
__device__ __forceinline__ bool ssc_aligned16_dev(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ float ssc_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ssc_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float ssc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh x = 1 - 2 / (1 + e^{2x}) on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute error < 3e-7 everywhere,
// exact limits (+-1) for large |x|.  Used where a kernel is bound by its tanh count (attention logits: G*R*A per step - 138 M
// at the decode shape, ~25 instructions each with the library tanhf); forward and backward of the attention use the SAME function.
__device__ __forceinline__ float ssc_tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^{2x}
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

// 2xFP16 pieces (ssc_model_cfg.gemm_mode 3): hi = x truncated to fp16, lo = (x - hi) truncated to fp16 (x - hi is exact in fp32) -
// 21-22 significant bits of x; two v_cvt_pkrtz + two conversions back + two subtractions per pair.  |x| must stay below 65504.
// ONE definition for the product kernels' producers, ssc_split_f16 and the cells that leave their output already split.
typedef float ssc_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ssc_u32x2 __attribute__((ext_vector_type(2)));
typedef __fp16 ssc_h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ssc_split4_f16(const ssc_f32x4& v, ssc_u32x2& hi, ssc_u32x2& lo) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const ssc_h16x2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * j], v[2 * j + 1]);
    const float r0 = v[2 * j] - (float)h[0], r1 = v[2 * j + 1] - (float)h[1];
    const ssc_h16x2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi[j] = __builtin_bit_cast(unsigned, h);
    lo[j] = __builtin_bit_cast(unsigned, l);
  }
}
// one value: the two halfs as raw bits (the same conversions, element by element)
__device__ __forceinline__ void ssc_split1_f16(float x, unsigned short& hi, unsigned short& lo) {
  const ssc_h16x2 h = __builtin_amdgcn_cvt_pkrtz(x, 0.f);
  const ssc_h16x2 l = __builtin_amdgcn_cvt_pkrtz(x - (float)h[0], 0.f);
  hi = (unsigned short)(__builtin_bit_cast(unsigned, h) & 0xffffu);
  lo = (unsigned short)(__builtin_bit_cast(unsigned, l) & 0xffffu);
}
// plane layout (ssc_gemm_seg.A16): word offset of the hi half-pair of columns k, k + 1 (k even) inside a row; the lo pair is 16 words on
__device__ __forceinline__ int ssc_plane_word(int k) { return (k >> 5) * 32 + ((k & 31) >> 1); }

// internal cross-TU entry points (not part of the C ABI)
int ssc_gemm_slabs(const ssc_gemm_desc* d, int splits, float* slabs, hipStream_t st);  // partial slabs only
int ssc_attn_fwd_qslabs(const float* qslabs, int nslab, size_t slab_stride, float* q_out, int ldqo, const float* pv,
                        const float* wa, const float* mask, const float* feats, int G, int R, int A, int F,
                        int rows_per_image, float* logits, float* alpha, float* att, int ldatt, hipStream_t st,
                        const float* obj = nullptr, int D = 0, float* pool = nullptr, int ldpool = 0);
int ssc_gemm_slabs_auto(const ssc_gemm_desc* d, float* slabs, size_t cap_floats, int* nslab, hipStream_t st);
// n <= 3 independent minibatch products (M <= 64, same operand layout) in ONE launch, each left as split-K slabs in its
// own region; falls back to one launch per product when they do not qualify for the 64x256 wave-specialised kernel
int ssc_gemm_slabs_group(const ssc_gemm_desc* const* d, int n, float* const* regions, const size_t* caps, int* nslab,
                         hipStream_t st);
// the independent weight-gradient products of one backward phase (C_i = A_i^T B_i, direct outputs) as grouped launches
int ssc_gemm_dw_group(const ssc_gemm_desc* const* d, int n, hipStream_t st);
// C (+)= sum of split-K slabs (M x N, ld N) in index order (+ bias)
int ssc_reduce_slabs(const float* slabs, int nslab, size_t stride, int M, int N, float* C, int ldc, const float* bias,
                     int accumulate, hipStream_t st);

// beam-search stages shared by decode.hip and fsm.hip
int ssc_beam_first_dense(bool norm, const float* lp, int ldlp, const uint8_t* fsm, const int* mach, int B, int S, int V, int beam,
                         int64_t* pred, float* lp_out, hipStream_t st);
int ssc_beam_rows_dense(bool norm, const float* lp, int ldlp, const uint8_t* fsm, const int* mach, const int64_t* last_pred, int B,
                        int S, int V, int beam, int per_node, int end_index, float* scratch_val, int64_t* scratch_idx,
                        hipStream_t st);
int ssc_beam_merge(const float* sval, const int64_t* sidx, const float* last_lp, int B, int S, int beam, int per_node,
                   int64_t* pred, float* lp_out, int64_t* backptr, int end_index, int* ctl, int step_index, int max_steps,
                   int* host_flag, hipStream_t st);
int ssc_decode_att_table_enabled();   // the "dec_att_table" switch (decode.hip)

// numerics mode of the calling thread's current sequence-level call (ssc_model_cfg.gemm_mode: 0 = the process default set by
// ssc_set_gemm_mode, 1 = 3xBF16, 2 = exact-fp32 MFMA); -1 = none in force
extern thread_local int ssc_tls_gemm_mode;
extern thread_local int ssc_tls_gemm_f16;
struct SscGemmModeScope {
  int prev, prev16;
  explicit SscGemmModeScope(const ssc_model_cfg* c) : prev(ssc_tls_gemm_mode), prev16(ssc_tls_gemm_f16) {
    if (c && (c->gemm_mode == 1 || c->gemm_mode == 3)) ssc_tls_gemm_mode = 1;
    else if (c && c->gemm_mode == 2) ssc_tls_gemm_mode = 0;
    if (c && c->gemm_mode != 0) ssc_tls_gemm_f16 = c->gemm_mode == 3 ? 1 : 0;   // (0: the process default stays in force)
  }
  ~SscGemmModeScope() { ssc_tls_gemm_mode = prev; ssc_tls_gemm_f16 = prev16; }
};
int ssc_decode_parts_enabled();        // the "dec_parts" switch (decode.hip): vocabulary head records instead of logits
int ssc_attn_weights_rows(const float* q, int ldq, const float* pv, const float* wa, const float* mask, int G, int R, int A,
                          int rows_per_image, float* logits, float* alpha, const int* rows, const int* row_count, hipStream_t st);

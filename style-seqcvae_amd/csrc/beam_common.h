// Selection helpers shared by the beam-search kernels (decode.hip, fsm.hip).  Selection order everywhere: value descending,
// index ascending ("k-pass selection": pass k finds the best candidate strictly after the previous pick in that order).
#pragma once
#include "ssc_common.h"

namespace {

struct Cand {
  float v;
  int i;
};
__device__ __forceinline__ bool better(float v, int i, const Cand& o) {  // (v,i) ranks before o
  return (v > o.v) || (v == o.v && i < o.i);
}
__device__ __forceinline__ bool after(float v, int i, const Cand& prev) {  // (v,i) ranks strictly after the previous pick
  return (prev.i < 0) || (v < prev.v) || (v == prev.v && i > prev.i);
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Cand t;
    t.v = __shfl_xor(c.v, o, 64);
    t.i = __shfl_xor(c.i, o, 64);
    if (t.i >= 0 && (c.i < 0 || better(t.v, t.i, c))) c = t;
  }
  return c;
}
__device__ __forceinline__ Cand block_best(Cand c, Cand* sh) {
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  c = wave_best(c);
  __syncthreads();
  if (lane == 0) sh[wv] = c;
  __syncthreads();
  Cand r = sh[0];
  for (int k = 1; k < nw; ++k)
    if (sh[k].i >= 0 && (r.i < 0 || better(sh[k].v, sh[k].i, r))) r = sh[k];
  return r;
}

// One-barrier forms for kernels that run several reductions in a row: consecutive reductions alternate between two LDS slots
// (`par` = 0, 1, 0, ...), so the write of reduction n + 1 cannot overtake a slow wave's read of reduction n - 1's slot (there is a
// barrier of reduction n in between), and the leading barrier of the two-barrier forms is not needed.  Same arithmetic and order.
__device__ __forceinline__ Cand block_best1(Cand c, Cand (*sh)[4], int par) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  c = wave_best(c);
  if (lane == 0) sh[par][wv] = c;
  __syncthreads();
  Cand r = sh[par][0];
  for (int k = 1; k < nw; ++k)
    if (sh[par][k].i >= 0 && (r.i < 0 || better(sh[par][k].v, sh[par][k].i, r))) r = sh[par][k];
  return r;
}
__device__ __forceinline__ float dec_block_reduce1(float v, float (*sh)[4], int par, bool is_max) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = is_max ? ssc_wave_max(v) : ssc_wave_sum(v);
  if (lane == 0) sh[par][wv] = v;
  __syncthreads();
  float r = sh[par][0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, sh[par][i]) : r + sh[par][i];
  return r;
}

// block reduction with exactly the arithmetic of log_softmax_kernel (256 threads, strided partials, this order)
__device__ __forceinline__ float dec_block_reduce(float v, float* sh, bool is_max) {
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = is_max ? ssc_wave_max(v) : ssc_wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

}  // namespace

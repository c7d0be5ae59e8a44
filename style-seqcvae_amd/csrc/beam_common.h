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

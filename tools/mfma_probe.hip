// Diagnostic: what the fp32 / bf16 matrix pipe sustains per CU for the GEMM's inner-loop shape (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: fp32 mfma only; 1: fp32 mfma + ds_read_b128 per 4 mfma (as gemm_kernel); 2: bf16 mfma only; 3: bf16 + ds_read
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2304];
  for (int i = threadIdx.x; i < 4 * 2304; i += 256) lds[i] = 1.0f + (i & 7);
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  int lane = threadIdx.x & 63;
  const float* pa = lds + (lane & 31) * 36 + 4 * (lane >> 5);
  const float* pb = lds + 2 * 2304 + (lane & 31) * 36 + 4 * (lane >> 5);
  float4 a = *(const float4*)pa, b = *(const float4*)pb;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (MODE == 1 || MODE == 3) { a = *(const float4*)(pa + c * 8); b = *(const float4*)(pb + c * 8); }
      if (MODE <= 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      } else {
        bf16x8 x = __builtin_bit_cast(bf16x8, a), y = __builtin_bit_cast(bf16x8, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.f) out[0] = s;
}
template <int MODE>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 3, 4}) {
    int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (MODE <= 1) ? 16.0 : 12.0;                       // mfma per iteration per wave
    double flop = (MODE <= 1) ? 4096.0 : 32768.0;                // per mfma
    double cyc = (MODE <= 1) ? 64.0 : 32.0;
    double waves = 256.0 * wgs_per_cu * 4;
    double tf = waves * iters * mf * flop / (ms * 1e-3) / 1e12;
    // implied clock if the pipe were perfectly busy: busy cycles per SIMD / time
    double busy = wgs_per_cu * iters * mf * cyc;
    printf("%-28s wg/cu %d: %8.1f us  %7.1f TF  pipe-cycles/us = %.0f MHz-equivalent\n", name, wgs_per_cu, ms * 1e3, tf, busy / (ms * 1e3));
  }
}
int main() {
  float* out; hipMalloc(&out, 64);
  run<0>("fp32 mfma only", out);
  run<1>("fp32 mfma + ds_read_b128", out);
  run<2>("bf16 mfma only", out);
  run<3>("bf16 mfma + ds_read_b128", out);
  return 0;
}

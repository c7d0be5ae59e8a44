// Diagnostic: add the GEMM loop's ingredients one at a time to a pure fp32-MFMA loop (1 workgroup per CU, 4 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// FEAT bits: 1 ds_read per 4 mfma, 2 barrier per iteration, 4 four ds_write_b128 per iteration, 8 four global loads per
// iteration consumed by the ds_writes one iteration later (register prefetch), 16 alternate LDS buffers
template <int FEAT>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ g, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2304];
  for (int i = threadIdx.x; i < 4 * 2304; i += 256) lds[i] = 1.0f + (i & 7);
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int tid = threadIdx.x, lane = tid & 63;
  float4 r[4];
  for (int u = 0; u < 4; ++u) r[u] = make_float4(1, 2, 3, 4);
  const float4* gp = g + (size_t)blockIdx.x * 4096 + tid;
  for (int it = 0; it < iters; ++it) {
    const int buf = (FEAT & 16) ? (it & 1) : 0;
    const float* pa = lds + buf * 2304 + (lane & 31) * 36 + 4 * (lane >> 5);
    const float* pb = lds + 2 * 2304 + buf * 2304 + (lane & 31) * 36 + 4 * (lane >> 5);
    float4 a = *(const float4*)pa, b = *(const float4*)pb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (FEAT & 1) { a = *(const float4*)(pa + c * 8); b = *(const float4*)(pb + c * 8); }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    if (FEAT & 4) {
      float* wa = lds + (buf ^ ((FEAT & 16) ? 1 : 0)) * 2304 + (tid >> 3) * 36 + 4 * (tid & 7);
      *(float4*)wa = r[0];
      *(float4*)(wa + 32 * 36) = r[1];
      *(float4*)(wa + 2 * 2304) = r[2];
      *(float4*)(wa + 2 * 2304 + 32 * 36) = r[3];
    }
    if (FEAT & 8) {
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = gp[(size_t)((it * 4 + u) & 7) * 256];
    }
    if (FEAT & 2) __syncthreads();
  }
  float s = r[0].x + r[1].y + r[2].z + r[3].w;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.f) out[0] = s;
}
template <int FEAT>
void run(const char* name, const float4* g, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 3, 4}) {
    int iters = 2000;
    hipLaunchKernelGGL(k<FEAT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, g, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FEAT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = 256.0 * wgs_per_cu * 4;
    double tf = waves * iters * 16.0 * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-44s wg/cu %d: %8.1f us  %6.1f TF (%.0f%%)  %.0f cycles/iter/wg-slot\n", name, wgs_per_cu, ms * 1e3, tf, tf / 1.573,
           ms * 1e-3 / iters / wgs_per_cu * 2.4e9 * wgs_per_cu);
  }
}
int main() {
  float* out; float4* g;
  hipMalloc(&out, 64); hipMalloc(&g, (size_t)1024 * 4096 * 16 + (1 << 20));
  hipMemset(g, 0, (size_t)1024 * 4096 * 16 + (1 << 20));
  run<1>("ds_read", g, out);
  run<1 | 2>("ds_read + barrier", g, out);
  run<1 | 2 | 4>("ds_read + barrier + ds_write", g, out);
  run<1 | 2 | 4 | 16>("ds_read + barrier + ds_write, double buffer", g, out);
  run<1 | 2 | 4 | 8 | 16>("all: + global loads (register prefetch)", g, out);
  return 0;
}

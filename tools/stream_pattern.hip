// Bandwidth of the GEMM's weight-read pattern without any compute (diagnostic, not part of the library).
// pattern 0: each workgroup (tile n, split z) reads rows [64n, 64n+64) x k-range, 128 B per row per step (the GEMM's)
// pattern 1: same bytes, but each step reads one contiguous 8 KB chunk (k-block-major packed layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ B, int N, int K, int steps_per, int pattern, float* out) {
  int n0 = blockIdx.x * 64, z = blockIdx.y, tid = threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  int s0 = z * steps_per;
  for (int s = s0; s < s0 + steps_per; ++s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + 256 * u;
      size_t off;
      if (pattern == 0) {
        int row = idx >> 3, kq = idx & 7;
        off = ((size_t)(n0 + row) * K + s * 32 + 4 * kq) / 4;
      } else {
        off = ((size_t)s * N * 32 + (size_t)n0 * 32 + idx * 4) / 4;
      }
      float4 v = B[off];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}
int main() {
  const int N = 4800, K = 5632;
  float* B; float* out;
  hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&out, 64);
  hipMemset(B, 0, (size_t)N * K * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int splits : {1, 4, 8, 11, 16, 22}) {
      int steps = K / 32 / splits;
      dim3 grid(N / 64, splits);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rd, grid, dim3(256), 0, 0, (const float4*)B, N, K, steps, pattern, out);
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(rd, grid, dim3(256), 0, 0, (const float4*)B, N, K, steps, pattern, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double bytes = (double)N * 32 * steps * splits * 4;
      printf("pattern %d splits %2d wgs %4d: %7.1f us  %6.2f TB/s\n", pattern, splits, 75 * splits, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    }
  return 0;
}

// Diagnostic: the 3xBF16 k-step (12 bf16 MFMA + 12 ds_read_b128 + split of 4 float4 + 12 ds_write_b64 + barriers) in
// isolation, as a function of resident workgroups per CU and of which ingredients are present.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& mid, u32x2& lo) {
  unsigned u[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i];
    u[i] = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(u[i]);
    m[i] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m[i]);
    l[i] = __float_as_uint(r2);
  }
  hi[0] = (u[0] >> 16) | u[1];  hi[1] = (u[2] >> 16) | u[3];
  mid[0] = (m[0] >> 16) | m[1]; mid[1] = (m[2] >> 16) | m[3];
  lo[0] = (l[0] >> 16) | (l[1] & 0xffff0000u); lo[1] = (l[2] >> 16) | (l[3] & 0xffff0000u);
}
// FEAT: 1 mfma+ds_read, 2 split VALU, 4 ds_write, 8 barriers
template <int FEAT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * 64 * 80];
  for (int i = threadIdx.x; i < 6 * 64 * 80 / 4; i += 256) ((unsigned*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  f32x4 r[4];
  for (int u = 0; u < 4; ++u) r[u] = f32x4{seed + tid, seed * 2, seed * 3 + u, seed * 5};
  const unsigned char* pa = lds + ((wave >> 1) * 32 + (lane & 31)) * 80 + (lane >> 5) * 16;
  const unsigned char* pb = lds + 3 * 5120 + ((wave & 1) * 32 + (lane & 31)) * 80 + (lane >> 5) * 16;
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
    if (FEAT & 1) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(pa + kk * 32));
        bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)(pa + 5120 + kk * 32));
        bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(pa + 10240 + kk * 32));
        bf16x8 bh = __builtin_bit_cast(bf16x8, *(const u32x4*)(pb + kk * 32));
        bf16x8 bm = __builtin_bit_cast(bf16x8, *(const u32x4*)(pb + 5120 + kk * 32));
        bf16x8 bl = __builtin_bit_cast(bf16x8, *(const u32x4*)(pb + 10240 + kk * 32));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      }
    }
    if (FEAT & 8) __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2 hi, mid, lo;
      if (FEAT & 2) split4(r[u], hi, mid, lo);
      else { hi[0] = __float_as_uint(r[u][0]); hi[1] = __float_as_uint(r[u][1]); mid = hi; lo = hi; }
      int idx = tid + 256 * (u & 1);
      unsigned char* p = lds + (u >> 1) * 3 * 5120 + (idx >> 3) * 80 + (idx & 7) * 8;
      if (FEAT & 4) { *(u32x2*)p = hi; *(u32x2*)(p + 5120) = mid; *(u32x2*)(p + 10240) = lo; }
      else sink ^= hi[0] ^ mid[1] ^ lo[0];
      r[u][0] += 1.0f;  // keep the split loop-variant
    }
    if (FEAT & 8) __syncthreads();
  }
  float s = (float)sink;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.f) out[0] = s;
}
template <int FEAT>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 4}) {
    int iters = 2000;
    hipLaunchKernelGGL(k<FEAT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FEAT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s wg/cu %d: %8.1f us  -> %6.0f cycles per workgroup-iteration on a CU\n", name, wgs_per_cu, ms * 1e3,
           ms * 1e-3 / iters / wgs_per_cu * 2.4e9);
  }
}
int main() {
  float* out; hipMalloc(&out, 64);
  run<1>("mfma+ds_read", out);
  run<2>("split only", out);
  run<2 | 4>("split + ds_write", out);
  run<1 | 2 | 4>("mfma + split + ds_write (no barrier)", out);
  run<1 | 2 | 4 | 8>("all (2 barriers)", out);
  run<1 | 4 | 8>("all minus split VALU", out);
  return 0;
}

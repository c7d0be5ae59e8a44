// Diagnostic: cost and correctness of a software grid barrier on MI355X (one 512-thread workgroup per CU).
// Every workgroup writes a value, crosses the barrier, reads its neighbours' values and checks them.
// build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// two-level arrival (16 groups of nb/16 workgroups) so that no address sees more than 16 atomics per barrier; `counter` is
// an array of 17 unsigned spaced 64 B apart: [0..15] group counters, [16] top counter
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned epoch, unsigned nb, unsigned* abort_flag) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                   // release: this workgroup's writes
    const unsigned grp = blockIdx.x & 15u, per = nb / 16u;
    unsigned* top = counter + 16 * 16;
    if (atomicAdd(counter + grp * 16, 1u) == per * epoch - 1u) atomicAdd(top, 1u);   // last arriver of the group
    unsigned spins = 0;
    while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u * epoch) {
      if (++spins > (1u << 24) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {  // bounded: never hang
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
    __threadfence();                                   // acquire: the other workgroups' writes
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(512) void probe(float* data, unsigned* counter, unsigned* abort_flag, int iters, int payload, int* errors,
                                             long long* ticks) {
  const int nb = gridDim.x, b = blockIdx.x;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    float* cur = data + (size_t)(it & 1) * nb * payload;
    for (int i = threadIdx.x; i < payload; i += blockDim.x) cur[(size_t)b * payload + i] = (float)(it * 1000 + b) + i * 0.001f;
    if (!grid_barrier(counter, (unsigned)(it + 1), (unsigned)nb, abort_flag)) break;
    // read two other workgroups' payloads (likely other XCDs)
    for (int k = 1; k <= 2; ++k) {
      const int o = (b + k * 37) % nb;
      for (int i = threadIdx.x; i < payload; i += blockDim.x) {
        const float want = (float)(it * 1000 + o) + i * 0.001f;
        if (cur[(size_t)o * payload + i] != want) atomicAdd(errors, 1);
      }
    }
  }
  if (b == 0 && threadIdx.x == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
  const int nb = 256, iters = 2000;
  for (int payload : {64, 4096, 65536}) {
    float* data; unsigned *counter, *abortf; int* errors; long long* ticks;
    hipMalloc(&data, sizeof(float) * 2 * nb * payload);
    hipMalloc(&counter, 4 * 16 * 17); hipMalloc(&abortf, 4); hipMalloc(&errors, 4); hipMalloc(&ticks, 8);
    hipMemset(counter, 0, 4 * 16 * 17); hipMemset(abortf, 0, 4); hipMemset(errors, 0, 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, data, counter, abortf, iters, payload, errors, ticks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    int e; unsigned a; long long t;
    hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost); hipMemcpy(&a, abortf, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("payload %6d floats/WG: %d iterations, %.2f us per (write + barrier + 2 reads), errors %d, aborted %u\n", payload, iters,
           t * 0.01 / iters, e, a);
    hipFree(data); hipFree(counter); hipFree(abortf); hipFree(errors); hipFree(ticks);
  }
  return 0;
}

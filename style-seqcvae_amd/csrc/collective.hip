// Direct all-reduce of a flat fp32 gradient range over peer-mapped buffers (hipIpc), for the data-parallel train step:
// one process per GPU, every rank's flat gradient buffer and a small flag block are mapped into every other rank's address
// space; a rank reads its peers' memory straight over xGMI.
//
// Why: the reference's only multi-GPU path is nn.DataParallel's reduce-add to device 0
// (var_updown/scripts/train.py:123-124).  Here the exchange is one sum over 446 MB of gradients per step.  xGMI is
// point-to-point (7 links per GPU): a ring moves 2 (N-1)/N S over ONE link per direction (~5 ms at C2), a direct
// reduce-scatter + all-gather uses all 7 links at once (2 S / N per link: ~0.7 ms).  This file is that direct form; RCCL
// (torch.distributed "nccl") stays the reference result it is checked against at start-up (ssc_runtime/xgmi.py).
//
//   reduce-scatter: rank r sums shard r of every rank's buffer (fixed rank order 0..W-1) in place into its own shard r
//   all-gather:     rank r copies the reduced shard j from rank j's buffer into its own, for every j != r
// Shards are disjoint, so both phases work in place: in phase 1 rank r writes only its shard r, which no peer reads in phase 1;
// in phase 2 it writes the shards j != r, which no peer reads in phase 2.
//
// Cross-process ordering uses monotonic sequence numbers in the flag blocks, written by a one-workgroup SIGNAL kernel (system
// scope release after the stream's earlier kernels have completed) and awaited by a one-workgroup WAIT kernel: no data kernel
// ever spins, so a waiting rank occupies one wave and can never keep a peer's kernels off the device (two ranks may share one
// GPU in tests).  Every wait is bounded: after `timeout` polls it raises the error word and returns, so the grid always drains.
#include "ssc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Peers {
  const float* src[SSC_XGMI_MAX_RANKS];
  int world, rank;
};

__global__ void xgmi_signal_kernel(ssc_xgmi_comm c, int stage, unsigned seq) {
  // everything this stream launched before has completed (kernel boundary); make it visible system-wide, then publish
  __threadfence_system();
  const int j = threadIdx.x;
  if (j < c.world)
    __hip_atomic_store(c.flags[j] + stage * SSC_XGMI_MAX_RANKS + c.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void xgmi_wait_kernel(ssc_xgmi_comm c, int stage, unsigned seq, unsigned timeout, int* err) {
  const int j = threadIdx.x;
  if (j < c.world) {
    const unsigned* f = c.flags[c.rank] + stage * SSC_XGMI_MAX_RANKS + j;   // my own flag block: peers write, I poll locally
    unsigned it = 0;
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(32);
      if (++it >= timeout) {   // a peer never arrived: flag the error and leave (the caller checks it; results are invalid)
        if (err) atomicExch(err, 1 + stage);
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();   // acquire: the data kernels that follow read what the signalling ranks wrote before their release
}

// W 16-byte loads that bypass this device's caches (sc0 sc1 = system scope: a peer's bytes must come from its memory, not from a
// line an earlier collective left in this L2), issued back to back and WAITED FOR INSIDE THE SAME asm statement: hipcc treats an
// asm output as valid when the statement ends, so a hand-issued load whose wait sits in a later statement may have its
// destination copied while it is still in flight (seen here: half of every second float4 came back stale).
#define XL(j) "global_load_dwordx4 %[o" #j "], %[p" #j "], off sc0 sc1\n"
#define XO(j) [o##j] "=&v"(v[j])
#define XP(j) [p##j] "v"(p[j])
#define XW "s_waitcnt vmcnt(0)"
template <int W>
__device__ __forceinline__ void load_sys(f32x4 (&v)[W], const float* const (&p)[W]) {
  static_assert(W >= 1 && W <= 8, "ranks");
  if constexpr (W == 1) asm volatile(XL(0) XW : XO(0) : XP(0) : "memory");
  else if constexpr (W == 2) asm volatile(XL(0) XL(1) XW : XO(0), XO(1) : XP(0), XP(1) : "memory");
  else if constexpr (W == 3) asm volatile(XL(0) XL(1) XL(2) XW : XO(0), XO(1), XO(2) : XP(0), XP(1), XP(2) : "memory");
  else if constexpr (W == 4) asm volatile(XL(0) XL(1) XL(2) XL(3) XW : XO(0), XO(1), XO(2), XO(3) : XP(0), XP(1), XP(2), XP(3) : "memory");
  else if constexpr (W == 5) asm volatile(XL(0) XL(1) XL(2) XL(3) XL(4) XW : XO(0), XO(1), XO(2), XO(3), XO(4) : XP(0), XP(1), XP(2), XP(3), XP(4) : "memory");
  else if constexpr (W == 6) asm volatile(XL(0) XL(1) XL(2) XL(3) XL(4) XL(5) XW : XO(0), XO(1), XO(2), XO(3), XO(4), XO(5) : XP(0), XP(1), XP(2), XP(3), XP(4), XP(5) : "memory");
  else if constexpr (W == 7) asm volatile(XL(0) XL(1) XL(2) XL(3) XL(4) XL(5) XL(6) XW : XO(0), XO(1), XO(2), XO(3), XO(4), XO(5), XO(6) : XP(0), XP(1), XP(2), XP(3), XP(4), XP(5), XP(6) : "memory");
  else asm volatile(XL(0) XL(1) XL(2) XL(3) XL(4) XL(5) XL(6) XL(7) XW : XO(0), XO(1), XO(2), XO(3), XO(4), XO(5), XO(6), XO(7) : XP(0), XP(1), XP(2), XP(3), XP(4), XP(5), XP(6), XP(7) : "memory");
}
#undef XL
#undef XO
#undef XP
#undef XW

// out[i] = sum_j src[j][i], i in [0, n4) float4 units, j in fixed rank order; out may alias src[rank].  W = world size (compile
// time): one float4 per rank in flight per thread, many waves in flight per CU
// err: the collective's error word - once a bounded wait of this (or an earlier) collective has given up, the peers' data cannot be
// trusted to be complete: the data kernels then leave the buffer as it is instead of writing a partial sum the update would use
template <int W>
__global__ __launch_bounds__(256) void xgmi_reduce_kernel(Peers p, float* __restrict__ out, size_t n4, const int* __restrict__ err) {
  if (err && *err) return;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 a[W];
    const float* q[W];
#pragma unroll
    for (int j = 0; j < W; ++j) q[j] = p.src[j] + 4 * i;
    load_sys<W>(a, q);
    f32x4 sa = a[0];
#pragma unroll
    for (int j = 1; j < W; ++j) sa += a[j];
    *reinterpret_cast<f32x4*>(out + 4 * i) = sa;
  }
}
typedef void (*reduce_fn)(Peers, float*, size_t, const int*);
inline reduce_fn pick_reduce(int W) {
  switch (W) {
    case 1: return xgmi_reduce_kernel<1>;
    case 2: return xgmi_reduce_kernel<2>;
    case 3: return xgmi_reduce_kernel<3>;
    case 4: return xgmi_reduce_kernel<4>;
    case 5: return xgmi_reduce_kernel<5>;
    case 6: return xgmi_reduce_kernel<6>;
    case 7: return xgmi_reduce_kernel<7>;
    default: return xgmi_reduce_kernel<8>;
  }
}

struct Gather {
  const float* src[SSC_XGMI_MAX_RANKS];   // shard j as it lies in rank j's buffer (nullptr for j == rank)
  float* dst[SSC_XGMI_MAX_RANKS];         // the same shard in the local buffer
  size_t n4[SSC_XGMI_MAX_RANKS];
  int world;
};
// blockIdx.y = source rank
__global__ __launch_bounds__(256) void xgmi_gather_kernel(Gather g, const int* __restrict__ err) {
  const int j = blockIdx.y;
  if (j >= g.world || !g.src[j] || (err && *err)) return;
  const float* s = g.src[j];
  float* d = g.dst[j];
  const size_t n4 = g.n4[j], stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
    f32x4 v[4];
    const float* q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = s + 4 * (i + u * stride < n4 ? i + u * stride : i);
    load_sys<4>(v, q);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n4) *reinterpret_cast<f32x4*>(d + 4 * (i + u * stride)) = v[u];
  }
}

inline hipStream_t S(void* s) { return (hipStream_t)s; }

int check_comm(const ssc_xgmi_comm* c) {
  if (!c || c->world < 1 || c->world > SSC_XGMI_MAX_RANKS || c->rank < 0 || c->rank >= c->world) return SSC_EINVAL;
  for (int j = 0; j < c->world; ++j) {
    if (!c->buf[j] || !c->flags[j]) return SSC_EINVAL;
    if (!ssc_aligned16(c->buf[j])) return SSC_EALIGN;
  }
  return SSC_OK;
}

}  // namespace

// IPC plumbing in plain HIP (no framework internals): the owner exports the ALLOCATION that contains `ptr` (hipIpcGetMemHandle
// wants its base address) plus ptr's offset inside it; a peer process opens the handle UNDER ITS OWN CURRENT DEVICE - the mapping
// is then made for the device whose kernels will read it (and its P2P peers), not for the device that owns the memory - and
// adds the offset.
extern "C" int ssc_xgmi_ipc_export(const void* ptr, void* handle64, size_t* offset) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  if (!ptr || !handle64 || !offset) return SSC_EINVAL;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
  if (e == hipSuccess) e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, (void*)base);
  if (e != hipSuccess) { ssc_tls_hip_error = (int)e; (void)hipGetLastError(); return SSC_EHIP; }
  *offset = (size_t)((const char*)ptr - (const char*)base);
  return SSC_OK;
}
extern "C" int ssc_xgmi_ipc_open(const void* handle64, void** base_out) {
  if (!handle64 || !base_out) return SSC_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(base_out, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { ssc_tls_hip_error = (int)e; (void)hipGetLastError(); return SSC_EHIP; }
  return SSC_OK;
}
extern "C" int ssc_xgmi_ipc_close(void* base) {
  if (!base) return SSC_OK;
  hipError_t e = hipIpcCloseMemHandle(base);
  if (e != hipSuccess) { ssc_tls_hip_error = (int)e; (void)hipGetLastError(); return SSC_EHIP; }
  return SSC_OK;
}

// Copies `bytes` from a (peer-mapped) device pointer to host memory with the runtime's copy engine: the first touch of a fresh
// mapping happens here, where a bad mapping comes back as an error code instead of a GPU fault inside a kernel.
extern "C" int ssc_xgmi_peek(const void* src, void* dst_host, size_t bytes) {
  if (!src || !dst_host || !bytes) return SSC_EINVAL;
  hipError_t e = hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { ssc_tls_hip_error = (int)e; (void)hipGetLastError(); return SSC_EHIP; }
  return SSC_OK;
}

extern "C" int ssc_xgmi_enable_peer(int peer_device) {
  hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return SSC_OK; }
  if (e != hipSuccess) { ssc_tls_hip_error = (int)e; (void)hipGetLastError(); return SSC_EHIP; }
  return SSC_OK;
}

// shard j of [lo, hi): floats [lo + j sh, min(hi, lo + (j+1) sh)), sh = ceil((hi - lo) / 4 / world) * 4
static inline void shard_of(size_t lo, size_t hi, int world, int j, size_t* s_lo, size_t* s_hi) {
  const size_t n4 = (hi - lo) / 4, sh4 = (n4 + world - 1) / world;
  size_t a = lo + 4 * sh4 * (size_t)j, b = a + 4 * sh4;
  if (a > hi) a = hi;
  if (b > hi) b = hi;
  *s_lo = a; *s_hi = b;
}

extern "C" int ssc_xgmi_allreduce(const ssc_xgmi_comm* c, size_t lo, size_t hi, unsigned seq, unsigned timeout, int* err,
                                  void* stream) {
  SSC_TRY(check_comm(c));
  if (hi < lo || (lo & 3) || (hi & 3)) return SSC_EALIGN;   // whole 16-byte units (FlatStore aligns every tensor to 16 bytes)
  if (hi == lo) return SSC_OK;
  if (timeout == 0) timeout = 1u << 22;   // ~ seconds of s_sleep(32) polls: a bound, never reached when every rank takes part
  const int W = c->world, r = c->rank;
  hipStream_t st = S(stream);
  // stage 0: every rank's gradients of this range are complete
  SSC_LAUNCH(xgmi_signal_kernel, dim3(1), dim3(64), 0, st, *c, 0, seq);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(xgmi_wait_kernel, dim3(1), dim3(64), 0, st, *c, 0, seq, timeout, err);
  SSC_CHECK_LAUNCH();
  size_t s_lo, s_hi;
  shard_of(lo, hi, W, r, &s_lo, &s_hi);
  if (s_hi > s_lo) {
    Peers p;
    p.world = W; p.rank = r;
    for (int j = 0; j < SSC_XGMI_MAX_RANKS; ++j) p.src[j] = j < W ? c->buf[j] + s_lo : nullptr;
    const size_t n4 = (s_hi - s_lo) / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    SSC_LAUNCH(pick_reduce(W), dim3(grid), dim3(256), 0, st, p, c->buf[r] + s_lo, n4, (const int*)err);
    SSC_CHECK_LAUNCH();
  }
  // stage 1: every rank's reduced shard is in its buffer
  SSC_LAUNCH(xgmi_signal_kernel, dim3(1), dim3(64), 0, st, *c, 1, seq);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(xgmi_wait_kernel, dim3(1), dim3(64), 0, st, *c, 1, seq, timeout, err);
  SSC_CHECK_LAUNCH();
  if (W > 1) {
    Gather g;
    g.world = W;
    size_t most = 0;
    for (int j = 0; j < SSC_XGMI_MAX_RANKS; ++j) {
      g.src[j] = nullptr; g.dst[j] = nullptr; g.n4[j] = 0;
      if (j >= W || j == r) continue;
      size_t a, b;
      shard_of(lo, hi, W, j, &a, &b);
      if (b <= a) continue;
      g.src[j] = c->buf[j] + a; g.dst[j] = c->buf[r] + a; g.n4[j] = (b - a) / 4;
      if (g.n4[j] > most) most = g.n4[j];
    }
    if (most) {
      int gx = (int)((most + 1023) / 1024);
      if (gx > 256) gx = 256;
      if (gx < 1) gx = 1;
      SSC_LAUNCH(xgmi_gather_kernel, dim3(gx, W), dim3(256), 0, st, g, (const int*)err);
      SSC_CHECK_LAUNCH();
    }
  }
  // stage 2: every rank has finished reading its peers' shards - the buffers may be overwritten (next backward)
  SSC_LAUNCH(xgmi_signal_kernel, dim3(1), dim3(64), 0, st, *c, 2, seq);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(xgmi_wait_kernel, dim3(1), dim3(64), 0, st, *c, 2, seq, timeout, err);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssc_common.h"
#include "ssc_debug.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int KC_LD = BK + 4;  // 36 floats: 144 B rows, 16-B aligned, conflict-free b128 reads

struct KSeg {
  const float* A;
  const float* B;
  int lda, ldb, K, nsteps;
  int avec, bvec;  // 16 B/lane global loads allowed for this segment's A / B operand
  // 2xFP16 form only: the operand also exists as PRE-SPLIT fp16 planes (ssc_gemm_seg.A16 / B16, made by ssc_split_f16): per row
  // and 32-k block 32 hi halfs then 32 lo halfs = the footprint of 32 floats, so a plane operand is addressed exactly like the
  // fp32 one (ld in 4-byte words) and the producers copy it into LDS without any arithmetic.  nullptr: split in the kernel.
  const float* A16;
  const float* B16;
  int lda16, ldb16;
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
  // optional device-side row compaction (gemm_x3b_kernel only; see ssc_gemm_desc)
  const int* mcount;  // M = min(M, *mcount)
  const int* arows;   // A (k-contiguous) row r is read from row arows[r]
  const int* crows;   // C row r is written to row crows[r]
  const int* kcount;  // K = min(K, *kcount)            (single segment, m/n-contiguous operands)
  const int* karows;  // A's k-row k is read from row karows[k]
  const int* kbrows;  // B's k-row k is read from row kbrows[k]
  int tile_gm;        // tile rows per group of the launch's tile order (tile_order)
  int store_wt;       // x3w epilogue: 1 = write-through (sc1) stores of the output tile (split-K slabs: nothing left dirty in L2
                      // for the kernel boundary to write back)
  int member;         // index of this product inside a grouped launch (address-audit build: which record it reports to)
  float* topk;            // wave-specialised 128x128 form only: instead of storing C, every (row, 128-column tile) leaves a record of 6
                          // floats at topk[(row * ntn + tile) * 6]: max, sum exp(x - max), best value, its column, second best, its column
                          // (columns as int bits; ties to the lower column) - ssc_gemm_desc.topk_part
  const float* a_scale;   // 2xFP16 form only: power-of-two factors (device scalars, optional) the A / B operands are multiplied with
  const float* b_scale;   // before the fp16 split; the result is multiplied with the exact inverse of their product
};

// Tile order of a launch (speed only; a bijection for any grid).  Workgroups are dealt round-robin over the 8 XCDs, each
// with its own L2: (1) every XCD gets a CONTIGUOUS run of the linear tile order; (2) the linear order walks the grid in
// groups of `gm` tile rows, column by column inside a group, so that the ~64 workgroups resident on an XCD at one time
// form a roughly square patch (gm x 8 tiles) and share their A row-panels AND B column-panels in that L2.  A row-major
// run of 64 tiles of a 40 x 79 grid touches 64 B panels + 1 A panel per k-step (35 % of the requested bytes hit in L2);
// an 8 x 8 patch touches 8 + 8 (87 %).  The large 3xBF16 products are bound by the CU's load path, not by the matrix
// pipe (MFMA busy ~50 % at 21 B/clk requested per CU), so L2 hits are what raises their rate.  gm = 0: row-major.
__device__ __forceinline__ void tile_order(int lin, int gx, int gy, int gm, int& bx, int& by) {
  const int total = gx * gy;
  const int q = total >> 3, rem = total & 7;          // XCD x owns q (+1 if x < rem) tiles
  const int xcd = lin & 7, slot = lin >> 3;
  const int nl = xcd * q + (xcd < rem ? xcd : rem) + slot;
  if (gm <= 1 || gy <= 1) {
    by = nl / gx;
    bx = nl - by * gx;
    return;
  }
  const int group_sz = gm * gx;
  const int g = nl / group_sz;
  const int first_row = g * gm;
  const int rows = min(gy - first_row, gm);
  const int r = nl - g * group_sz;
  bx = r / rows;
  by = first_row + (r - bx * rows);
}

// ---- global -> register staging of one ROWS x 32 operand tile (ROWS/8 floats per thread) --------------
// Loads are UNCONDITIONAL: out-of-range rows / k are clamped to a valid address and (for k) zeroed with a select
// afterwards.  No exec-masked branch surrounds a load, so hipcc keeps counted vmcnt waits and the register
// prefetch really stays in flight (a guarded load makes it branch and wait vmcnt(0) per element).
// Out-of-range ROWS need no zeroing: they only feed output rows / columns the epilogue never stores.
// VEC (16 B/lane) requires: base 16-B aligned, ld % 4 == 0 and K % 4 == 0 (KC) resp. rows % 4 == 0 (MC), so a
// float4 is never partially out of range.
// The k-range select is DEFERRED to store_tile (bit u of the returned mask = element/vector u is in range): consuming
// a loaded value right after its load would force the wait to the load site and serialise the prefetch.
template <bool KC, int ROWS, bool VEC>
__device__ __forceinline__ unsigned load_tile(float (&r)[ROWS / 8], const float* __restrict__ base, int ld, int rows, int K,
                                              int row0, int k0, int tid) {
  constexpr int NV = ROWS / 32;  // float4 per thread
  unsigned okm = 0;
  if constexpr (KC && VEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int idx = tid + 256 * u;
      int row = idx >> 3, kq = idx & 7;
      int grow = min(row0 + row, rows - 1), k = k0 + 4 * kq;
      float4 v = *reinterpret_cast<const float4*>(base + (size_t)grow * ld + min(k, K - 4));
      okm |= (k < K ? 1u : 0u) << u;
      r[4 * u] = v.x; r[4 * u + 1] = v.y; r[4 * u + 2] = v.z; r[4 * u + 3] = v.w;
    }
  } else if constexpr (KC && !VEC) {
#pragma unroll
    for (int u = 0; u < 4 * NV; ++u) {
      int idx = tid + 256 * u;
      int row = idx >> 5, kk = idx & 31;
      int grow = min(row0 + row, rows - 1), k = k0 + kk;
      r[u] = base[(size_t)grow * ld + min(k, K - 1)];
      okm |= (k < K ? 1u : 0u) << u;
    }
  } else if constexpr (!KC && VEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int idx = tid + 256 * u;
      int kk = idx / (ROWS / 4), mq = idx % (ROWS / 4);
      int gk = k0 + kk, gm = min(row0 + 4 * mq, rows - 4);
      float4 v = *reinterpret_cast<const float4*>(base + (size_t)min(gk, K - 1) * ld + gm);
      okm |= (gk < K ? 1u : 0u) << u;
      r[4 * u] = v.x; r[4 * u + 1] = v.y; r[4 * u + 2] = v.z; r[4 * u + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4 * NV; ++u) {
      int idx = tid + 256 * u;
      int kk = idx / ROWS, m = idx % ROWS;
      int gk = k0 + kk, gm = min(row0 + m, rows - 1);
      r[u] = base[(size_t)min(gk, K - 1) * ld + gm];
      okm |= (gk < K ? 1u : 0u) << u;
    }
  }
  return okm;
}

template <bool KC, int ROWS, bool VEC>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float (&r)[ROWS / 8], int tid, unsigned okm) {
  constexpr int NV = ROWS / 32;
#define SSC_SEL(u, x) (((okm >> (u)) & 1u) ? (x) : 0.f)
  if constexpr (KC && VEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int idx = tid + 256 * u;
      int row = idx >> 3, kq = idx & 7;
      *reinterpret_cast<float4*>(&s[row * KC_LD + 4 * kq]) =
          make_float4(SSC_SEL(u, r[4 * u]), SSC_SEL(u, r[4 * u + 1]), SSC_SEL(u, r[4 * u + 2]), SSC_SEL(u, r[4 * u + 3]));
    }
  } else if constexpr (KC && !VEC) {
#pragma unroll
    for (int u = 0; u < 4 * NV; ++u) {
      int idx = tid + 256 * u;
      s[(idx >> 5) * KC_LD + (idx & 31)] = SSC_SEL(u, r[u]);
    }
  } else if constexpr (!KC && VEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int idx = tid + 256 * u;
      int kk = idx / (ROWS / 4), mq = idx % (ROWS / 4);
      *reinterpret_cast<float4*>(&s[kk * ROWS + 4 * mq]) =
          make_float4(SSC_SEL(u, r[4 * u]), SSC_SEL(u, r[4 * u + 1]), SSC_SEL(u, r[4 * u + 2]), SSC_SEL(u, r[4 * u + 3]));
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4 * NV; ++u) {
      int idx = tid + 256 * u;
      s[(idx / ROWS) * ROWS + (idx % ROWS)] = SSC_SEL(u, r[u]);
    }
  }
#undef SSC_SEL
}

// fragment of 4 k-values (k = chunk*8 + 4*half + j) for tile row/col `rc`
template <bool KC, int ROWS>
__device__ __forceinline__ void read_frag(float (&f)[4], const float* __restrict__ s, int rc, int chunk, int half) {
  if constexpr (KC) {
    float4 v = *reinterpret_cast<const float4*>(&s[rc * KC_LD + chunk * 8 + 4 * half]);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = s[(chunk * 8 + 4 * half + j) * ROWS + rc];
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

// Walks the k-steps of the segment list one step at a time, keeping the current segment's descriptor in scalar
// registers: the steady-state loop then issues no scalar (kernarg) loads - they only happen on a segment change.
struct Cursor {
  const float* A;
  const float* B;
  int lda, ldb, K, k0, seg, left;  // left = k-steps remaining in this segment after the current one
  __device__ __forceinline__ void fetch(const KArgs& g) {
    A = g.seg[seg].A; B = g.seg[seg].B; lda = g.seg[seg].lda; ldb = g.seg[seg].ldb; K = g.seg[seg].K;
  }
  __device__ __forceinline__ void init(const KArgs& g, int step) {
    locate(g, step, seg, k0);
    fetch(g);
    left = g.seg[seg].nsteps - 1 - k0 / BK;
  }
  // move to the next k-step unless `stay` (past the end of this workgroup's range: keep re-reading the last tile)
  // returns 0: stayed, 1: next k-step of the same segment, 2: first k-step of the next segment
  __device__ __forceinline__ int advance(const KArgs& g, bool stay) {
    if (stay) return 0;
    if (left > 0) {
      --left;
      k0 += BK;
      return 1;
    }
    ++seg;
    fetch(g);
    k0 = 0;
    left = g.seg[seg].nsteps - 1;
    return 2;
  }
};

// One staged operand-pair (A tile + B tile of one k-step) held in registers between its global load and its LDS store.
// Generic (4 B/lane) form: compiler-managed loads.
template <bool A_KC, bool B_KC, int RA, int RB, bool VEC>
struct Stage {
  static constexpr int NLOADS = 0;  // not hand-counted
  float a[RA / 8], b[RB / 8];
  unsigned oka, okb;
  __device__ __forceinline__ void load(const KArgs& g, const Cursor& c, int m0, int n0, int tid) {
    oka = load_tile<A_KC, RA, VEC>(a, c.A, c.lda, g.M, c.K, m0, c.k0, tid);
    okb = load_tile<B_KC, RB, VEC>(b, c.B, c.ldb, g.N, c.K, n0, c.k0, tid);
  }
  template <int YOUNGER>
  __device__ __forceinline__ void wait() {}
  __device__ __forceinline__ void store(float* As, float* Bs, int tid) const {
    store_tile<A_KC, RA, VEC>(As, a, tid, oka);
    store_tile<B_KC, RB, VEC>(Bs, b, tid, okb);
  }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16 B/lane form: the loads are inline asm (global_load_dwordx4) and the waits are hand-counted.  hipcc's own
// waitcnt insertion turned the register prefetch into load -> vmcnt(0) -> use (conservative merges at the loop's
// joins, and a register copy scheduled right behind the loads); with asm loads nothing is consumed before
// wait<YOUNGER>() - `s_waitcnt vmcnt(YOUNGER)` leaves exactly the younger stage's loads in flight - and the
// "+v" operands pin every use of the staged registers behind that wait (cdna_hip_programming.md §5.7).
// All loads are unconditional (clamped addresses), so the counts are static.
template <bool A_KC, bool B_KC, int RA, int RB>
struct Stage<A_KC, B_KC, RA, RB, true> {
  static constexpr int NVA = RA / 32, NVB = RB / 32;
  static constexpr int NLOADS = NVA + NVB;
  f32x4 a[NVA], b[NVB];
  unsigned oka, okb;

  template <bool KC, int ROWS, int NV>
  static __device__ __forceinline__ unsigned issue(f32x4 (&r)[NV], const float* __restrict__ base, int ld, int rows, int K,
                                                   int row0, int k0, int tid) {
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int idx = tid + 256 * u;
      const float* p;
      if constexpr (KC) {
        int row = idx >> 3, kq = idx & 7;
        int grow = min(row0 + row, rows - 1), k = k0 + 4 * kq;
        p = base + (size_t)grow * ld + min(k, K - 4);
        okm |= (k < K ? 1u : 0u) << u;
      } else {
        int kk = idx / (ROWS / 4), mq = idx % (ROWS / 4);
        int gk = k0 + kk, gm = min(row0 + 4 * mq, rows - 4);
        p = base + (size_t)min(gk, K - 1) * ld + gm;
        okm |= (gk < K ? 1u : 0u) << u;
      }
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r[u]) : "v"(p) : "memory");
    }
    return okm;
  }

  __device__ __forceinline__ void load(const KArgs& g, const Cursor& c, int m0, int n0, int tid) {
    oka = issue<A_KC, RA, NVA>(a, c.A, c.lda, g.M, c.K, m0, c.k0, tid);
    okb = issue<B_KC, RB, NVB>(b, c.B, c.ldb, g.N, c.K, n0, c.k0, tid);
  }

  // NT fast path: per-thread chunk pointers kept across k-steps (advanced by 32 floats per step, recomputed on a
  // segment change), so the steady state issues its loads without any 64-bit multiply / clamp arithmetic.
  // `full` = the whole 32-wide step is inside the segment (no k clamp, all chunks valid).
  template <bool KC, int ROWS, int NV>
  static __device__ __forceinline__ unsigned tail_issue(f32x4 (&r)[NV], const float* const (&p)[NV], int k0, int K, int ld, int tid) {
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 256 * u;
      const float* q;
      if constexpr (KC) {
        const int k = k0 + 4 * (idx & 7);
        q = p[u] + (min(k, K - 4) - k);
        okm |= (k < K ? 1u : 0u) << u;
      } else {
        const int gk = k0 + idx / (ROWS / 4);
        q = p[u] + (ptrdiff_t)(min(gk, K - 1) - gk) * ld;
        okm |= (gk < K ? 1u : 0u) << u;
      }
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r[u]) : "v"(q) : "memory");
    }
    return okm;
  }
  // Branch-free: every k-step takes the clamped form (a k-step that lies inside its segment clamps nothing).  No branch
  // may surround a staged load: the hand-counted waits rely on every path issuing the same loads in the same order, and
  // tools/check_staged_loads.py proves the register discipline path by path only under that condition.
  __device__ __forceinline__ void load_ptrs(const float* const (&pa)[NVA], const float* const (&pb)[NVB], int k0, int K, int lda,
                                            int ldb, int tid) {
    oka = tail_issue<A_KC, RA, NVA>(a, pa, k0, K, lda, tid);
    okb = tail_issue<B_KC, RB, NVB>(b, pb, k0, K, ldb, tid);
  }

  template <int YOUNGER>
  __device__ __forceinline__ void wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    if constexpr (NVA == 2 && NVB == 2) {
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1])::"memory");
    } else if constexpr (NVA == 2 && NVB == 4) {
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
    } else {
      static_assert(NVA == 4 && NVB == 4, "unsupported tile");
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
    }
  }

  template <bool KC, int ROWS, int NV>
  static __device__ __forceinline__ void put(float* __restrict__ s, const f32x4 (&r)[NV], int tid, unsigned okm) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int idx = tid + 256 * u;
      const bool ok = (okm >> u) & 1u;
      float4 v = make_float4(ok ? r[u][0] : 0.f, ok ? r[u][1] : 0.f, ok ? r[u][2] : 0.f, ok ? r[u][3] : 0.f);
      if constexpr (KC) {
        int row = idx >> 3, kq = idx & 7;
        *reinterpret_cast<float4*>(&s[row * KC_LD + 4 * kq]) = v;
      } else {
        int kk = idx / (ROWS / 4), mq = idx % (ROWS / 4);
        *reinterpret_cast<float4*>(&s[kk * ROWS + 4 * mq]) = v;
      }
    }
  }

  __device__ __forceinline__ void store(float* As, float* Bs, int tid) const {
    put<A_KC, RA, NVA>(As, a, tid, oka);
    put<B_KC, RB, NVB>(Bs, b, tid, okb);
  }
};

// Per-thread chunk pointers of the 16 B/lane kernels, kept across k-steps: advanced by one k-step (32 floats for a
// k-contiguous operand, 32 rows for an m/n-contiguous one) and recomputed only when the segment changes, so the
// steady state issues its loads without 64-bit multiply / clamp arithmetic.
template <bool A_KC, bool B_KC, int RA, int RB>
struct ThreadPtrs {
  static constexpr int NVA = RA / 32, NVB = RB / 32;
  const float* pa[NVA];
  const float* pb[NVB];
  template <bool KC, int ROWS, int NV>
  static __device__ __forceinline__ void base(const float* (&p)[NV], const float* src, int ld, int rows, int row0, int k0, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 256 * u;
      if constexpr (KC) p[u] = src + (size_t)min(row0 + (idx >> 3), rows - 1) * ld + k0 + 4 * (idx & 7);
      else p[u] = src + (size_t)(k0 + idx / (ROWS / 4)) * ld + min(row0 + 4 * (idx % (ROWS / 4)), rows - 4);
    }
  }
  __device__ __forceinline__ void recompute(const KArgs& g, const Cursor& c, int m0, int n0, int tid) {
    base<A_KC, RA, NVA>(pa, c.A, c.lda, g.M, m0, c.k0, tid);
    base<B_KC, RB, NVB>(pb, c.B, c.ldb, g.N, n0, c.k0, tid);
  }
  __device__ __forceinline__ void step(const KArgs& g, const Cursor& c, int how, int m0, int n0, int tid) {
    if (how == 1) {
      const size_t da = A_KC ? (size_t)BK : (size_t)BK * c.lda, db = B_KC ? (size_t)BK : (size_t)BK * c.ldb;
#pragma unroll
      for (int u = 0; u < NVA; ++u) pa[u] += da;
#pragma unroll
      for (int u = 0; u < NVB; ++u) pb[u] += db;
    } else if (how == 2) {
      recompute(g, c, m0, n0, tid);
    }
  }
};

// WM x WN MFMA tiles (32x32 each) per wave; block tile (64*WM) x (64*WN); PF = k-steps prefetched in registers.
template <bool A_KC, bool B_KC, int WM, int WN, int PF, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const KArgs a) {
  constexpr int RA = 64 * WM, RB = 64 * WN;
  constexpr int TA = RA * KC_LD, TB = RB * KC_LD;  // floats per LDS buffer (KC image is the larger one)
  __shared__ __attribute__((aligned(16))) float lds[2 * TA + 2 * TB];
  float* As = lds;
  float* Bs = lds + 2 * TA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, so give
  // each XCD a contiguous run of row-major tile ids - tiles that share an A panel (same M-tile) then share an L2.
  // Speed only; the map is a bijection for any grid size.
  int bx, by;
  tile_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, a.tile_gm, bx, by);
  const int n0 = bx * RB, m0 = by * RA, z = blockIdx.z;

  const int s_lo = z * a.steps_per_split;
  int s_hi = s_lo + a.steps_per_split;
  if (s_hi > a.steps_total) s_hi = a.steps_total;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  typedef Stage<A_KC, B_KC, RA, RB, VEC> StageT;
  constexpr int NL = StageT::NLOADS;
  StageT st[PF];
  const int s_last = s_hi - 1;
  // prologue: tile s_lo -> LDS[0]; tiles s_lo+1 .. s_lo+PF stay in flight in registers (stage j holds tile s+1+j).
  // Loads are unconditional - a step index past the end is clamped and re-reads the last tile, which is never stored.
  Cursor cur;
  int s_ld = s_lo;  // step the cursor points at (the newest tile requested)
  ThreadPtrs<A_KC, B_KC, RA, RB> tp;
  auto issue_loads = [&](StageT& x) {
    if constexpr (VEC) x.load_ptrs(tp.pa, tp.pb, cur.k0, cur.K, cur.lda, cur.ldb, tid);
    else x.load(a, cur, m0, n0, tid);
  };
  auto advance = [&]() {
    const int how = cur.advance(a, s_ld >= s_last);
    s_ld = min(s_ld + 1, s_last);
    if constexpr (VEC) tp.step(a, cur, how, m0, n0, tid);
  };
  if (s_lo < s_hi) {
    cur.init(a, s_lo);
    if constexpr (VEC) tp.recompute(a, cur, m0, n0, tid);
    // the first PF tiles are requested back to back (one exposed memory latency, not two - what a short K range is
    // made of); afterwards stage (r + 1) % PF holds tile s_lo + r + 1 when sub-iteration r starts
    issue_loads(st[0]);
#pragma unroll
    for (int j = 1; j < PF; ++j) {
      advance();
      issue_loads(st[j]);
    }
    st[0].template wait<(PF - 1) * NL>();
    st[0].store(As, Bs, tid);
    advance();
    issue_loads(st[0]);
  }
  __syncthreads();

  auto compute = [&](int buf) {
    const float* as = As + buf * TA;
    const float* bs = Bs + buf * TB;
    constexpr int CH = (WM * WN == 1) ? 4 : 1;  // chunks fetched per batch
#pragma unroll
    for (int c0 = 0; c0 < 4; c0 += CH) {
      float fa[CH][WM][4], fb[CH][WN][4];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) read_frag<A_KC, RA>(fa[c][mi], as, (wm * WM + mi) * 32 + l31, c0 + c, half);
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) read_frag<B_KC, RB>(fb[c][ni], bs, (wn * WN + ni) * 32 + l31, c0 + c, half);
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][mi][j], fb[c][ni][j], acc[mi][ni], 0, 0, 0);
    }
  };

  // steady state, unrolled so that stage and LDS-buffer indices are static: in sub-iteration j tile s+j is in
  // LDS[j&1], stage j%PF holds tile s+j+1 (the OLDEST loads in flight; the other PF-1 stages are younger).
  constexpr int U = PF > 2 ? PF : 2;
  // Every sub-iteration runs the same load / wait / store / barrier sequence; only the MFMA work is skipped in the
  // sub-iterations past the end of the k-range (the trip count is rounded up to a multiple of U).  A store past the end
  // goes to the LDS buffer whose last readers finished before the previous barrier; loads past the end re-read the last
  // tile (clamped cursor).
  for (int s = s_lo; s < s_hi; s += U) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      StageT& x = st[(j + 1) % PF];
      if (s + j < s_hi) compute(j & 1);
      x.template wait<(PF - 1) * NL>();
      x.store(As + ((j + 1) & 1) * TA, Bs + ((j + 1) & 1) * TB, tid);
      cur.advance(a, s_ld >= s_last); s_ld = min(s_ld + 1, s_last);
      x.load(a, cur, m0, n0, tid);
      __syncthreads();
    }
  }
  // Drain the clamped tail loads.  Their results are never used, but the wait must still PIN the staged registers
  // ("+v" in wait()): past the loop the compiler sees them as dead and would hand them to the epilogue's address
  // arithmetic while the loads are still in flight - the returning data then lands in a live pointer (observed as a
  // codegen-dependent memory fault).
  if constexpr (VEC) {
#pragma unroll
    for (int j = 0; j < PF; ++j) st[j].template wait<0>();
  }

  // epilogue: acc[r] -> row (r&3) + 8*(r>>2) + 4*half, col l31 of each 32x32 tile
  float* out = a.out + (size_t)z * a.slab_stride;
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    const int col = n0 + (wn * WN + ni) * 32 + l31;
    if (col >= a.N) continue;
    const float bv = (a.bias != nullptr) ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      const int rbase = m0 + (wm * WM + mi) * 32 + 4 * half;
      float old[16];
      if (a.accumulate) {  // all 16 read-backs in flight together (clamped rows), one wait
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = out[(size_t)min(rbase + (r & 3) + 8 * (r >> 2), a.M - 1) * a.ldo + col];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < a.M) out[(size_t)row * a.ldo + col] = acc[mi][ni][r] + bv + old[r];
      }
    }
  }
}

// =====================================================================================================
// "3xBF16" NT kernel: fp32-accurate GEMM on the bf16 matrix cores.
//
// Every fp32 operand is split EXACTLY into three bf16 pieces x = hi + mid + lo (8 significant bits each, by
// truncation: hi = x & 0xffff0000, mid = (x-hi) & 0xffff0000, lo = x-hi-mid, which has <= 8 significant bits left and
// is itself a bf16).  a*b is then the sum of the six partial products of order <= 2 (hi*hi, hi*mid, mid*hi, hi*lo,
// lo*hi, mid*mid), each exact in fp32 (8+8 significand bits), accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16; the dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-23 |a*b|, i.e. the size of one
// fp32 rounding.  Six bf16 MFMAs (32 cycles, K=16) replace sixteen fp32 MFMAs (64 cycles, K=2) per K=32:
// 384 vs 1024 matrix-pipe cycles, which moves the skinny (M = minibatch) gate GEMMs from fp32-MFMA-bound to
// HBM-bound - the regime BASELINE.json's roofline target is quoted in.  Operands stay fp32 in HBM (4 B/weight
// streamed once); the split happens in registers on the way into LDS.  tests/test_gemm_gpu.py bounds the error
// against float64 next to the exact-fp32 MFMA path.
//
// Layout: 64x64 block tile, 2x2 waves of 32x32; BK = 32; per operand three LDS planes [row][32 bf16 + 8 pad]
// (80-B rows: the 16-lane ds_read_b128 groups hit 16 distinct 16-B slots); A/B fragments of
// v_mfma_f32_32x32x16_bf16 are 8 consecutive k of one row = one ds_read_b128 per plane.
// =====================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int PL_ROW_B = 80;               // bytes per LDS plane row (32 bf16 + 16 B pad)

// split 4 consecutive fp32 (one float4 of k) into the three bf16 planes, packed two bf16 per dword
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& mid, u32x2& lo) {
  unsigned xu[4], r1u[4], r2u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i];
    xu[i] = __float_as_uint(x);
    const float r1 = x - __uint_as_float(xu[i] & 0xffff0000u);
    r1u[i] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(r1u[i] & 0xffff0000u);
    r2u[i] = __float_as_uint(r2);  // <= 8 significant bits: exact as bf16 (low 16 bits are zero)
  }
  // pack the UPPER halves of two dwords into one (element 2j in the low half): one v_perm_b32 per pair and plane - the upper
  // 16 bits of x / r1 / r2 are the hi / mid / lo pieces themselves, so the masked values are needed for the subtractions only
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    hi[j] = __builtin_amdgcn_perm(xu[2 * j + 1], xu[2 * j], 0x07060302u);
    mid[j] = __builtin_amdgcn_perm(r1u[2 * j + 1], r1u[2 * j], 0x07060302u);
    lo[j] = __builtin_amdgcn_perm(r2u[2 * j + 1], r2u[2 * j], 0x07060302u);
  }
}

// 2xFP16 (ssc_model_cfg.gemm_mode 3): 4 consecutive fp32 into TWO fp16 planes - ssc_split4_f16 (ssc_common.h)
__device__ __forceinline__ void split4_f16(const f32x4& v, u32x2& hi, u32x2& lo) { ssc_split4_f16(v, hi, lo); }

// WN = 32x32 MFMA tiles per wave along N: block tile 64 x (64*WN).  WN = 2 halves the A re-reads per streamed weight
// byte (the CU-side load path, ~24 GB/s per CU, is what the skinny products saturate) at one workgroup per CU.
// NBUF = LDS stages: 2 = one barrier per k-step (61 KB: two workgroups per CU); 1 = two barriers per k-step but half the
// LDS (31 KB: four workgroups per CU) - each wave's in-order stream (loads, split, LDS stores, MFMAs) is what a single
// workgroup is bound by, so more resident waves per SIMD matter more than the extra barrier.
template <int PF, int WN, int NBUF>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const KArgs a) {
  constexpr int RB = 64 * WN;
  constexpr int PLA = 64 * PL_ROW_B, PLB = RB * PL_ROW_B;   // bytes per plane
  constexpr int STAGE_B = 3 * PLA + 3 * PLB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * STAGE_B];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  int bx, by;
  tile_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, a.tile_gm, bx, by);
  const int n0 = bx * RB, m0 = by * 64, z = blockIdx.z;
  const int s_lo = z * a.steps_per_split;
  int s_hi = s_lo + a.steps_per_split;
  if (s_hi > a.steps_total) s_hi = a.steps_total;

  f32x16 acc[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ni][i] = 0.f;

  typedef Stage<true, true, 64, RB, true> StageT;
  constexpr int NL = StageT::NLOADS;
  StageT st[PF];
  const int s_last = s_hi - 1;

  // registers -> three bf16 planes per operand (k-range select applied here, as in the fp32 kernel)
  auto put_planes = [&](unsigned char* base, const StageT& x) {
#pragma unroll
    for (int u = 0; u < StageT::NVA; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, kq = idx & 7;
      f32x4 v = x.a[u];
      if (!((x.oka >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 hi, mid, lo;
      split4(v, hi, mid, lo);
      unsigned char* p = base + row * PL_ROW_B + kq * 8;
      *reinterpret_cast<u32x2*>(p) = hi;
      *reinterpret_cast<u32x2*>(p + PLA) = mid;
      *reinterpret_cast<u32x2*>(p + 2 * PLA) = lo;
    }
#pragma unroll
    for (int u = 0; u < StageT::NVB; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, kq = idx & 7;
      f32x4 v = x.b[u];
      if (!((x.okb >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 hi, mid, lo;
      split4(v, hi, mid, lo);
      unsigned char* p = base + 3 * PLA + row * PL_ROW_B + kq * 8;
      *reinterpret_cast<u32x2*>(p) = hi;
      *reinterpret_cast<u32x2*>(p + PLB) = mid;
      *reinterpret_cast<u32x2*>(p + 2 * PLB) = lo;
    }
  };

  Cursor cur;
  int s_ld = s_lo;
  ThreadPtrs<true, true, 64, RB> tp;
  auto issue_loads = [&](StageT& x) { x.load_ptrs(tp.pa, tp.pb, cur.k0, cur.K, cur.lda, cur.ldb, tid); };
  if (s_lo < s_hi) {
    cur.init(a, s_lo);
    tp.recompute(a, cur, m0, n0, tid);
    // first PF tiles requested back to back (see gemm_kernel); stage (r + 1) % PF then holds tile s_lo + r + 1
    issue_loads(st[0]);
#pragma unroll
    for (int j = 1; j < PF; ++j) {
      tp.step(a, cur, cur.advance(a, s_ld >= s_last), m0, n0, tid); s_ld = min(s_ld + 1, s_last);
      issue_loads(st[j]);
    }
    st[0].template wait<(PF - 1) * NL>();
    put_planes(lds, st[0]);
    tp.step(a, cur, cur.advance(a, s_ld >= s_last), m0, n0, tid); s_ld = min(s_ld + 1, s_last);
    issue_loads(st[0]);
  }
  __syncthreads();

  auto compute = [&](int buf) {
    const unsigned char* pa = lds + buf * STAGE_B + (wm * 32 + l31) * PL_ROW_B + half * 16;
    const unsigned char* pb = lds + buf * STAGE_B + 3 * PLA + (wn * 32 * WN + l31) * PL_ROW_B + half * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pa + kk * 32));
      bf16x8 am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pa + PLA + kk * 32));
      bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pa + 2 * PLA + kk * 32));
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) {
        const unsigned char* pbn = pb + ni * 32 * PL_ROW_B;
        bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pbn + kk * 32));
        bf16x8 bm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pbn + PLB + kk * 32));
        bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pbn + 2 * PLB + kk * 32));
        // smallest partial products first
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ni], 0, 0, 0);
      }
    }
  };

  constexpr int U = PF > 2 ? PF : 2;
  // uniform sub-iterations (see gemm_kernel): only the MFMA work is skipped past the end of the k-range
  for (int s = s_lo; s < s_hi; s += U) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      StageT& x = st[(j + 1) % PF];
      const bool live = s + j < s_hi;
      // One basic block: this step's MFMA chain and the NEXT tile's fp32 -> 3xbf16 split (VALU) + plane stores.  A wave
      // issues in order and a dependent MFMA blocks everything behind it, so the split must sit BETWEEN the MFMAs to
      // run in their shadow (sched_group_barrier below); the store is unconditional (past the end it re-stores the
      // clamped last tile into the stage nobody reads again).
      x.template wait<(PF - 1) * NL>();
      if constexpr (NBUF == 2) {
        if (live) {
          compute(j & 1);
          put_planes(lds + ((j + 1) & 1) * STAGE_B, x);
#pragma unroll
          for (int q = 0; q < 12 * WN; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);  // up to 7 VALU in its shadow
          }
        } else {
          put_planes(lds + ((j + 1) & 1) * STAGE_B, x);
        }
      } else {
        if (live) compute(0);
        __syncthreads();  // every wave is done reading the single stage
        put_planes(lds, x);
      }
      tp.step(a, cur, cur.advance(a, s_ld >= s_last), m0, n0, tid); s_ld = min(s_ld + 1, s_last);
      issue_loads(x);
      __syncthreads();
    }
  }
  // drain the clamped tail loads with the staged registers pinned (see gemm_kernel)
#pragma unroll
  for (int j = 0; j < PF; ++j) st[j].template wait<0>();

  float* out = a.out + (size_t)z * a.slab_stride;
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    const int col = n0 + (wn * WN + ni) * 32 + l31;
    if (col >= a.N) continue;
    const float bv = (a.bias != nullptr) ? a.bias[col] : 0.f;
    const int rbase = m0 + wm * 32 + 4 * half;
    float old[16];
    if (a.accumulate) {
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = out[(size_t)min(rbase + (r & 3) + 8 * (r >> 2), a.M - 1) * a.ldo + col];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = rbase + (r & 3) + 8 * (r >> 2);
      if (row < a.M) out[(size_t)row * a.ldo + col] = acc[ni][r] + bv + old[r];
    }
  }
}

// =====================================================================================================
// 3xBF16 kernel for LARGE products in all three operand layouts (NT, NN, TN): 128x128 block tile, 2x2 waves, each wave
// a 64x64 tile (2x2 v_mfma_f32_32x32x16_bf16 tiles, 64 fp32 accumulators).  Per k-step of 32 a wave issues 48 MFMAs
// against 24 16-byte LDS fragment reads, so the matrix pipe - not the LDS or the wave's instruction stream, which bound
// the 64x64 kernels above - is the limit.  One LDS stage of six bf16 planes (60 KB: two workgroups per CU), one k-step
// prefetched in registers.
//
// Operand images in LDS (10 240 B per plane either way):
//   k-contiguous operand   [128 rows][32 bf16 + 16 B pad] (80-B rows, as gemm_x3_kernel): fragment = ds_read_b128
//   m/n-contiguous operand [32 k][128 bf16 + 64 B pad] (320-B rows): the fp32 tile arrives k-major, is split and stored
//     as it comes (no register transpose), and the fragment - 8 consecutive k of one column - is two
//     ds_read_b64_tr_b16 (cdna_hip_programming.md T10): 320-B rows put the 4 k-rows x 64 B a 32-lane half touches on
//     64 distinct banks.
//
// Device-side row compaction (ssc_gemm_desc m_count / a_rows / c_rows / k_count / ka_rows / kb_rows) lives here: the
// padded (t, b) rows of a caption batch are skipped without the host knowing how many there are.  KG = k-row gather
// lists are in use (weight-gradient products): the row numbers for step s+2 are fetched while step s+1 is in flight, so
// the pointer arithmetic never waits on them in the steady state.
// =====================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int X3B_PLANE = 10240;
constexpr int X3B_MC_ROW_B = 320;

template <bool A_KC, bool B_KC, bool KG>
__global__ __launch_bounds__(256, 2) void gemm_x3b_kernel(const KArgs a) {
  constexpr int PLN = X3B_PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * PLN];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  int bx, by;
  // With a device-side row count only the first ceil(Meff / 128) tile rows hold work.  The tile order deals a CONTIGUOUS run of its
  // linear order to each XCD, so ordering the full grid and letting the workgroups of empty rows exit left all work on the
  // first XCDs (5000 x 4800 x 2400 with 1000 live rows: 552 us against 743 us for all rows - 6 of 8 XCDs idle).  The order is
  // therefore taken over the LIVE tile rows only; workgroups past them exit.
  const int Meff = a.mcount ? min(a.M, *a.mcount) : a.M;  // uniform per launch
  const int gy_live = min((int)gridDim.y, (Meff + 127) / 128);
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  if (lin >= (int)gridDim.x * gy_live) return;
  tile_order(lin, gridDim.x, gy_live, a.tile_gm, bx, by);
  const int n0 = bx * 128, m0 = by * 128, z = blockIdx.z;
  if (m0 >= Meff) return;
  const int Kc = a.kcount ? max(0, min(a.seg[0].K, *a.kcount)) : 0;
  int steps_total = a.steps_total, steps_per_split = a.steps_per_split;
  if (a.kcount) {  // K is known only on the device: partition the k-steps here
    steps_total = (Kc + BK - 1) / BK;
    steps_per_split = (steps_total + (int)gridDim.z - 1) / (int)gridDim.z;
  }
  const int s_lo = z * steps_per_split;
  int s_hi = s_lo + steps_per_split;
  if (s_hi > steps_total) s_hi = steps_total;
  const int s_last = s_hi - 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  // ---- staged tile: 4 + 4 float4 per thread --------------------------------------------------------
  f32x4 ra[4], rb[4];
  unsigned oka = 0, okb = 0;
  bool full_st = false;   // the staged k-step lies inside its segment (workgroup-uniform): no chunk needs zeroing
  const float* pa[4];
  const float* pb[4];
  int ia[4], ib[4];  // KG: gathered k-row numbers of the NEXT step
  Cursor cur;
  int s_ld = s_lo;

  // chunk u of this thread: k-contiguous operand -> (row idx>>3, k 4*(idx&7)); m/n-contiguous -> (k idx>>5, col 4*(idx&31))
  auto base_ptrs = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      if constexpr (A_KC) {
        int r = min(m0 + (idx >> 3), Meff - 1);
        if (a.arows) r = a.arows[r];
        pa[u] = cur.A + (size_t)r * cur.lda + cur.k0 + 4 * (idx & 7);
      } else {
        int kr = cur.k0 + (idx >> 5);
        if constexpr (KG) { if (a.karows) kr = a.karows[min(kr, cur.K - 1)]; }
        pa[u] = cur.A + (size_t)kr * cur.lda + min(m0 + 4 * (idx & 31), Meff - 4);
      }
      if constexpr (B_KC) {
        pb[u] = cur.B + (size_t)min(n0 + (idx >> 3), a.N - 1) * cur.ldb + cur.k0 + 4 * (idx & 7);
      } else {
        int kr = cur.k0 + (idx >> 5);
        if constexpr (KG) { if (a.kbrows) kr = a.kbrows[min(kr, cur.K - 1)]; }
        pb[u] = cur.B + (size_t)kr * cur.ldb + min(n0 + 4 * (idx & 31), a.N - 4);
      }
    }
  };
  // KG: fetch the row numbers of the k-step after the cursor's (clamped; a step past the end is never used)
  auto prefetch_rows = [&]() {
    if constexpr (KG) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kr = min(cur.k0 + BK + ((tid + 256 * u) >> 5), cur.K - 1);
        ia[u] = a.karows ? a.karows[kr] : kr;
        ib[u] = a.kbrows ? a.kbrows[kr] : kr;
      }
    }
  };
  auto step_ptrs = [&](int how) {
    if (how == 2) {
      base_ptrs();
    } else if (how == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u;
        if constexpr (A_KC) pa[u] += BK;
        else if constexpr (KG) pa[u] = cur.A + (size_t)ia[u] * cur.lda + min(m0 + 4 * (idx & 31), Meff - 4);
        else pa[u] += (size_t)BK * cur.lda;
        if constexpr (B_KC) pb[u] += BK;
        else if constexpr (KG) pb[u] = cur.B + (size_t)ib[u] * cur.ldb + min(n0 + 4 * (idx & 31), a.N - 4);
        else pb[u] += (size_t)BK * cur.ldb;
      }
      prefetch_rows();
    }
  };
  // Branch-free (no branch may surround a staged load: see Stage::load_ptrs): every k-step takes the clamped form - a step
  // inside its segment clamps nothing - and remembers which chunks lie past the end of K.
  auto issue_loads = [&]() {
    oka = okb = 0;
    full_st = cur.k0 + BK <= cur.K;
    const float* qa[4];
    const float* qb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      if constexpr (A_KC) {
        const int k = cur.k0 + 4 * (idx & 7);
        qa[u] = pa[u] + (min(k, cur.K - 4) - k);
        oka |= (k < cur.K ? 1u : 0u) << u;
      } else {
        const int gk = cur.k0 + (idx >> 5);
        qa[u] = KG ? pa[u] : pa[u] + (ptrdiff_t)(min(gk, cur.K - 1) - gk) * cur.lda;
        oka |= (gk < cur.K ? 1u : 0u) << u;
      }
      if constexpr (B_KC) {
        const int k = cur.k0 + 4 * (idx & 7);
        qb[u] = pb[u] + (min(k, cur.K - 4) - k);
        okb |= (k < cur.K ? 1u : 0u) << u;
      } else {
        const int gk = cur.k0 + (idx >> 5);
        qb[u] = KG ? pb[u] : pb[u] + (ptrdiff_t)(min(gk, cur.K - 1) - gk) * cur.ldb;
        okb |= (gk < cur.K ? 1u : 0u) << u;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ra[u]) : "v"(qa[u]) : "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rb[u]) : "v"(qb[u]) : "memory");
  };
  // all staged loads have landed; "+v" pins every use of the staged registers behind the wait
  auto wait_loads = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3])::"memory");
  };
  auto advance = [&]() {
    const int how = cur.advance(a, s_ld >= s_last);
    s_ld = min(s_ld + 1, s_last);
    step_ptrs(how);
  };
  // registers -> three bf16 planes per operand (VALU + LDS only: the uniform branch on `full_st` surrounds no load)
  auto put_planes_impl = [&](const bool mask) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      f32x4 v = ra[u];
      if (mask && !((oka >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 hi, mid, lo;
      split4(v, hi, mid, lo);
      unsigned char* p = lds + (A_KC ? (idx >> 3) * PL_ROW_B + (idx & 7) * 8 : (idx >> 5) * X3B_MC_ROW_B + (idx & 31) * 8);
      *reinterpret_cast<u32x2*>(p) = hi;
      *reinterpret_cast<u32x2*>(p + PLN) = mid;
      *reinterpret_cast<u32x2*>(p + 2 * PLN) = lo;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      f32x4 v = rb[u];
      if (mask && !((okb >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 hi, mid, lo;
      split4(v, hi, mid, lo);
      unsigned char* p = lds + 3 * PLN + (B_KC ? (idx >> 3) * PL_ROW_B + (idx & 7) * 8 : (idx >> 5) * X3B_MC_ROW_B + (idx & 31) * 8);
      *reinterpret_cast<u32x2*>(p) = hi;
      *reinterpret_cast<u32x2*>(p + PLN) = mid;
      *reinterpret_cast<u32x2*>(p + 2 * PLN) = lo;
    }
  };
  auto put_planes = [&]() {
    if (full_st) put_planes_impl(false);
    else put_planes_impl(true);
  };

  if (s_lo < s_hi) {
    cur.init(a, s_lo);
    if (a.kcount) { cur.K = Kc; cur.left = steps_total - 1 - cur.k0 / BK; }
    base_ptrs();
    prefetch_rows();
    issue_loads();
    wait_loads();
    put_planes();
    advance();
    issue_loads();
  }
  __syncthreads();

  // fragment = 8 consecutive k (16*kk + 8*half ...) of tile row/column `rc` of one plane
  const int g16 = lane >> 4, i16 = lane & 15;
  auto frag = [&](const unsigned char* plane, bool kc, int rc0, int kk) -> bf16x8 {
    if (kc) {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(plane + (rc0 + l31) * PL_ROW_B + kk * 32 + half * 16));
    } else {
      // transposed read: lane 4q+p of a 16-lane group addresses k-row q, columns 4p..4p+3 of the group's 4 x 16 block and
      // receives column (lane & 15), k-rows 0..3; the second read takes the next 4 k-rows
      const unsigned char* p = plane + (16 * kk + 8 * (g16 >> 1) + (i16 >> 2)) * X3B_MC_ROW_B + (rc0 + 16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * X3B_MC_ROW_B));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      return __builtin_bit_cast(bf16x8, v);
    }
  };
  auto compute = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          fa[t][pl] = frag(lds + pl * PLN, A_KC, wm * 64 + t * 32, kk);
          fb[t][pl] = frag(lds + (3 + pl) * PLN, B_KC, wn * 64 + t * 32, kk);
        }
      // six partial products, smallest first; consecutive MFMAs go to different accumulators
#define SSC_X3B_MFMA(PA, PB)                                                                                        \
  _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][PA], fb[ni][PB], acc[mi][ni], 0, 0, 0);
      SSC_X3B_MFMA(2, 0)
      SSC_X3B_MFMA(0, 2)
      SSC_X3B_MFMA(1, 1)
      SSC_X3B_MFMA(1, 0)
      SSC_X3B_MFMA(0, 1)
      SSC_X3B_MFMA(0, 0)
#undef SSC_X3B_MFMA
    }
  };

  for (int s = s_lo; s < s_hi; ++s) {
    compute();
    wait_loads();
    __syncthreads();  // every wave is done reading the stage
    if (s + 1 < s_hi) put_planes();   // VALU + LDS only
    advance();
    issue_loads();
    __syncthreads();
  }
  wait_loads();  // drain the clamped tail loads with the staged registers pinned (see gemm_kernel)

  float* out = a.out + (size_t)z * a.slab_stride;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + l31;
    if (col >= a.N) continue;
    const float bv = (a.bias != nullptr) ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int rbase = m0 + wm * 64 + mi * 32 + 4 * half;
      int rr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rr[r] = min(rbase + (r & 3) + 8 * (r >> 2), Meff - 1);
        if (a.crows) rr[r] = a.crows[rr[r]];
      }
      float old[16];
      if (a.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = out[(size_t)rr[r] * a.ldo + col];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (rbase + (r & 3) + 8 * (r >> 2) < Meff) out[(size_t)rr[r] * a.ldo + col] = acc[mi][ni][r] + bv + old[r];
      }
    }
  }
}


// =====================================================================================================
// Wave-specialised 3xBF16 kernel (same planes and compaction semantics as gemm_x3b_kernel): 8 waves per workgroup.
// Waves 4-7 PRODUCE - global loads, fp32 -> 3 x bf16 split (VALU) and plane stores into LDS stage (s+1)&1 - while
// waves 0-3 CONSUME stage s&1 (LDS fragment reads + 48 MFMAs per wave and k-step, each wave a 64x64 tile).  Each SIMD
// hosts one producer and one consumer wave, so loads and the split run beside the matrix pipe instead of in series with
// it (in the 4-wave kernels the phases of one workgroup are strictly serial and co-resident workgroups only partly
// overlap: rocprof K-scaling, tools/ktime.py + tools/gemm_clock.py).  ONE barrier per k-step; PF k-steps of operand
// tiles in flight in the producers' registers (the consumers need theirs for accumulators and fragments).
//
// Block tile TM x TN:
//   128 x 128  large products (consumer waves 2 x 2)
//    64 x 256  M = minibatch (<= 64 rows) against a wide weight matrix (consumer waves 1 x 4): per streamed weight byte
//              the 64-row activation tile is re-read and re-split a quarter as often as with a 64-wide tile, and one
//              k-step moves 32 KB of weights per CU, so a single workgroup per CU keeps HBM busy.
// Two LDS stages of six planes: 120 KB / 150 KB of dynamic LDS, one workgroup (8 waves) per CU.
// =====================================================================================================
// Diagnostics build only (tools/x3w_stamp.py compiles a second library with -DSSC_X3W_STAMP; the product build has none of
// this): one workgroup records the shader clock at the phase boundaries of its producer wave 4 and consumer wave 0 into
// 2 KB of LDS behind the operand stages and dumps them at the end.
#ifdef SSC_X3W_STAMP
__device__ unsigned long long* g_stamp_ptr = nullptr;
__device__ int g_stamp_wg = -1;
#define SSC_STAMP(i)                                                                                  \
  do {                                                                                                \
    if (stamp_on && lane == 0 && (i) < 124) stamp_lds[(i)] = __builtin_readcyclecounter();           \
  } while (0)
constexpr int X3W_STAMP_BYTES = 2048;
#else
#define SSC_STAMP(i) do {} while (0)
constexpr int X3W_STAMP_BYTES = 0;
#endif
// Address-audit build only (tools/x3w_audit.py compiles a second library with -DSSC_X3W_AUDIT; the product build has none of
// this): every hand-issued operand load of the wave-specialised kernels reports the byte range it touches, relative to the
// operand base of its K segment; the launch sites synchronise and compare the ranges with the spans the descriptors imply.
#ifdef SSC_X3W_AUDIT
__device__ long long g_audit[6][SSC_MAX_SEG][2][2];   // [group member][segment][A | B][lowest offset, highest offset + 16]
#define SSC_AUDIT_TOUCH(member, seg, op, off)                                       \
  do {                                                                              \
    atomicMin(&g_audit[(member)][(seg)][(op)][0], (long long)(off));                \
    atomicMax(&g_audit[(member)][(seg)][(op)][1], (long long)(off) + 16);           \
  } while (0)
#else
#define SSC_AUDIT_TOUCH(member, seg, op, off) do {} while (0)
#endif
template <int R> struct X3wPlane {   // one bf16 plane of an R-row operand tile: k-contiguous or m/n-contiguous image
  static constexpr int MC_ROW_B = 2 * R + 64;                        // [32 k][R bf16 + 64 B pad]: 4 k-rows x 64 B on 64 banks
  static constexpr int KC_BYTES = R * PL_ROW_B, MC_BYTES = 32 * MC_ROW_B;
  static constexpr int BYTES = KC_BYTES > MC_BYTES ? KC_BYTES : MC_BYTES;
};
template <int TM, int TN> constexpr int x3w_lds_bytes() { return 2 * 3 * (X3wPlane<TM>::BYTES + X3wPlane<TN>::BYTES) + X3W_STAMP_BYTES; }
// the 2xFP16 form keeps two planes per operand: 80 KB for a 128x128 tile - TWO workgroups per CU (160 KB of LDS) when they have 8 waves
template <int TM, int TN> constexpr int x3w_lds_bytes_f16() { return 2 * 2 * (X3wPlane<TM>::BYTES + X3wPlane<TN>::BYTES) + X3W_STAMP_BYTES; }

// F16: the operands are split into two fp16 planes and multiplied with THREE partial products (lo*hi, hi*lo, hi*hi) on
// v_mfma_f32_32x32x16_f16 instead of three bf16 planes and six products - half the matrix-pipe work of a product that is bound by
// it (the decode's 10000-row products: DESIGN.md).  LDS layout unchanged (plane 2 of each operand stays unused).
template <bool A_KC, bool B_KC, bool KG, int TM, int TN, int PF, int NPW, bool F16 = false>
__device__ __forceinline__ void x3w_body(const KArgs& a, const int blk_x, const int blk_y, const int blk_z, const int grid_x,
                                         const int grid_y, const int grid_z) {
  static_assert((TM == 128 && TN == 128) || (TM == 64 && TN == 256), "unsupported tile");
  static_assert(PF >= 1 && PF <= 3, "prefetch depth");
  constexpr int UNR = (PF == 3) ? 6 : 2;         // lcm(LDS stages, register sets)
  // Device-side row lists (mcount / arows / crows) exist for the 128x128 form only: the minibatch kernels are never launched
  // with them, and their mere presence costs the k-loop a compiler-visible gather load + `s_waitcnt vmcnt(0)` at every
  // K-segment change (base_ptrs), which drains the hand-staged prefetch too.
  constexpr bool RL = TM == 128;
  static_assert(!KG || PF == 1, "the gather lists' own loads share the vector-memory counter");
  constexpr int NPL = F16 ? 2 : 3;   // planes per operand
  constexpr int PLA = X3wPlane<TM>::BYTES, PLB = X3wPlane<TN>::BYTES, STAGE = NPL * (PLA + PLB);
  static_assert(TM * (TN + 4) * 4 <= 2 * STAGE, "the epilogue's C tile reuses the operand stages");
  constexpr int MCA = X3wPlane<TM>::MC_ROW_B, MCB = X3wPlane<TN>::MC_ROW_B;
  static_assert(NPW == 4 || NPW == 8, "producer waves");
  constexpr int NPT = 64 * NPW;                  // producer threads
  constexpr int NTHR = 256 + NPT;                // workgroup size: 4 consumer waves + NPW producer waves
  constexpr int NA = TM * 8 / NPT, NB = TN * 8 / NPT;   // float4 chunks per producer thread and k-step
  constexpr int QA = TM / 4, QB = TN / 4;        // float4 per k-row of an m/n-contiguous tile
  // (Tried in round 4: dealing a half-wave the plane rows {r, r+4, r+8, r+12} - 16 banks apart, so that a ds_write_b64 of the
  // 80-byte plane rows touches every bank once instead of wrapping the fourth consecutive row onto the first - made the 128x128
  // products 5-18 % SLOWER (10000 x 10000 x 1200, 2xFP16 form: 846 -> 1003 us): consecutive rows per wave stay.)
  auto kc_row = [&](int t, int u, int) __attribute__((always_inline)) -> int { return (t + NPT * u) >> 3; };
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6;
  const bool producer = wave >= 4;
  const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;  // index within the role's threads
  const int lane = tid & 63;
  int bx, by;
  const int Meff = (RL && a.mcount) ? min(a.M, *a.mcount) : a.M;  // uniform per launch
  // (tile order over the LIVE tile rows only: see gemm_x3b_kernel)
  const int gy_live = min(grid_y, (Meff + TM - 1) / TM);
  if (blk_y * grid_x + blk_x >= grid_x * gy_live) return;
  tile_order(blk_y * grid_x + blk_x, grid_x, gy_live, a.tile_gm, bx, by);
  const int n0 = bx * TN, m0 = by * TM, z = blk_z;
  if (m0 >= Meff) return;
  const int Kc = a.kcount ? max(0, min(a.seg[0].K, *a.kcount)) : 0;
  int steps_total = a.steps_total, steps_per_split = a.steps_per_split;
  if (a.kcount) {  // K is known only on the device: partition the k-steps here
    steps_total = (Kc + BK - 1) / BK;
    steps_per_split = (steps_total + grid_z - 1) / grid_z;
  }
  const int s_lo = z * steps_per_split;
  int s_hi = s_lo + steps_per_split;
  if (s_hi > steps_total) s_hi = steps_total;
  const int s_last = s_hi - 1;
  constexpr int CT_LD = TN + 4;   // epilogue C tile in LDS
  // 2xFP16: operands are brought into the middle of the fp16 range by exact power-of-two factors (scalar loads: they do not touch
  // the vector-memory counter of the staged loads) so that the lo pieces of all but negligibly small entries are normal fp16 numbers
  float f16_sa = 1.f, f16_sb = 1.f;
  if constexpr (F16) {
    if (a.a_scale) f16_sa = *a.a_scale;
    if (a.b_scale) f16_sb = *a.b_scale;
  }
#ifdef SSC_X3W_STAMP
  const bool stamp_on = (int)blockIdx.x == g_stamp_wg && (wave == 0 || wave == 4);
  unsigned long long* stamp_lds = reinterpret_cast<unsigned long long*>(lds + 2 * STAGE) + (wave == 4 ? 0 : 128);
  if (stamp_on && lane == 0) { stamp_lds[126] = __builtin_readcyclecounter(); stamp_lds[127] = wall_clock64(); }
  const unsigned long long stamp_entry = wall_clock64();   // g_stamp_wg == -2: every workgroup reports entry / exit wall clock
#endif

  if (producer) {
    // ================================ producer waves ================================
    f32x4 ra[PF][NA], rb[PF][NB];
    unsigned oka[PF], okb[PF];
    bool fullk[PF];   // uniform: the set's k-step lies wholly inside its K segment (no chunk needs zeroing)
    // Addresses: a UNIFORM base per operand and k-step (scalar registers: segment base + k0, recomputed from the cursor by
    // scalar arithmetic) plus a 32-bit byte offset per chunk that only changes with the segment (`global_load ... v_off,
    // s[base]`): the per-k-step address work of a thread is one select per chunk instead of two 64-bit vector adds (the stamps
    // of tools/x3w_stamp.py put 0.3-0.5 us per k-step into pointer arithmetic + issue).  Operand spans < 4 GB: checked on
    // the host (x3w_span_ok).
    unsigned oa[NA], ob[NB];  // offset of the chunk at a k-step inside the segment
    unsigned za[NA], zb[NB];  // offset used when the chunk lies past the end of K (same row / column, first k of the step: in range)
    int ia[2][NA], ib[2][NB];   // KG: gathered k-row numbers of the next step, buffer = parity of that step (relative to s_lo)
    Cursor cur;
    int s_ld = s_lo;
    // 2xFP16: the current segment's operands come as pre-split planes (uniform; KSeg::A16 / B16).  Same addressing as fp32 - the
    // cursor's base and leading dimension are swapped - and no chunk is ever masked (the planes' K padding holds zeros).
    bool pa = false, pb = false;
    auto planes_of_segment = [&]() __attribute__((always_inline)) {
      if constexpr (F16) {
        const KSeg& sg = a.seg[cur.seg];
        pa = sg.A16 != nullptr; pb = sg.B16 != nullptr;
        if (pa) { cur.A = sg.A16; cur.lda = sg.lda16; }
        if (pb) { cur.B = sg.B16; cur.ldb = sg.ldb16; }
      }
    };
    // chunk idx: k-contiguous operand -> (row idx>>3, k 4*(idx&7)); m/n-contiguous -> (k idx/Q, column 4*(idx%Q))
    auto base_ptrs = [&]() __attribute__((always_inline)) {
      const unsigned lda4 = (unsigned)cur.lda * 4u, ldb4 = (unsigned)cur.ldb * 4u;
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int idx = tid + NPT * u;
        if constexpr (A_KC) {
          int r = min(m0 + kc_row(tid, u, NA), Meff - 1);
          if constexpr (RL) { if (a.arows) r = a.arows[r]; }
          za[u] = (unsigned)r * lda4;
          oa[u] = za[u] + 16u * (idx & 7);
        } else {
          za[u] = 4u * (unsigned)min(m0 + 4 * (idx % QA), Meff - 4);
          if constexpr (KG) {   // oa = the gathered row NUMBER of the current step (the offset is formed at issue time)
            int kr = cur.k0 + idx / QA;
            kr = a.karows[min(kr, cur.K - 1)];   // KG kernels are launched with BOTH lists (launch())
            oa[u] = (unsigned)kr;
          } else {
            oa[u] = (unsigned)(idx / QA) * lda4 + za[u];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = tid + NPT * u;
        if constexpr (B_KC) {
          zb[u] = (unsigned)min(n0 + kc_row(tid, u, NB), a.N - 1) * ldb4;
          ob[u] = zb[u] + 16u * (idx & 7);
        } else {
          zb[u] = 4u * (unsigned)min(n0 + 4 * (idx % QB), a.N - 4);
          if constexpr (KG) {
            int kr = cur.k0 + idx / QB;
            kr = a.kbrows[min(kr, cur.K - 1)];
            ob[u] = (unsigned)kr;
          } else {
            ob[u] = (unsigned)(idx / QB) * ldb4 + zb[u];
          }
        }
      }
    };
    // KG: the row numbers of the k-step after the cursor's (clamped) go to buffer `buf` = parity of that step.  Two buffers with
    // compile-time roles (the loop is unrolled by two): with ONE array that was copied into the offsets a step later, hipcc
    // resolved the loop-carried copy right behind the list loads - `s_waitcnt vmcnt(0)` in the middle of every k-step, a whole
    // memory round trip in the producers' issue phase (stamps: 0.55 us against 0.25 us without lists).
    auto prefetch_rows = [&](int buf) __attribute__((always_inline)) {
      if constexpr (KG) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          const int kr = min(cur.k0 + BK + (tid + NPT * u) / QA, cur.K - 1);
          ia[buf][u] = a.karows[kr];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int kr = min(cur.k0 + BK + (tid + NPT * u) / QB, cur.K - 1);
          ib[buf][u] = a.kbrows[kr];
        }
      }
    };
    auto step_ptrs = [&](int how, int buf) __attribute__((always_inline)) {
      if constexpr (KG) {
        // gathered k-rows (single segment by construction: how is 0 or 1): the row numbers follow the lists, the base stays at
        // the segment start.  No base_ptrs() here: a second site that stores list entries into the same arrays makes hipcc
        // sink the two stores into one store through a pointer phi, which keeps the arrays in scratch memory (a scratch reload
        // is a vector-memory operation and its wait drains the staged loads).
        if (how != 0) {
#pragma unroll
          for (int u = 0; u < NA; ++u) oa[u] = (unsigned)ia[buf][u];
#pragma unroll
          for (int u = 0; u < NB; ++u) ob[u] = (unsigned)ib[buf][u];
          prefetch_rows(buf ^ 1);
        }
      } else {
        if (how == 2) base_ptrs();   // offsets only change with the segment; the bases follow the cursor
      }
    };
    // Branch-free (no branch may surround a staged load: see Stage::load_ptrs): every k-step takes the same form - a chunk
    // past the end of K reads the first k of the step instead (always in range) and is zeroed at LDS-store time.
    auto issue_loads = [&](f32x4 (&xa)[NA], f32x4 (&xb)[NB], unsigned& ma, unsigned& mb, bool& full) __attribute__((always_inline)) {
      ma = pa ? 0x80000000u : 0u;   // bit 31: this set's tile is a plane tile (travels with the register set to its LDS store)
      mb = pb ? 0x80000000u : 0u;
      full = cur.k0 + BK <= cur.K;
      // uniform bases of this k-step
      const float* sa = cur.A + (A_KC ? (size_t)cur.k0 : (KG ? (size_t)0 : (size_t)cur.k0 * cur.lda));
      const float* sb = cur.B + (B_KC ? (size_t)cur.k0 : (KG ? (size_t)0 : (size_t)cur.k0 * cur.ldb));
      unsigned ea[NA], eb[NB];
      // k-contiguous operands: chunk u of this thread starts at k0 + 4*(tid & 7) for every u (idx = tid + NPT u)
      const bool kc_in = cur.k0 + 4 * (tid & 7) < cur.K;
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int idx = tid + NPT * u;
        if constexpr (A_KC) {
          ea[u] = (kc_in || pa) ? oa[u] : za[u];
          ma |= (kc_in ? 1u : 0u) << u;
        } else {
          const bool in = cur.k0 + idx / QA < cur.K;
          if constexpr (KG) ea[u] = __umul24(oa[u], (unsigned)cur.lda * 4u) + za[u];   // 24-bit multiply-add: the 64-bit form (v_mad_u64_u32) took a register PAIR as addend whose undefined high half hipcc placed on the destination of a list load in flight - a false dependency and a vmcnt wait in front of every staged load
          else ea[u] = in ? oa[u] : za[u];
          ma |= (in ? 1u : 0u) << u;
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = tid + NPT * u;
        if constexpr (B_KC) {
          eb[u] = (kc_in || pb) ? ob[u] : zb[u];
          mb |= (kc_in ? 1u : 0u) << u;
        } else {
          const bool in = cur.k0 + idx / QB < cur.K;
          if constexpr (KG) eb[u] = __umul24(ob[u], (unsigned)cur.ldb * 4u) + zb[u];
          else eb[u] = in ? ob[u] : zb[u];
          mb |= (in ? 1u : 0u) << u;
        }
      }
#ifdef SSC_X3W_AUDIT
#pragma unroll
      for (int u = 0; u < NA; ++u) SSC_AUDIT_TOUCH(a.member, cur.seg, 0, (reinterpret_cast<const char*>(sa) - reinterpret_cast<const char*>(cur.A)) + (long long)ea[u]);
#pragma unroll
      for (int u = 0; u < NB; ++u) SSC_AUDIT_TOUCH(a.member, cur.seg, 1, (reinterpret_cast<const char*>(sb) - reinterpret_cast<const char*>(cur.B)) + (long long)eb[u]);
#endif
#pragma unroll
      for (int u = 0; u < NA; ++u) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(xa[u]) : "v"(ea[u]), "s"(sa) : "memory");
#pragma unroll
      for (int u = 0; u < NB; ++u) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(xb[u]) : "v"(eb[u]), "s"(sb) : "memory");
    };
    // wait until at most YOUNGER of the hand-issued loads are outstanding; "+v" pins every use of this stage's registers
    // behind the wait (cdna_hip_programming.md 5.7)
    auto pin = [&](f32x4 (&xa)[NA], f32x4 (&xb)[NB]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NA; ++u) asm volatile("" : "+v"(xa[u])::"memory");
#pragma unroll
      for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(xb[u])::"memory");
    };
    auto advance = [&](int buf) __attribute__((always_inline)) {   // buf: parity of the step the cursor moves TO (KG row buffers)
      const int how = cur.advance(a, s_ld >= s_last);
      s_ld = min(s_ld + 1, s_last);
      if (how == 2) planes_of_segment();
      step_ptrs(how, buf);
    };
    auto put_chunk = [&](unsigned char* p, int plane, f32x4 v, bool ok, float scale) __attribute__((always_inline)) {
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (F16) {
        u32x2 hi, lo;
        v = v * scale;
        split4_f16(v, hi, lo);
        *reinterpret_cast<u32x2*>(p) = hi;
        *reinterpret_cast<u32x2*>(p + plane) = lo;
      } else {
        u32x2 hi, mid, lo;
        split4(v, hi, mid, lo);
        *reinterpret_cast<u32x2*>(p) = hi;
        *reinterpret_cast<u32x2*>(p + plane) = mid;
        *reinterpret_cast<u32x2*>(p + 2 * plane) = lo;
      }
    };
    // a chunk of a plane tile: 16 bytes = 8 consecutive k of ONE plane (chunks 0-3 of a row's k-block: hi, 4-7: lo)
    auto put_pre = [&](unsigned char* row, int plane, int q, f32x4 v) __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(row + (q & 3) * 16 + (q >> 2) * plane) = v;
    };
    auto put_planes_impl = [&](unsigned char* st, const f32x4 (&xa)[NA], const f32x4 (&xb)[NB], unsigned ma, unsigned mb, const bool all) __attribute__((always_inline)) {
      bool a_pre = false, b_pre = false;   // uniform: scalar branches around VALU + LDS work only
      if constexpr (F16 && A_KC) a_pre = (ma >> 31) != 0;
      if constexpr (F16 && B_KC) b_pre = (mb >> 31) != 0;
      if (a_pre) {
#pragma unroll
        for (int u = 0; u < NA; ++u) put_pre(st + kc_row(tid, u, NA) * PL_ROW_B, PLA, (tid + NPT * u) & 7, xa[u]);
      } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          const int idx = tid + NPT * u;
          put_chunk(st + (A_KC ? kc_row(tid, u, NA) * PL_ROW_B + (idx & 7) * 8 : (idx / QA) * MCA + (idx % QA) * 8), PLA, xa[u], all || ((ma >> u) & 1u), f16_sa);
        }
      }
      if (b_pre) {
#pragma unroll
        for (int u = 0; u < NB; ++u) put_pre(st + NPL * PLA + kc_row(tid, u, NB) * PL_ROW_B, PLB, (tid + NPT * u) & 7, xb[u]);
      } else {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int idx = tid + NPT * u;
          put_chunk(st + NPL * PLA + (B_KC ? kc_row(tid, u, NB) * PL_ROW_B + (idx & 7) * 8 : (idx / QB) * MCB + (idx % QB) * 8), PLB, xb[u], all || ((mb >> u) & 1u), f16_sb);
        }
      }
    };
    // a k-step inside its segment (uniform test: a scalar branch around VALU + LDS work only) needs no per-chunk select
    auto put_planes = [&](unsigned char* st, const f32x4 (&xa)[NA], const f32x4 (&xb)[NB], unsigned ma, unsigned mb, bool full) __attribute__((always_inline)) {
      if (full) put_planes_impl(st, xa, xb, ma, mb, true);
      else put_planes_impl(st, xa, xb, ma, mb, false);
    };
    constexpr int NL = NA + NB;

    if (s_lo < s_hi) {
      cur.init(a, s_lo);
      if (a.kcount) { cur.K = Kc; cur.left = steps_total - 1 - cur.k0 / BK; }
      planes_of_segment();
      base_ptrs();
      prefetch_rows(1);   // rows of step s_lo + 1
      // the first PF tiles are requested back to back (one exposed memory latency, not two); afterwards set (r+1) % PF
      // holds tile s_lo + r + 1 when iteration r starts
      SSC_STAMP(120);   // setup done, nothing requested yet
      issue_loads(ra[0], rb[0], oka[0], okb[0], fullk[0]);
#pragma unroll
      for (int j = 1; j < PF; ++j) {
        advance(j & 1);
        issue_loads(ra[j], rb[j], oka[j], okb[j], fullk[j]);
      }
      SSC_STAMP(121);   // first PF tiles requested
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
      pin(ra[0], rb[0]);
      SSC_STAMP(122);   // first tile landed
      put_planes(lds, ra[0], rb[0], oka[0], okb[0], fullk[0]);
      advance(PF & 1);
      issue_loads(ra[0], rb[0], oka[0], okb[0], fullk[0]);
    }
    __syncthreads();
    // Iteration r = s + h - s_lo: tile r + 1 (register set (r + 1) % PF) has landed once at most PF - 1 younger tiles are
    // outstanding; it goes to LDS stage (r + 1) & 1 while the consumers work on stage r & 1, and its register set takes tile
    // r + 1 + PF.  The body is unrolled UNR = lcm(2, PF) times so that set and stage are compile-time.  All UNR parts always
    // run (the trip count is rounded up to a multiple of UNR; the consumers take the same number of barriers): ONE loop exit,
    // every path issues the same loads and waits - with a second exit hipcc resolves the register sets' phi at the exits with
    // copies of in-flight registers (seen with a mid-body break; caught by tools/check_staged_loads.py).  Loads past the end
    // re-read the last tile (clamped cursor) and are never stored.
    for (int s = s_lo; s < s_hi; s += UNR) {
#pragma unroll
      for (int h = 0; h < UNR; ++h) {
        const int j = (h + 1) % PF;   // (s + h - s_lo + 1) % PF: s - s_lo is a multiple of UNR here
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
        pin(ra[j], rb[j]);
        SSC_STAMP(4 * (s + h - s_lo));
        if (s + h + 1 < s_hi) put_planes(lds + ((h + 1) & 1) * STAGE, ra[j], rb[j], oka[j], okb[j], fullk[j]);   // VALU + LDS only
        SSC_STAMP(4 * (s + h - s_lo) + 1);
        advance((h + 1 + PF) & 1);   // the cursor moves to step (s + h - s_lo) + 1 + PF
        issue_loads(ra[j], rb[j], oka[j], okb[j], fullk[j]);
        SSC_STAMP(4 * (s + h - s_lo) + 2);
        __syncthreads();
        SSC_STAMP(4 * (s + h - s_lo) + 3);
      }
    }
    // drain the clamped tail loads with the staged registers pinned (see gemm_kernel)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < PF; ++j) pin(ra[j], rb[j]);
  } else {
  // ================================ consumer waves ================================
  constexpr int WNW = TN / 64;  // consumer waves along N
  const int wm = wave / WNW, wn = wave % WNW;
  const int half = lane >> 5, l31 = lane & 31;
  const int g16 = lane >> 4, i16 = lane & 15;
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  // fragment = 8 consecutive k (16*kk + 8*half ...) of tile row/column rc0 + (lane & 31) of one plane
  auto frag = [&](const unsigned char* plane, bool kc, int mc_row_b, int rc0, int kk) __attribute__((always_inline)) -> bf16x8 {
    if (kc) {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(plane + (rc0 + l31) * PL_ROW_B + kk * 32 + half * 16));
    } else {
      // transposed read: lane 4q+p of a 16-lane group addresses k-row q, columns 4p..4p+3 of the group's 4 x 16 block and
      // receives column (lane & 15), k-rows 0..3; the second read takes the next 4 k-rows
      const unsigned char* p = plane + (16 * kk + 8 * (g16 >> 1) + (i16 >> 2)) * mc_row_b + (rc0 + 16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * mc_row_b));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      return __builtin_bit_cast(bf16x8, v);
    }
  };
#define SSC_X3W_MFMA(FA, FB, PA, PB)                                                                                 \
  _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[mi][PA], FB[ni][PB], acc[mi][ni], 0, 0, 0);
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define SSC_X3W_MFMA_H(FA, FB, PA, PB)                                                                               \
  _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = \
      __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, FA[mi][PA]), __builtin_bit_cast(f16x8, FB[ni][PB]), acc[mi][ni], 0, 0, 0);
  auto compute = [&](const unsigned char* st) __attribute__((always_inline)) {
    if constexpr (F16) {
      // planes: 0 = hi, 1 = lo.  Three partial products per accumulator, smallest first (lo*hi, hi*lo, hi*hi); the second k-half's
      // fragments are requested under the first half's MFMAs
      bf16x8 fa[2][2][2], fb[2][2][2];  // [kk][tile][plane] (bit patterns of fp16 pairs)
      auto rda = [&](int kk, int pl) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[kk][t][pl] = frag(st + pl * PLA, A_KC, MCA, wm * 64 + t * 32, kk);
      };
      auto rdb = [&](int kk, int pl) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[kk][t][pl] = frag(st + NPL * PLA + pl * PLB, B_KC, MCB, wn * 64 + t * 32, kk);
      };
      rda(0, 1); rdb(0, 0); rda(0, 0); rdb(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      SSC_X3W_MFMA_H(fa[0], fb[0], 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      rda(1, 1); rdb(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      SSC_X3W_MFMA_H(fa[0], fb[0], 0, 1)
      __builtin_amdgcn_sched_barrier(0);
      rda(1, 0); rdb(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      SSC_X3W_MFMA_H(fa[0], fb[0], 0, 0)
      SSC_X3W_MFMA_H(fa[1], fb[1], 1, 0)
      SSC_X3W_MFMA_H(fa[1], fb[1], 0, 1)
      SSC_X3W_MFMA_H(fa[1], fb[1], 0, 0)
      return;
    }
    bf16x8 fa[2][2][3], fb[2][2][3];  // [kk][tile][plane]
    auto rda = [&](int kk, int pl) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) fa[kk][t][pl] = frag(st + pl * PLA, A_KC, MCA, wm * 64 + t * 32, kk);
    };
    auto rdb = [&](int kk, int pl) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) fb[kk][t][pl] = frag(st + NPL * PLA + pl * PLB, B_KC, MCB, wn * 64 + t * 32, kk);
    };
    // six partial products per accumulator, smallest first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi); the reads are
    // ordered by first use and the first fragments of the second k-half are requested under the first half's MFMAs
    rda(0, 2); rdb(0, 0); rda(0, 0); rdb(0, 2); rda(0, 1); rdb(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    SSC_X3W_MFMA(fa[0], fb[0], 2, 0)
    SSC_X3W_MFMA(fa[0], fb[0], 0, 2)
    SSC_X3W_MFMA(fa[0], fb[0], 1, 1)
    __builtin_amdgcn_sched_barrier(0);
    rda(1, 2); rdb(1, 0); rda(1, 0); rdb(1, 2);
    __builtin_amdgcn_sched_barrier(0);
    SSC_X3W_MFMA(fa[0], fb[0], 1, 0)
    SSC_X3W_MFMA(fa[0], fb[0], 0, 1)
    SSC_X3W_MFMA(fa[0], fb[0], 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    rda(1, 1); rdb(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    SSC_X3W_MFMA(fa[1], fb[1], 2, 0)
    SSC_X3W_MFMA(fa[1], fb[1], 0, 2)
    SSC_X3W_MFMA(fa[1], fb[1], 1, 1)
    SSC_X3W_MFMA(fa[1], fb[1], 1, 0)
    SSC_X3W_MFMA(fa[1], fb[1], 0, 1)
    SSC_X3W_MFMA(fa[1], fb[1], 0, 0)
  };

  __syncthreads();
  for (int s = s_lo; s < s_hi; s += UNR) {   // UNR barriers per trip, like the producers
#pragma unroll
    for (int h = 0; h < UNR; ++h) {
      SSC_STAMP(4 * (s + h - s_lo));
      if (s + h < s_hi) compute(lds + (h & 1) * STAGE);
      SSC_STAMP(4 * (s + h - s_lo) + 1);
      __syncthreads();
      SSC_STAMP(4 * (s + h - s_lo) + 2);
    }
  }
#undef SSC_X3W_MFMA
#undef SSC_X3W_MFMA_H

  // accumulators -> fp32 C tile in LDS (the operand stages are free after the last barrier): row-major [TM][TN + 4]
  float* ct = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[(wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CT_LD + wn * 64 + ni * 32 + l31] = acc[mi][ni][r];
  }  // role split

  // ---- epilogue, all 8 waves: C tile rows leave LDS as 16-byte stores, 1 KB contiguous per 64 lanes (the accumulator
  // layout itself would give 64 four-byte stores per lane in two 128-B pieces each: store-issue-bound, ~8 us per launch)
  __syncthreads();
  if constexpr (TM == 128 && TN == 128 && A_KC && B_KC && !KG) {
    if (a.topk) {
      // Vocabulary head of a decode step without the logits: per (row, this 128-column tile) the log-sum-exp partials and the two
      // best columns.  Four threads per tile row, 32 columns each, combined by shuffles; order: value descending, column ascending.
      const float* ct = reinterpret_cast<const float*>(lds);
      const int t = (int)threadIdx.x;
      if (t < 512) {
        const int row = t >> 2, part = t & 3;
        const int grow = m0 + row;
        const int ntn = (a.N + TN - 1) / TN;
        float mx = -INFINITY, v0 = -INFINITY, v1 = -INFINITY;
        int i0 = -1, i1 = -1;
        const float inv = F16 ? 1.0f / (f16_sa * f16_sb) : 1.0f;
        // this thread's 32 columns: eight 16-byte LDS reads, kept in registers for both passes
        float x[32];
        const bool wide_b = a.bias && !(reinterpret_cast<uintptr_t>(a.bias) & 15) && n0 + TN <= a.N;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 c4 = *reinterpret_cast<const float4*>(&ct[row * CT_LD + part * 32 + 4 * j]);
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          const int gc = n0 + part * 32 + 4 * j;
          if (wide_b) b4 = *reinterpret_cast<const float4*>(a.bias + gc);
          else if (a.bias) { b4.x = gc < a.N ? a.bias[gc] : 0.f; b4.y = gc + 1 < a.N ? a.bias[gc + 1] : 0.f; b4.z = gc + 2 < a.N ? a.bias[gc + 2] : 0.f; b4.w = gc + 3 < a.N ? a.bias[gc + 3] : 0.f; }
          x[4 * j] = c4.x * inv + b4.x; x[4 * j + 1] = c4.y * inv + b4.y; x[4 * j + 2] = c4.z * inv + b4.z; x[4 * j + 3] = c4.w * inv + b4.w;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int gcol = n0 + part * 32 + j;
          if (gcol < a.N) {
            mx = fmaxf(mx, x[j]);
            if (x[j] > v0) { v1 = v0; i1 = i0; v0 = x[j]; i0 = gcol; }
            else if (x[j] > v1) { v1 = x[j]; i1 = gcol; }
          }
        }
        // the row's maximum over the four parts, then the sum of exp(x - max) with that common maximum (hardware exp2: the sum only
        // feeds the row's log-sum-exp, ~1e-7 relative)
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int gcol = n0 + part * 32 + j;
          if (gcol < a.N) se += __builtin_amdgcn_exp2f((x[j] - mx) * 1.4426950408889634f);
        }
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) se += __shfl_xor(se, o, 64);
        // merge the parts' pairs (each sorted; a lower part holds lower columns)
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
          const float w0 = __shfl_xor(v0, o, 64), w1 = __shfl_xor(v1, o, 64);
          const int j0 = __shfl_xor(i0, o, 64), j1 = __shfl_xor(i1, o, 64);
          auto before = [](float xa, int ia, float xb, int ib) { return ib < 0 || (ia >= 0 && (xa > xb || (xa == xb && ia < ib))); };
          float r0, r1; int k0, k1;
          if (before(v0, i0, w0, j0)) {
            r0 = v0; k0 = i0;
            if (before(v1, i1, w0, j0)) { r1 = v1; k1 = i1; } else { r1 = w0; k1 = j0; }
          } else {
            r0 = w0; k0 = j0;
            if (before(v0, i0, w1, j1)) { r1 = v0; k1 = i0; } else { r1 = w1; k1 = j1; }
          }
          v0 = r0; i0 = k0; v1 = r1; i1 = k1;
        }
        if (part == 0 && grow < Meff) {
          const int rr = (RL && a.crows) ? a.crows[grow] : grow;
          float* rec = a.topk + ((size_t)rr * ntn + bx) * 6;
          rec[0] = mx; rec[1] = se; rec[2] = v0; rec[3] = __int_as_float(i0); rec[4] = v1; rec[5] = __int_as_float(i1);
        }
      }
      return;
    }
  }
  {
    const float* ct = reinterpret_cast<const float*>(lds);
    float* out = a.out + (size_t)z * a.slab_stride;
    const bool wide = !(a.ldo & 3) && !(a.N & 3) && !(a.slab_stride & 3) && !(reinterpret_cast<uintptr_t>(a.out) & 15) &&
                      !(reinterpret_cast<uintptr_t>(a.bias) & 15);
    constexpr int CPR = TN / 4;  // float4 chunks per tile row
#pragma unroll
    for (int i = 0; i < (TM * CPR + NTHR - 1) / NTHR; ++i) {
      const int c = (int)threadIdx.x + NTHR * i;
      if (TM * CPR % NTHR != 0 && c >= TM * CPR) break;
      const int row = c / CPR, col = 4 * (c % CPR);
      const int grow = m0 + row, gcol = n0 + col;
      if (grow >= Meff || gcol >= a.N) continue;
      const int rr = (RL && a.crows) ? a.crows[grow] : grow;
      float4 v = *reinterpret_cast<const float4*>(&ct[row * CT_LD + col]);
      if constexpr (F16) { const float inv = 1.0f / (f16_sa * f16_sb); v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv; }
      float* dst = out + (size_t)rr * a.ldo + gcol;
      if (wide) {
        if (a.bias) { const float4 b4 = *reinterpret_cast<const float4*>(a.bias + gcol); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
        if (a.accumulate) { const float4 o4 = *reinterpret_cast<const float4*>(dst); v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w; }
        if (a.store_wt) {
          const f32x4 v4 = {v.x, v.y, v.z, v.w};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v4) : "memory");
        } else {
          *reinterpret_cast<float4*>(dst) = v;
        }
      } else {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (gcol + q < a.N) dst[q] = e[q] + (a.bias ? a.bias[gcol + q] : 0.f) + (a.accumulate ? dst[q] : 0.f);
        }
      }
    }
  }
#ifdef SSC_X3W_STAMP
  if (stamp_on && lane == 0) { stamp_lds[124] = __builtin_readcyclecounter(); stamp_lds[125] = wall_clock64(); }
  __syncthreads();
  if ((int)blockIdx.x == g_stamp_wg && threadIdx.x < 256 && g_stamp_ptr)
    g_stamp_ptr[threadIdx.x] = reinterpret_cast<const unsigned long long*>(lds + 2 * STAGE)[threadIdx.x];
  if (g_stamp_wg == -2 && threadIdx.x == 0 && g_stamp_ptr && blockIdx.x < 1024) {
    g_stamp_ptr[2 * blockIdx.x] = stamp_entry;
    g_stamp_ptr[2 * blockIdx.x + 1] = wall_clock64();
  }
#endif
}


// Launch form: a GROUP of up to SSC_GROUP_MAX independent products (own operands, K and output each) in one grid, e.g.
// the three hidden-state gradients BPTT carries to step t-1 (each alone covers 150 of the 256 CUs and pays its own
// ramp); a single product is a group of one.  Workgroup w belongs to the product p with first[p] <= w < first[p+1];
// inside it the usual (tile, split) decomposition applies.
constexpr int SSC_GROUP_MAX = 6;
struct KGroup {
  KArgs a[SSC_GROUP_MAX];
  int n;
  int first[SSC_GROUP_MAX + 1];
  int gx[SSC_GROUP_MAX], gy[SSC_GROUP_MAX], gz[SSC_GROUP_MAX];
};
template <bool A_KC, bool B_KC, bool KG, int TM, int TN, int PF, int NPW = 4, bool F16 = false>
__global__ __launch_bounds__(256 + 64 * NPW, (F16 && NPW == 4) ? 4 : 1) void gemm_x3w_kernel(const KGroup g) {
  const int w = blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < SSC_GROUP_MAX; ++i)
    if (i < g.n && w >= g.first[i]) p = i;   // workgroup-uniform
  const int local = w - g.first[p];
  const int gx = g.gx[p], gy = g.gy[p], gz = g.gz[p];
  const int x = local % gx, yz = local / gx;
  x3w_body<A_KC, B_KC, KG, TM, TN, PF, NPW, F16>(g.a[p], x, yz % gy, yz / gy, gx, gy, gz);
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N,
                                    float* __restrict__ C, int ldc, const float* __restrict__ bias, int accumulate,
                                    const int* __restrict__ mcount, const int* __restrict__ crows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (mcount) M = min(M, *mcount);
  size_t total = (size_t)M * N;
  if (i >= total) return;
  int row = (int)(i / N), col = (int)(i % N);
  float v = 0.f;
  for (int s0 = 0; s0 < nslab; s0 += 8) {  // fixed summation order, 8 loads in flight
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slabs[(size_t)min(s0 + u, nslab - 1) * slab_stride + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += (s0 + u < nslab) ? t[u] : 0.f;
  }
  if (bias) v += bias[col];
  if (crows) row = crows[row];
  float* p = C + (size_t)row * ldc + col;
  if (accumulate) v += *p;
  *p = v;
}

typedef void (*gemm_fn)(const KArgs);

// ---- optional in-situ profiling (bench.py roofline leg): hipEvent pair around every GEMM launch ------
struct ProfMember { int M, N, K, slot_m, slot_k; };   // slot_*: index of the device-side row / k-row count read back behind the launch, or -1
struct ProfRec {
  hipEvent_t e0, e1;
  int kind, M, N, K, splits;
  double bytes, flops;   // algorithmic: every operand and the result once, 2 M N K - summed over the members of a grouped launch
  ProfMember mem[8];     // the members' extents; products with device-side row compaction (m_count / k_count) are priced at
  int nmem;              // collect time on the rows they really processed (true flops, not the nominal T*B extent)
};
int* g_prof_counts = nullptr;   // pinned host array: one int per compacted extent of a profiled launch
int g_prof_slots = 0;
constexpr int PROF_SLOTS = 8192;
inline int prof_count_slot(const int* dev_count, hipStream_t st) {
  if (!dev_count || !g_prof_counts || g_prof_slots >= PROF_SLOTS) return -1;
  const int slot = g_prof_slots++;
  g_prof_counts[slot] = -1;
  if (hipMemcpyAsync(&g_prof_counts[slot], dev_count, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
  return slot;
}
inline void prof_desc(ProfRec* rec, const ssc_gemm_desc* d, hipStream_t st) {   // adds one product to a record
  int K = 0;
  for (int i = 0; i < d->nseg; ++i) K += d->seg[i].K;
  if (rec->nmem < 8) {
    ProfMember& m = rec->mem[rec->nmem++];
    m.M = d->M; m.N = d->N; m.K = K;
    m.slot_m = prof_count_slot(d->m_count, st);
    m.slot_k = prof_count_slot(d->k_count, st);
  }
}
constexpr int PROF_MAX = 4096;
ProfRec* g_prof = nullptr;
int g_prof_n = 0;
bool g_prof_on = false;

// Write-through (sc1) output stores of the wave-specialised kernels: the 16 MB of split-K slabs a gate product leaves behind
// reach memory while the kernel is still running instead of being written back at the kernel boundary (rocprof timeline:
// 1-3.5 us between such a kernel's end and its consumer's start).  Train step 8.76 -> 8.57-8.65 ms (same-box A/B, twice).  Round
// 2's first try of this showed no difference: the 8-wave kernels were slower then and their own tail hid the write-back.
int g_store_wt = ssc_env_int("SSC_STORE_WT", 1);
int g_tile_gm = ssc_env_int("SSC_TILE_GM", 8);   // tile_order(): tile rows per group (0 = row-major)
int build_args(const ssc_gemm_desc* d, KArgs& k) {
  if (!d || d->nseg < 1 || d->nseg > SSC_MAX_SEG || d->M <= 0 || d->N <= 0) return SSC_EINVAL;
  k.tile_gm = g_tile_gm;
  k.store_wt = g_store_wt;
  k.member = 0;
  k.nseg = d->nseg;
  k.M = d->M;
  k.N = d->N;
  k.mcount = d->m_count; k.arows = d->a_rows; k.crows = d->c_rows;
  k.kcount = d->k_count; k.karows = d->ka_rows; k.kbrows = d->kb_rows;
  k.a_scale = d->a_scale; k.b_scale = d->b_scale;
  k.topk = d->topk_part;
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
    k.seg[i].avec = (ssc_aligned16(s.A) && !(s.lda & 3) && !((d->a_kc ? s.K : d->M) & 3)) ? 1 : 0;
    k.seg[i].bvec = (ssc_aligned16(s.B) && !(s.ldb & 3) && !((d->b_kc ? s.K : d->N) & 3)) ? 1 : 0;
    // pre-split planes (optional; read by the 2xFP16 form only): k-contiguous operands, whole 32-k blocks, 16-byte rows
    const int kp = ssc_cdiv(s.K, BK) * BK;
    if (s.A16 && (!d->a_kc || s.lda16 < kp || (s.lda16 & 3) || !ssc_aligned16(s.A16))) return SSC_EINVAL;
    if (s.B16 && (!d->b_kc || s.ldb16 < kp || (s.ldb16 & 3) || !ssc_aligned16(s.B16))) return SSC_EINVAL;
    k.seg[i].A16 = static_cast<const float*>(s.A16); k.seg[i].lda16 = s.lda16;
    k.seg[i].B16 = static_cast<const float*>(s.B16); k.seg[i].ldb16 = s.ldb16;
  }
  return SSC_OK;
}

template <int WM, int WN, int PF, bool VEC>
gemm_fn pick_layout(const ssc_gemm_desc* d) {
  if (d->a_kc && d->b_kc) return gemm_kernel<true, true, WM, WN, PF, VEC>;
  if (d->a_kc && !d->b_kc) return gemm_kernel<true, false, WM, WN, PF, VEC>;
  if (!d->a_kc && !d->b_kc) return gemm_kernel<false, false, WM, WN, PF, VEC>;
  return gemm_kernel<false, true, WM, WN, PF, VEC>;
}

// tile choice: 128x128 (2x2 MFMA tiles per wave) when both dimensions are large, else 64x64 with a deeper prefetch
// SSC_GEMM_MODE: "x3" (default) = 3xBF16 split kernel for NT products with 16 B/lane operands, "f32" = exact fp32 MFMA
int g_gemm_mode = -1;  // -1: take the default from the environment on first use
}
// A sequence-level call (ssc_train_fwd / _bwd, ssc_decode_*) runs under the numerics mode of ITS ssc_model_cfg (gemm_mode field):
// the calling thread's override, in force while the call issues its launches.  Two engines of one process can so differ.
thread_local int ssc_tls_gemm_mode = -1;
thread_local int ssc_tls_gemm_f16 = -1;   // ssc_model_cfg.gemm_mode 3: the wave-specialised 128x128 NT products of this call take the 2xFP16 form
namespace {
inline int gemm_mode() {
  if (ssc_tls_gemm_mode >= 0) return ssc_tls_gemm_mode;
  if (g_gemm_mode < 0) {
    const char* e = ssc_env_debug() ? getenv("SSC_GEMM_MODE") : nullptr;
    g_gemm_mode = (e && e[0] == 'f') ? 0 : 1;
  }
  return g_gemm_mode;
}
inline bool use_x3(const ssc_gemm_desc* d, bool vec) { return gemm_mode() == 1 && d->a_kc && d->b_kc && vec; }
// Tuning / diagnostic switches (include/ssc_debug.h: ssc_debug_set / ssc_debug_get; environment defaults in parentheses).
// They select between kernel forms that compute the same product; none is part of the product ABI.
// Large products (M, N >= 512): 0 = 64x64 kernels, 3 = always the 4-wave 128x128 3xBF16 kernel, 2 = always its wave-specialised
// form, 1 (default since round 2) = by grid size: wave-specialised from 768 workgroups on (three rounds of the chip: decode
// +8 % tokens/s, alternating same-box A/B), 4-wave below (two workgroups per CU even out grids of one or two rounds).  History: round 1 kept the
// wave-specialised form opt-in after ONE unexplained GPU memory fault in a long decode run.  Since then its loop
// structure, addressing and register discipline were rebuilt (branch-free staged loads, uniform loops, 12 waves, scalar bases)
// under an exact ISA gate, the same kernel family has run the grouped weight gradients of every train step, and the 3000-image
// decode of the fault ran clean with it twice (before and after the rebuild).  `ssc_debug_set("large_form", 3)` / SSC_X3B=3
// selects the 4-wave kernel everywhere.
int g_x3b = ssc_env_int("SSC_X3B", 1);
int g_x3_nbuf = 1;  // single LDS stage: 31 KB per workgroup -> four resident workgroups per CU (rocprof r01: 37 vs 43 us)
int g_x3_wide = 0;
int g_x3_pf = 2;  // tuning hook: 1 = 64x128 block tile for skinny (M <= 64, N >= 1024) 3xBF16 products
inline bool x3_wide(int M, int N) { return g_x3_wide && M <= 64 && N >= 1024; }
// 128x128 tiles from 65 rows on: a minibatch of 65-511 rows (C5: B = 128 per GPU) is MFMA-bound in 3xBF16, not HBM-bound - the
// 64-wide kernels streamed the weights once per 64 rows (B = 128: 16.2 -> 14.3 ms per train step, B = 256: 27.0 -> 22.9 ms with
// the wave-specialised 128x128 form, same box; 65-127 rows: one padded tile row still beats two 64-row passes, B = 96: 14.6 ->
// 12.5 ms).  "big_min_m" (SSC_BIG_MIN_M) = 512 restores the earlier behaviour.
int g_big_min_m = ssc_env_int("SSC_BIG_MIN_M", 65);
inline bool big_tile(int M, int N) { return M >= g_big_min_m && N >= 512; }
// M <= 64 with a wide N: 64x128 block tile (wave tile 32x64).  Every workgroup re-reads the whole A operand
// (the minibatch activations, from L2) for its K-range, and the CU-side load path (~24 GB/s per CU) is what these
// products run into first (rocprof r01: loads per CU saturate with A+B at BN=64), so halving the A re-reads per
// streamed weight byte matters more than occupancy.
int g_wide_min_n = 1024;
inline bool wide_tile(int M, int N) { return M <= 64 && N >= g_wide_min_n; }

typedef void (*group_fn)(const KGroup);
#ifdef SSC_X3W_AUDIT
static_assert(SSC_GROUP_MAX == 6, "g_audit");
long g_audit_launches = 0, g_audit_records = 0, g_audit_violations = 0;
struct AuditSummary { ~AuditSummary() { fprintf(stderr, "[x3w audit] %ld launches, %ld (member, segment, operand) ranges checked, %ld VIOLATIONS\n", g_audit_launches, g_audit_records, g_audit_violations); } } g_audit_summary;
void audit_begin() {
  long long init[6][SSC_MAX_SEG][2][2];
  for (auto& m : init) for (auto& sg : m) for (auto& op : sg) { op[0] = 0x7fffffffffffffffLL; op[1] = -0x7fffffffffffffffLL; }
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_audit), init, sizeof(init));
}
// highest entry + 1 of a device-side row list of *count entries (<= nominal), or `nominal` without a list
long audit_rows(const int* list, const int* count, int nominal) {
  int n = nominal;
  if (count) { (void)hipMemcpy(&n, count, sizeof(int), hipMemcpyDeviceToHost); if (n > nominal) n = nominal; if (n < 0) n = 0; }
  if (!list) return n;
  if (n == 0) return 0;
  int* h = (int*)malloc((size_t)n * sizeof(int));
  (void)hipMemcpy(h, list, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
  long mx = 0;
  for (int i = 0; i < n; ++i) if (h[i] + 1 > mx) mx = h[i] + 1;
  free(h);
  return mx;
}
void audit_end(const KGroup& g, bool a_kc, bool b_kc, const char* what, hipStream_t st) {
  (void)hipStreamSynchronize(st);
  long long got[6][SSC_MAX_SEG][2][2];
  (void)hipMemcpyFromSymbol(got, HIP_SYMBOL(g_audit), sizeof(got));
  ++g_audit_launches;
  for (int m = 0; m < g.n; ++m) {
    const KArgs& k = g.a[m];
    for (int sg = 0; sg < k.nseg; ++sg) {
      const long K = k.kcount ? audit_rows(nullptr, k.kcount, k.seg[sg].K) : k.seg[sg].K;
      for (int op = 0; op < 2; ++op) {
        const long long lo = got[m][sg][op][0], hi = got[m][sg][op][1];
        if (hi <= lo) continue;   // never touched (e.g. no k-step of this segment in any workgroup's range)
        const bool kc = op == 0 ? a_kc : b_kc;
        const long ld = op == 0 ? k.seg[sg].lda : k.seg[sg].ldb;
        const long width = op == 0 ? k.M : k.N;   // rows (k-contiguous) or columns (m/n-contiguous) of the operand
        long long span;
        if (kc) {
          const long rows = op == 0 ? audit_rows(k.arows, k.mcount, k.M) : width;
          span = rows > 0 ? ((long long)(rows - 1) * ld + k.seg[sg].K) * 4 : 0;
        } else {
          const long krows = audit_rows(op == 0 ? k.karows : k.kbrows, k.kcount, k.seg[sg].K);
          span = krows > 0 ? ((long long)(krows - 1) * ld + width) * 4 : (long long)width * 4;   // (an empty k range re-reads k-row 0)
        }
        (void)K;
        ++g_audit_records;
        if (lo < 0 || hi > span) {
          ++g_audit_violations;
          fprintf(stderr, "[x3w audit] VIOLATION %s member %d segment %d operand %c: touched bytes [%lld, %lld) of a span of %lld (M %d N %d K %d ld %ld)\n",
                  what, m, sg, op == 0 ? 'A' : 'B', lo, hi, span, k.M, k.N, k.seg[sg].K, ld);
        }
      }
    }
  }
}
#define SSC_AUDIT_BEGIN() audit_begin()
#define SSC_AUDIT_END(g, a_kc, b_kc, what, st) audit_end(g, a_kc, b_kc, what, st)
#else
#define SSC_AUDIT_BEGIN() do {} while (0)
#define SSC_AUDIT_END(g, a_kc, b_kc, what, st) do {} while (0)
#endif
// a single product as a group of one
inline void group_of_one(KGroup& g, const KArgs& k, dim3 grid) {
  g.a[0] = k;
  g.a[0].member = 0;
  g.n = 1;
  g.first[0] = 0;
  for (int i = 0; i < SSC_GROUP_MAX; ++i) {
    g.first[i + 1] = (int)(grid.x * grid.y * grid.z);
    g.gx[i] = (int)grid.x; g.gy[i] = (int)grid.y; g.gz[i] = (int)grid.z;
  }
}

// the wave-specialised kernels need more than the default 64 KB of LDS per workgroup: raise the limit once
int x3w_prepare() {
  // per DEVICE: the attribute belongs to the function's code object on the device it is set under - a process that drives two GPUs
  // (or a rank that changes its device) must set it on each
  static bool done_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SSC_EHIP;
  bool& done = done_dev[dev];
  if (done) return SSC_OK;
  group_fn big[8] = {gemm_x3w_kernel<true, true, false, 128, 128, 2>, gemm_x3w_kernel<true, false, false, 128, 128, 2>,
                     gemm_x3w_kernel<false, false, true, 128, 128, 1>, gemm_x3w_kernel<false, false, false, 128, 128, 2>,
                     gemm_x3w_kernel<true, true, false, 128, 128, 2, 8>, gemm_x3w_kernel<true, false, false, 128, 128, 2, 8>,
                     gemm_x3w_kernel<false, false, true, 128, 128, 1, 8>, gemm_x3w_kernel<false, false, false, 128, 128, 2, 8>};
  for (group_fn f : big)
    if (hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes<128, 128>()) != hipSuccess) return SSC_EHIP;
  if (hipFuncSetAttribute((const void*)gemm_x3w_kernel<true, true, false, 128, 128, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          x3w_lds_bytes_f16<128, 128>()) != hipSuccess)
    return SSC_EHIP;
  if (hipFuncSetAttribute((const void*)gemm_x3w_kernel<true, true, false, 128, 128, 2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          x3w_lds_bytes_f16<128, 128>()) != hipSuccess)
    return SSC_EHIP;
  group_fn skinny[6] = {gemm_x3w_kernel<true, true, false, 64, 256, 2>, gemm_x3w_kernel<true, false, false, 64, 256, 2>,
                        gemm_x3w_kernel<true, true, false, 64, 256, 3>, gemm_x3w_kernel<true, false, false, 64, 256, 3>,
                        gemm_x3w_kernel<true, true, false, 64, 256, 2, 8>, gemm_x3w_kernel<true, false, false, 64, 256, 2, 8>};
  for (group_fn f : skinny)
    if (hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes<64, 256>()) != hipSuccess) return SSC_EHIP;
  done = true;
  return SSC_OK;
}
int g_f16_npw = ssc_env_int("SSC_F16_NPW", 4);   // producer waves of the 2xFP16 kernel: 4 (default: two 8-wave workgroups per CU - 80 KB of LDS each - so that one's pipeline fill and epilogue run under the other's k-loop: 10000 x 4800 x 2400 843 -> 750 us, same box) | 8 (one 12-wave workgroup per CU)
int g_gemm_f16 = ssc_env_int("SSC_GEMM_F16", 0);   // process default of the 2xFP16 form when no sequence-level call's cfg is in force
int g_x3w_skinny = ssc_env_int("SSC_X3W_SKINNY", 1);  // 0 off, 1 NT and NN, 2 NN only (hook -11 / -12 / -13)
int g_x3w_min_n = ssc_env_int("SSC_X3W_MIN_N", 1024);   // narrower products do not fill the chip with 256-column tiles (rocprof: slower than the 64-wide kernels)
int g_gemm_group = ssc_env_int("SSC_GEMM_GROUP", 1);   // grouped launches of independent minibatch products
int g_dw_group = ssc_env_int("SSC_DW_GROUP", 1);       // grouped launches of the weight-gradient products (wave-specialised 128x128 form); 0 = one 4-wave launch per product
// k-steps of operand tiles in flight in the producers' registers of the 64x256 kernels: 2 (default) | 3.  Three (120 staged
// VGPRs, loop unrolled 6x) measured 2-6 % SLOWER on every gate product (40.5 -> 42.8 us at 64 x 4800 x 5648; train step 9.22 ->
// 9.45 ms): the k-loop is not short of bytes in flight.
int g_x3w_pf = ssc_env_int("SSC_X3W_PF", 2);
int g_x3w_npw = ssc_env_int("SSC_X3W_NPW", 8);   // producer waves of the 64x256 kernels (4 | 8)
int g_x3w_big_npw = ssc_env_int("SSC_X3W_BIG_NPW", 8);   // producer waves of the wave-specialised 128x128 kernels (4 | 8)
inline int x3w_big_threads() { return g_x3w_big_npw == 8 ? 768 : 512; }
inline group_fn x3w_big_fn(bool a_kc, bool b_kc, bool kg) {   // layouts NT, NN, TN (+ k-row gather lists)
  if (g_x3w_big_npw == 8)
    return (a_kc && b_kc) ? gemm_x3w_kernel<true, true, false, 128, 128, 2, 8>
           : a_kc         ? gemm_x3w_kernel<true, false, false, 128, 128, 2, 8>
           : kg           ? gemm_x3w_kernel<false, false, true, 128, 128, 1, 8>
                          : gemm_x3w_kernel<false, false, false, 128, 128, 2, 8>;
  return (a_kc && b_kc) ? gemm_x3w_kernel<true, true, false, 128, 128, 2>
         : a_kc         ? gemm_x3w_kernel<true, false, false, 128, 128, 2>
         : kg           ? gemm_x3w_kernel<false, false, true, 128, 128, 1>
                        : gemm_x3w_kernel<false, false, false, 128, 128, 2>;
}
inline int x3w_skinny_threads() { return g_x3w_npw == 8 && g_x3w_pf != 3 ? 768 : 512; }
inline group_fn x3w_skinny_fn(bool b_kc) {
  if (x3w_skinny_threads() == 768) return b_kc ? gemm_x3w_kernel<true, true, false, 64, 256, 2, 8> : gemm_x3w_kernel<true, false, false, 64, 256, 2, 8>;
  if (g_x3w_pf == 3) return b_kc ? gemm_x3w_kernel<true, true, false, 64, 256, 3> : gemm_x3w_kernel<true, false, false, 64, 256, 3>;
  return b_kc ? gemm_x3w_kernel<true, true, false, 64, 256, 2> : gemm_x3w_kernel<true, false, false, 64, 256, 2>;
}
inline bool x3w_skinny_shape(int M, int N) { return g_x3w_skinny && gemm_mode() == 1 && M <= 64 && N >= g_x3w_min_n; }
inline bool x3w_skinny(const ssc_gemm_desc* d, bool vec) {   // 2 = only where the weight matrix is [K][N] (backward dG W)
  return vec && d->a_kc && x3w_skinny_shape(d->M, d->N) && (g_x3w_skinny == 1 || !d->b_kc);
}

// the wave-specialised kernels address an operand as (uniform 64-bit base) + (32-bit byte offset per lane)
inline bool x3w_span_ok(const ssc_gemm_desc* d) {
  const size_t lim = (size_t)1 << 30;   // floats
  if (d->ka_rows || d->kb_rows)         // gathered k-rows: offset = row * (ld * 4) by a 24-bit multiply
    for (int i = 0; i < d->nseg; ++i)
      if (d->seg[i].lda >= (1 << 22) || d->seg[i].ldb >= (1 << 22) || d->seg[i].K >= (1 << 24)) return false;
  for (int i = 0; i < d->nseg; ++i) {
    const size_t K = (size_t)d->seg[i].K;
    if (K == 0) continue;
    const size_t sa = d->a_kc ? (size_t)(d->M - 1) * d->seg[i].lda + K : (K - 1) * d->seg[i].lda + d->M;
    const size_t sb = d->b_kc ? (size_t)(d->N - 1) * d->seg[i].ldb + K : (K - 1) * d->seg[i].ldb + d->N;
    if (sa >= lim || sb >= lim) return false;
    const size_t kp = (K + BK - 1) / BK * BK;
    if (d->seg[i].A16 && (size_t)(d->M - 1) * d->seg[i].lda16 + kp >= lim) return false;
    if (d->seg[i].B16 && (size_t)(d->N - 1) * d->seg[i].ldb16 + kp >= lim) return false;
  }
  return true;
}
inline bool x3w_group_member(const ssc_gemm_desc* d, bool vec) {
  return vec && d->a_kc && g_x3w_skinny && gemm_mode() == 1 && d->M <= 64 && d->N >= 64 && (g_x3w_skinny == 1 || !d->b_kc) &&
         x3w_span_ok(d);
}

int launch(const ssc_gemm_desc* d, KArgs& k, int splits, hipStream_t st) {
  k.steps_per_split = ssc_cdiv(k.steps_total, splits);
  bool vec = true;  // every segment of both operands must allow 16 B/lane loads, else the 4 B/lane kernel runs
  for (int i = 0; i < k.nseg; ++i) vec = vec && k.seg[i].avec && k.seg[i].bvec;
  const bool compact = k.mcount || k.arows || k.crows || k.kcount || k.karows || k.kbrows;
  if (compact) {  // device-side row compaction: gemm_x3b_kernel only
    if (!vec) return SSC_EALIGN;
    if (gemm_mode() != 1 || (!d->a_kc && d->b_kc)) return SSC_EINVAL;
    if ((k.kcount || k.karows || k.kbrows) && (k.nseg != 1 || d->a_kc || d->b_kc)) return SSC_EINVAL;
    if ((k.mcount || k.arows || k.crows) && !d->a_kc) return SSC_EINVAL;
  }
  if (k.topk && !(gemm_mode() == 1 && vec && d->a_kc && d->b_kc && splits == 1 && x3w_span_ok(d) && !k.kcount && !k.karows && !k.kbrows))
    return SSC_EINVAL;   // records instead of C exist in the wave-specialised 128x128 NT form only: never fall through to a form that would store C
  if (x3w_skinny(d, vec) && !compact && !k.topk && x3w_span_ok(d)) {  // M = minibatch against a wide weight matrix: 64 x 256 wave-specialised tile
    dim3 grid(ssc_cdiv(d->N, 256), ssc_cdiv(d->M, 64), splits);
    ProfRec* rec = nullptr;
    if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {
      rec = &g_prof[g_prof_n++];
      rec->bytes = rec->flops = 0.0; rec->nmem = 0;
      rec->kind = (d->a_kc ? 0 : 2) + (d->b_kc ? 0 : 1);
      rec->M = d->M; rec->N = d->N; rec->splits = splits; rec->K = 0;
      for (int i = 0; i < d->nseg; ++i) rec->K += d->seg[i].K;
      prof_desc(rec, d, st);
      (void)hipEventRecord(rec->e0, st);
    }
    SSC_TRY(x3w_prepare());
    KGroup g1;
    group_of_one(g1, k, grid);
    SSC_AUDIT_BEGIN();
    SSC_LAUNCH(x3w_skinny_fn(d->b_kc), dim3(g1.first[1]), dim3(x3w_skinny_threads()), (x3w_lds_bytes<64, 256>()), st, g1);
    SSC_AUDIT_END(g1, true, d->b_kc != 0, "64x256", st);
    if (rec) (void)hipEventRecord(rec->e1, st);
    SSC_CHECK_LAUNCH();
    return SSC_OK;
  }
  if (gemm_mode() == 1 && vec && !(!d->a_kc && d->b_kc) && (compact || k.topk || (g_x3b && big_tile(d->M, d->N)))) {
    dim3 grid(ssc_cdiv(d->N, 128), ssc_cdiv(d->M, 128), splits);
    ProfRec* rec = nullptr;
    if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {
      rec = &g_prof[g_prof_n++];
      rec->bytes = rec->flops = 0.0; rec->nmem = 0;
      rec->kind = (d->a_kc ? 0 : 2) + (d->b_kc ? 0 : 1);
      rec->M = d->M; rec->N = d->N; rec->splits = splits; rec->K = 0;
      for (int i = 0; i < d->nseg; ++i) rec->K += d->seg[i].K;
      prof_desc(rec, d, st);
      (void)hipEventRecord(rec->e0, st);
    }
    const bool kg = k.karows || k.kbrows;
    const bool kg_both = k.karows && k.kbrows;   // the wave-specialised gather kernel reads both lists unconditionally
    // Form: the 4-wave kernel keeps two workgroups per CU, which evens out small grids (380 tiles on 256 CUs); the
    // wave-specialised one runs ~13 % fewer cycles per k-step and wins once the grid is several rounds deep (decode:
    // 1520 tiles, +10 % tokens/s).  g_x3b: 1 = choose by grid size, 2 = always wave-specialised, 3 = always 4-wave.
    const long wgs = (long)grid.x * grid.y * grid.z;
    // (below 512 rows - one to four tile rows, split-K - the wave-specialised form wins at every grid size: 59 vs 71 us at 128 x 4800 x 5648)
    // (2xFP16 numerics requested and applicable: the wave-specialised form at every grid size - its F16 variant is 1.5x the 3xBF16
    // one, which outweighs what the 4-wave kernel gains on grids below three rounds)
    // (NT only.  The NN / TN / TN-with-row-lists forms were instantiated and measured in round 4 for the TRAIN step's large products -
    // weight gradients 4800 x 5448 x 1344 with k-row lists 355 -> 266 us, vocabulary-head input gradient 328 -> 267 us, error against
    // float64 6-8e-7 of sum|a||b| with measured scales, as good as 3xBF16 - but 1.23-1.33x on 20 % of a step, minus the per-step
    // absmax passes over the gradient operands, is ~3 % of the train step: not taken, the training numerics stay 3xBF16.)
    const bool f16 = (ssc_tls_gemm_f16 >= 0 ? ssc_tls_gemm_f16 : g_gemm_f16) && d->a_kc && d->b_kc && !kg && g_x3w_big_npw == 8 && g_x3b != 3;
    if (k.topk && !(d->a_kc && d->b_kc && !kg && splits == 1 && x3w_span_ok(d))) return SSC_EINVAL;   // the records exist in this form's epilogue only
    if ((k.topk || g_x3b == 2 || (g_x3b == 1 && (f16 || wgs >= 768 || (d->M < 512 && big_tile(d->M, d->N))))) && x3w_span_ok(d) && (!kg || kg_both)) {  // wave-specialised form: 12 waves, 120 KB of dynamic LDS
      group_fn fn = x3w_big_fn(d->a_kc, d->b_kc, kg);
      // 2xFP16: two planes per operand = 80 KB per workgroup; with 4 producer waves (8 waves per workgroup) TWO workgroups share a
      // CU, so one's pipeline fill / epilogue runs under the other's k-loop (g_f16_npw = 4)
      if (f16) fn = g_f16_npw == 4 ? gemm_x3w_kernel<true, true, false, 128, 128, 2, 4, true> : gemm_x3w_kernel<true, true, false, 128, 128, 2, 8, true>;
      SSC_TRY(x3w_prepare());
      KGroup g1;
      group_of_one(g1, k, grid);
      SSC_AUDIT_BEGIN();
      if (f16) SSC_LAUNCH(fn, dim3(g1.first[1]), dim3(g_f16_npw == 4 ? 512 : 768), (x3w_lds_bytes_f16<128, 128>()), st, g1);
      else SSC_LAUNCH(fn, dim3(g1.first[1]), dim3(x3w_big_threads()), (x3w_lds_bytes<128, 128>()), st, g1);
      SSC_AUDIT_END(g1, d->a_kc != 0, d->b_kc != 0, "128x128", st);
    } else
    if (d->a_kc && d->b_kc) SSC_LAUNCH((gemm_x3b_kernel<true, true, false>), grid, dim3(256), 0, st, k);
    else if (d->a_kc) SSC_LAUNCH((gemm_x3b_kernel<true, false, false>), grid, dim3(256), 0, st, k);
    else if (kg) SSC_LAUNCH((gemm_x3b_kernel<false, false, true>), grid, dim3(256), 0, st, k);
    else SSC_LAUNCH((gemm_x3b_kernel<false, false, false>), grid, dim3(256), 0, st, k);
    if (rec) (void)hipEventRecord(rec->e1, st);
    SSC_CHECK_LAUNCH();
    return SSC_OK;
  }
  if (use_x3(d, vec)) {
    const bool wide = x3_wide(d->M, d->N);
    dim3 grid(ssc_cdiv(d->N, wide ? 128 : 64), ssc_cdiv(d->M, 64), splits);
    ProfRec* rec = nullptr;
    if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {
      rec = &g_prof[g_prof_n++];
      rec->bytes = rec->flops = 0.0; rec->nmem = 0;
      rec->kind = 0; rec->M = d->M; rec->N = d->N; rec->splits = splits; rec->K = 0;
      for (int i = 0; i < d->nseg; ++i) rec->K += d->seg[i].K;
      prof_desc(rec, d, st);
      (void)hipEventRecord(rec->e0, st);
    }
    if (wide && g_x3_nbuf == 1) SSC_LAUNCH((gemm_x3_kernel<2, 2, 1>), grid, dim3(256), 0, st, k);
    else if (wide) SSC_LAUNCH((gemm_x3_kernel<2, 2, 2>), grid, dim3(256), 0, st, k);
    else if (g_x3_nbuf == 1 && g_x3_pf == 4) SSC_LAUNCH((gemm_x3_kernel<4, 1, 1>), grid, dim3(256), 0, st, k);
    else if (g_x3_nbuf == 1 && g_x3_pf == 1) SSC_LAUNCH((gemm_x3_kernel<1, 1, 1>), grid, dim3(256), 0, st, k);
    else if (g_x3_nbuf == 1) SSC_LAUNCH((gemm_x3_kernel<2, 1, 1>), grid, dim3(256), 0, st, k);
    else SSC_LAUNCH((gemm_x3_kernel<2, 1, 2>), grid, dim3(256), 0, st, k);
    if (rec) (void)hipEventRecord(rec->e1, st);
    SSC_CHECK_LAUNCH();
    return SSC_OK;
  }
  const bool big = big_tile(d->M, d->N);
  const bool wide = !big && vec && wide_tile(d->M, d->N);
  const int bm = big ? 128 : 64, bn = (big || wide) ? 128 : 64;
  dim3 grid(ssc_cdiv(d->N, bn), ssc_cdiv(d->M, bm), splits);
  gemm_fn fn = big ? (vec ? pick_layout<2, 2, 1, true>(d) : pick_layout<2, 2, 1, false>(d))
                   : wide ? pick_layout<1, 2, 2, true>(d)
                          : (vec ? pick_layout<1, 1, 4, true>(d) : pick_layout<1, 1, 2, false>(d));
  ProfRec* rec = nullptr;
  if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {
    rec = &g_prof[g_prof_n++];
      rec->bytes = rec->flops = 0.0; rec->nmem = 0;
    rec->kind = (d->a_kc ? 0 : 2) + (d->b_kc ? 0 : 1);  // 0 NT, 1 NN, 3 TN
    rec->M = d->M; rec->N = d->N; rec->splits = splits;
    rec->K = 0;
    for (int i = 0; i < d->nseg; ++i) rec->K += d->seg[i].K;
      prof_desc(rec, d, st);
    (void)hipEventRecord(rec->e0, st);
  }
  SSC_LAUNCH(fn, grid, dim3(256), 0, st, k);
  if (rec) (void)hipEventRecord(rec->e1, st);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

}  // namespace

// Split-K choice by a small cost model fitted to rocprof timings (profiles/r01_c_gemm_shapes.csv, tools/gemm_probe.py):
// time ~ (workgroups on the busiest CU) x (k-steps per workgroup + fixed per-workgroup cost) / efficiency(occupancy)
//        + per-slab cost.  One wave per SIMD leaves the MFMA pipe ~45 % busy, three or four ~85-90 %.
// C (+)= sum of `nslab` partial slabs (M x N, ld N, `stride` floats apart), fixed order
int ssc_reduce_slabs(const float* slabs, int nslab, size_t stride, int M, int N, float* C, int ldc, const float* bias,
                     int accumulate, hipStream_t st) {
  if (!slabs || nslab < 1 || M <= 0 || N <= 0 || !C) return SSC_EINVAL;
  const size_t total = (size_t)M * N;
  SSC_LAUNCH(reduce_slabs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slabs, nslab, stride, M, N, C, ldc, bias,
             accumulate, (const int*)nullptr, (const int*)nullptr);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_gemm_auto_splits(int M, int N, int ksteps) {
  if (x3w_skinny_shape(M, N)) {
    // one workgroup per CU: split K until the grid covers the chip once, at least 4 k-steps per workgroup
    const int tiles = ssc_cdiv(N, 256);
    int s = 256 / tiles;
    if (s > ksteps / 4) s = ksteps / 4;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    const int per = ssc_cdiv(ksteps, s);
    return ssc_cdiv(ksteps, per);
  }
  const bool big = big_tile(M, N);
  const bool wide = !big && wide_tile(M, N);
  const long tiles = big ? (long)ssc_cdiv(M, 128) * ssc_cdiv(N, 128)
                         : (long)ssc_cdiv(M, BM) * ssc_cdiv(N, wide ? 128 : BN);
  const int occ = (big || wide) ? 2 : 4;               // resident workgroups per CU
  static const float eff_small[5] = {0.f, 0.45f, 0.70f, 0.85f, 0.90f};
  static const float eff_big[3] = {0.f, 0.55f, 0.85f};
  const float fixed = big ? 3.f : 6.f;                 // prologue/epilogue per workgroup, in k-steps
  const float slab = big ? 2.5f : 0.4f;                // write + re-read of one partial slab, in k-steps
  int best = 1;
  float best_cost = 1e30f;
  const int smax = ksteps / (big ? 8 : 4) < 1 ? 1 : ksteps / (big ? 8 : 4);
  for (int s = 1; s <= 40 && s <= smax; ++s) {
    const int per = ssc_cdiv(ksteps, s);
    if (ssc_cdiv(ksteps, per) != s) continue;          // would leave an empty trailing split
    const long wgs = tiles * s;
    const int c = (int)((wgs + 255) / 256);            // workgroups on the busiest CU
    const int resident = c < occ ? c : occ;
    const float eff = (big || wide) ? eff_big[resident] : eff_small[resident];
    float cost = (float)c * ((float)per + fixed) / eff + (s > 1 ? slab * s : 0.f);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
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
  k.crows = nullptr;   // slabs hold compact rows; the consumer scatters
  return launch(d, k, splits, st);
}

// Skinny GEMM for a fused epilogue: leaves its split-K slabs (M x N, ld N) in `slabs`.  Segments that allow
// 16 B/lane loads and segments that do not (e.g. the z-block of W_ih^dec, which starts at an odd column when the
// sentiment column is present) go to separate launches, so one misaligned segment does not push the whole product
// onto the 4 B/lane kernel.  *nslab = total number of slabs written.
int ssc_gemm_slabs_auto(const ssc_gemm_desc* d, float* slabs, size_t cap_floats, int* nslab, hipStream_t st) {
  KArgs k;
  SSC_TRY(build_args(d, k));
  ssc_gemm_desc part[2];
  int np[2] = {0, 0};
  for (int g = 0; g < 2; ++g) { part[g] = *d; part[g].nseg = 0; }
  for (int i = 0; i < k.nseg; ++i) {
    int g = (k.seg[i].avec && k.seg[i].bvec) ? 0 : 1;
    part[g].seg[np[g]++] = d->seg[i];
  }
  if (np[0] == 0 || np[1] == 0) { np[0] = d->nseg; np[1] = 0; part[0] = *d; }
  const size_t mn = (size_t)d->M * d->N;
  int total = 0;
  for (int g = 0; g < 2; ++g) {
    if (!np[g]) continue;
    part[g].nseg = np[g];
    int ksteps = 0;
    for (int i = 0; i < np[g]; ++i) ksteps += ssc_cdiv(part[g].seg[i].K, BK);
    int splits = ssc_gemm_auto_splits(d->M, d->N, ksteps);
    // a product over a device-side row list leaves ONE compact slab (its consumers index it by slot; a decode step of 512-800
    // rows is a grid below one round of workgroups, which the split heuristic would otherwise cut along K)
    if (d->m_count || d->a_rows || d->c_rows) splits = 1;
    while (splits > 1 && (size_t)(total + splits) * mn > cap_floats) --splits;
    if ((size_t)(total + splits) * mn > cap_floats) return SSC_EWORKSPACE;
    int per = ssc_cdiv(ksteps, splits);
    splits = ssc_cdiv(ksteps, per);
    SSC_TRY(ssc_gemm_slabs(&part[g], splits, slabs + (size_t)total * mn, st));
    total += splits;
  }
  *nslab = total;
  return SSC_OK;
}

int ssc_gemm_slabs_group(const ssc_gemm_desc* const* d, int n, float* const* regions, const size_t* caps, int* nslab,
                         hipStream_t st) {
  if (!d || n < 1 || n > SSC_GROUP_MAX || !regions || !caps || !nslab) return SSC_EINVAL;
  const bool group_on = g_gemm_group != 0;
  KGroup g;
  bool ok = n >= 2 && group_on, any_wide = false;
  long work = 0;
  // two classes of members: minibatches of up to 64 rows on the 64x256 kernels, and of 65-511 rows (MFMA-bound in 3xBF16: C5's
  // B = 128 per GPU) on the wave-specialised 128x128 kernels; a group is of one class
  const bool mid = d[0] && d[0]->M > 64 && d[0]->M >= g_big_min_m;
  const int tw = mid ? 128 : 256;
  for (int i = 0; i < n && ok; ++i) {
    SSC_TRY(build_args(d[i], g.a[i]));
    KArgs& k = g.a[i];
    bool vec = true;
    for (int s = 0; s < k.nseg; ++s) vec = vec && k.seg[s].avec && k.seg[s].bvec;
    const bool compact = k.mcount || k.arows || k.crows || k.kcount || k.karows || k.kbrows;
    // a member needs the minibatch shape and 16 B/lane operands; a NARROW member (N < the 256-column kernels' minimum width,
    // e.g. the 128-column dz product) may ride along in a group that has at least one wide member
    if (mid) {
      ok = !compact && vec && d[i]->a_kc && gemm_mode() == 1 && g_x3b != 0 && g_x3b != 3 && d[i]->M >= g_big_min_m && d[i]->M < 512 &&
           d[i]->M == d[0]->M && d[i]->N >= 64 && x3w_span_ok(d[i]) && d[i]->b_kc == d[0]->b_kc && regions[i];
      any_wide = any_wide || big_tile(d[i]->M, d[i]->N);
    } else {
      ok = !compact && x3w_group_member(d[i], vec) && d[i]->b_kc == d[0]->b_kc && regions[i];
      any_wide = any_wide || x3w_skinny(d[i], vec);
    }
    work += (long)ssc_cdiv(d[i]->N, tw) * ssc_cdiv(d[i]->M, mid ? 128 : 64) * k.steps_total;
  }
  ok = ok && any_wide;
  if (!ok) {
    for (int i = 0; i < n; ++i) SSC_TRY(ssc_gemm_slabs_auto(d[i], regions[i], caps[i], &nslab[i], st));
    return SSC_OK;
  }
  // one workgroup per CU in total: every product gets splits in proportion to its k-steps, at least 4 k-steps per workgroup
  int per = (int)((work + 255) / 256);
  if (per < 4) per = 4;
  for (;; ++per) {  // the whole group in one round of 256 workgroups (a few stragglers in a second round double the time)
    long wgs = 0;
    for (int i = 0; i < n; ++i) wgs += (long)ssc_cdiv(d[i]->N, tw) * ssc_cdiv(d[i]->M, mid ? 128 : 64) * ssc_cdiv(g.a[i].steps_total, per);
    if (wgs <= 256 || per >= 4096) break;
  }
  g.n = n;
  g.first[0] = 0;
  int Ksum = 0, Nmax = 0;
  for (int i = 0; i < n; ++i) {
    KArgs& k = g.a[i];
    k.member = i;
    const size_t mn = (size_t)d[i]->M * d[i]->N;
    int splits = ssc_cdiv(k.steps_total, per);
    const int cap = (int)(caps[i] / mn);
    if (cap < 1) return SSC_EWORKSPACE;
    if (splits > cap) splits = cap;
    const int p2 = ssc_cdiv(k.steps_total, splits);
    splits = ssc_cdiv(k.steps_total, p2);
    k.steps_per_split = p2;
    k.out = regions[i];
    k.ldo = d[i]->N;
    k.slab_stride = mn;
    k.bias = nullptr;
    k.accumulate = 0;
    k.crows = nullptr;
    g.gx[i] = ssc_cdiv(d[i]->N, tw);
    g.gy[i] = mid ? ssc_cdiv(d[i]->M, 128) : 1;
    g.gz[i] = splits;
    g.first[i + 1] = g.first[i] + g.gx[i] * g.gy[i] * splits;
    nslab[i] = splits;
    for (int s = 0; s < d[i]->nseg; ++s) Ksum += d[i]->seg[s].K;
    if (d[i]->N > Nmax) Nmax = d[i]->N;
  }
  for (int i = n; i < SSC_GROUP_MAX; ++i) { g.first[i + 1] = g.first[n]; g.gx[i] = 1; g.gy[i] = 1; g.gz[i] = 1; }
  SSC_TRY(x3w_prepare());
  ProfRec* rec = nullptr;
  if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {  // one record for the group
    rec = &g_prof[g_prof_n++];
      rec->bytes = rec->flops = 0.0; rec->nmem = 0;
    rec->kind = d[0]->b_kc ? 0 : 1;
    rec->M = d[0]->M; rec->N = Nmax; rec->splits = nslab[0]; rec->K = Ksum;   // (N, K: nominal; bytes / flops are exact)
    for (int i = 0; i < n; ++i) prof_desc(rec, d[i], st);
    (void)hipEventRecord(rec->e0, st);
  }
  if (mid) {
    SSC_AUDIT_BEGIN();
    SSC_LAUNCH(x3w_big_fn(true, d[0]->b_kc != 0, false), dim3(g.first[n]), dim3(x3w_big_threads()), (x3w_lds_bytes<128, 128>()), st, g);
    SSC_AUDIT_END(g, true, d[0]->b_kc != 0, "128x128 group", st);
    if (rec) (void)hipEventRecord(rec->e1, st);
    SSC_CHECK_LAUNCH();
    return SSC_OK;
  }
  SSC_AUDIT_BEGIN();
  SSC_LAUNCH(x3w_skinny_fn(d[0]->b_kc), dim3(g.first[n]), dim3(x3w_skinny_threads()), (x3w_lds_bytes<64, 256>()), st, g);
  SSC_AUDIT_END(g, true, d[0]->b_kc != 0, "64x256 group", st);
  if (rec) (void)hipEventRecord(rec->e1, st);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

// n independent LARGE products C_i = A_i^T B_i (the weight gradients of one backward phase) with direct outputs: the
// eligible ones (3xBF16 mode, 16-B operands, m/n-contiguous single segment, all with or all without k-row gather lists)
// go out as grouped launches of the wave-specialised 128x128 kernel - several rounds of workgroups per launch, so its one
// workgroup per CU no longer loses to tile quantisation (380 tiles on 256 CUs) - the others one by one.
int ssc_gemm_dw_group(const ssc_gemm_desc* const* d, int n, hipStream_t st) {
  if (!d || n < 1) return SSC_EINVAL;
  const bool group_on = g_gemm_group != 0 && g_dw_group != 0;
  int i = 0;
  while (i < n) {
    KGroup g;
    int m = 0;
    bool kg0 = false;
    long Ksum = 0, MN = 0;
    int j = i;
    for (; j < n && m < SSC_GROUP_MAX; ++j) {
      const ssc_gemm_desc* dj = d[j];
      KArgs& k = g.a[m];
      SSC_TRY(build_args(dj, k));
      k.member = m;
      const bool vec = k.nseg == 1 && k.seg[0].avec && k.seg[0].bvec;
      const bool kg = k.karows || k.kbrows;
      const bool ok = group_on && gemm_mode() == 1 && vec && !dj->a_kc && !dj->b_kc && dj->C && dj->ldc >= dj->N && !k.mcount &&
                      !k.arows && !k.crows && (kg ? (k.kcount && k.karows && k.kbrows) : !k.kcount) && (m == 0 || kg == kg0) &&
                      x3w_span_ok(dj);
      if (!ok) break;
      kg0 = kg;
      k.steps_per_split = k.steps_total;
      k.out = dj->C; k.ldo = dj->ldc; k.slab_stride = 0; k.bias = dj->bias; k.accumulate = dj->accumulate;
      g.gx[m] = ssc_cdiv(dj->N, 128); g.gy[m] = ssc_cdiv(dj->M, 128); g.gz[m] = 1;
      if (m == 0) g.first[0] = 0;
      g.first[m + 1] = g.first[m] + g.gx[m] * g.gy[m];
      Ksum = dj->seg[0].K; MN += (long)dj->M * dj->N;
      ++m;
    }
    if (m >= 2) {
      g.n = m;
      g.first[0] = 0;
      for (int q = m; q < SSC_GROUP_MAX; ++q) { g.first[q + 1] = g.first[m]; g.gx[q] = g.gy[q] = g.gz[q] = 1; }
      SSC_TRY(x3w_prepare());
      ProfRec* rec = nullptr;
      if (g_prof_on && g_prof && g_prof_n < PROF_MAX) {  // one record: 2*K*sum(M_i N_i) flops
        rec = &g_prof[g_prof_n++];
      rec->bytes = rec->flops = 0.0; rec->nmem = 0;
        rec->kind = 3;
        rec->M = (int)(MN / d[i]->N); rec->N = d[i]->N; rec->splits = 1; rec->K = (int)Ksum;
        for (int q = i; q < j; ++q) prof_desc(rec, d[q], st);
        (void)hipEventRecord(rec->e0, st);
      }
      SSC_AUDIT_BEGIN();
      SSC_LAUNCH(x3w_big_fn(false, false, kg0), dim3(g.first[m]), dim3(x3w_big_threads()), (x3w_lds_bytes<128, 128>()), st, g);
      SSC_AUDIT_END(g, false, false, "128x128 TN group", st);
      if (rec) (void)hipEventRecord(rec->e1, st);
      SSC_CHECK_LAUNCH();
      i += m;
    } else {  // a single eligible product gains nothing from the group form; ineligible ones take the usual path
      SSC_TRY(ssc_gemm(d[i], (void*)st));
      ++i;
    }
  }
  return SSC_OK;
}

// ssc_split_f16: an operand's two fp16 pieces in the plane layout (KSeg::A16), with exactly the arithmetic of the 2xFP16 producers
// (x * scale, split4_f16): one thread per 4 consecutive k of a row
namespace {
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int rows, int K, int ldx, const float* __restrict__ scale,
                                                        unsigned* __restrict__ out, int ldo, const int* __restrict__ row_list,
                                                        const int* __restrict__ row_count) {
  const int kq = blockIdx.x * blockDim.x + threadIdx.x;   // chunk of 4 k
  const int kp4 = (K + BK - 1) / BK * (BK / 4);
  if (kq >= kp4) return;
  const float sc = scale ? *scale : 1.f;
  const int n = row_list ? min(rows, *row_count) : rows;
  const bool vec = !(ldx & 3) && ssc_aligned16_dev(x);
  const int k = 4 * kq;
  for (int i = blockIdx.y; i < n; i += gridDim.y) {
    const int r = row_list ? row_list[i] : i;
    const float* xr = x + (size_t)r * ldx;
    f32x4 v;
    if (k + 4 <= K && vec) v = *reinterpret_cast<const f32x4*>(xr + k);
    else
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = k + j < K ? xr[k + j] : 0.f;
    v = v * sc;
    u32x2 hi, lo;
    split4_f16(v, hi, lo);
    unsigned* o = out + (size_t)r * ldo + (kq >> 3) * 32 + (kq & 7) * 2;
    *reinterpret_cast<u32x2*>(o) = hi;
    *reinterpret_cast<u32x2*>(o + 16) = lo;
  }
}
}  // namespace
extern "C" int ssc_split_f16(const float* x, int rows, int K, int ldx, const float* scale, void* out, int ldo, const int* row_list,
                             const int* row_count, void* stream) {
  const int kp = ssc_cdiv(K, BK) * BK;
  if (!x || !out || rows <= 0 || K <= 0 || ldx < K || ldo < kp || (ldo & 3) || !ssc_aligned16(out) || (row_list != nullptr) != (row_count != nullptr))
    return SSC_EINVAL;
  SSC_LAUNCH(split_f16_kernel, dim3(ssc_cdiv(kp / 4, 256), rows < 65535 ? rows : 65535), dim3(256), 0, (hipStream_t)stream, x, rows, K, ldx, scale,
             static_cast<unsigned*>(out), ldo, row_list, row_count);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_gemm(const ssc_gemm_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  KArgs k;
  SSC_TRY(build_args(d, k));
  if (d->topk_part) {   // records instead of C: one pass, wave-specialised 128x128 NT form (launch() refuses anything else)
    k.out = nullptr; k.ldo = d->N; k.slab_stride = 0; k.bias = d->bias; k.accumulate = 0;
    const bool vec_all = [&] { for (int i = 0; i < k.nseg; ++i) if (!k.seg[i].avec || !k.seg[i].bvec) return false; return true; }();
    if (!vec_all || gemm_mode() != 1 || !d->a_kc || !d->b_kc) return SSC_EINVAL;
    return launch(d, k, 1, st);
  }
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
    // auto mode with a small workspace: the largest split count that fits (1 = single pass)
    splits = d->workspace ? (int)(d->workspace_floats / ((size_t)d->M * d->N)) : 1;
    if (splits < 1) splits = 1;
    int per = ssc_cdiv(k.steps_total, splits);
    splits = ssc_cdiv(k.steps_total, per);
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
  k.crows = nullptr;   // slabs hold compact rows; reduce_slabs_kernel scatters
  SSC_TRY(launch(d, k, splits, st));
  size_t total = (size_t)d->M * d->N;
  SSC_LAUNCH(reduce_slabs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d->workspace, splits,
                     k.slab_stride, d->M, d->N, d->C, d->ldc, d->bias, d->accumulate, d->m_count, d->c_rows);
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
  if (on && !g_prof_counts && hipHostMalloc((void**)&g_prof_counts, PROF_SLOTS * sizeof(int), hipHostMallocDefault) != hipSuccess) return SSC_EHIP;
  g_prof_on = on != 0;
  if (on) { g_prof_n = 0; g_prof_slots = 0; }
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
    float* o = out + (size_t)i * 8;
    g_prof[i].bytes = g_prof[i].flops = 0.0;
    for (int q = 0; q < g_prof[i].nmem; ++q) {   // true extents: rows / k-rows past the device-side count are never touched
      const ProfMember& m = g_prof[i].mem[q];
      double M = m.M, K = m.K;
      if (m.slot_m >= 0 && g_prof_counts[m.slot_m] >= 0 && g_prof_counts[m.slot_m] < m.M) M = g_prof_counts[m.slot_m];
      if (m.slot_k >= 0 && g_prof_counts[m.slot_k] >= 0 && g_prof_counts[m.slot_k] < m.K) K = g_prof_counts[m.slot_k];
      g_prof[i].bytes += 4.0 * (K * m.N + M * K + M * m.N);
      g_prof[i].flops += 2.0 * M * m.N * K;
    }
    o[0] = (float)g_prof[i].kind; o[1] = (float)g_prof[i].M; o[2] = (float)g_prof[i].N; o[3] = (float)g_prof[i].K;
    o[4] = (float)g_prof[i].splits; o[5] = ms; o[6] = (float)g_prof[i].bytes; o[7] = (float)g_prof[i].flops;
  }
  g_prof_n = 0;
  g_prof_slots = 0;
  return n;
}

// Process-wide numerics switch for NT products: 1 = 3xBF16 split on the bf16 matrix cores (default, fp32-accurate),
// 0 = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  Returns the previous mode.  Default from SSC_GEMM_MODE=f32|x3.
extern "C" int ssc_set_gemm_mode(int mode) {
  int prev = gemm_mode();
  if (mode == 0 || mode == 1) g_gemm_mode = mode;
  return prev;
}

// ---- include/ssc_debug.h -----------------------------------------------------------------------------------------
extern int ssc_g_dec_att_table, ssc_g_dec_dedup, ssc_g_beam_reg, ssc_g_dec_ungathered, ssc_g_dec_parts, ssc_g_dec_planes;   // decode.hip
extern int ssc_g_img_mfma;   // pointwise.hip
namespace {
struct DebugKey { const char* name; int* var; };
const DebugKey g_debug_keys[] = {
    {"large_form", &g_x3b},          // large products (M, N >= 512): 0 = 64-wide kernels, 3 = 4-wave 128x128 3xBF16 kernel (default), 2 = its wave-specialised form, 1 = chosen by grid size   (SSC_X3B)
    {"x3_wide", &g_x3_wide},         // 64x128 block tile for 3xBF16 products with M <= 64, N >= 1024 on the 4-wave kernel
    {"x3_nbuf", &g_x3_nbuf},         // LDS stages of the 64-wide 3xBF16 kernel (1 | 2)
    {"x3_pf", &g_x3_pf},             // register prefetch depth of the 64-wide 3xBF16 kernel (1 | 2 | 4)
    {"x3w_skinny", &g_x3w_skinny},   // minibatch products on the wave-specialised 64x256 kernel: 0 off, 1 NT and NN (default), 2 NN only   (SSC_X3W_SKINNY)
    {"x3w_min_n", &g_x3w_min_n},     // ... from this output width on   (SSC_X3W_MIN_N)
    {"wide_min_n", &g_wide_min_n},   // exact-fp32 kernels: 64x128 tile for M <= 64 from this width on
    {"gemm_group", &g_gemm_group},   // grouped launches of independent minibatch products (0 | 1)   (SSC_GEMM_GROUP)
    {"dw_group", &g_dw_group},       // grouped launches of the weight-gradient products (0 | 1)   (SSC_DW_GROUP)
    {"x3w_big_npw", &g_x3w_big_npw}, // wave-specialised 128x128 kernels: producer waves (4 | 8)   (SSC_X3W_BIG_NPW)
    {"x3w_npw", &g_x3w_npw},         // 64x256 kernels: producer waves per workgroup (4 | 8)   (SSC_X3W_NPW)
    {"x3w_pf", &g_x3w_pf},           // 64x256 kernels: k-steps in flight in the producers' registers (2 | 3)   (SSC_X3W_PF)
    {"store_wt", &g_store_wt},       // wave-specialised kernels: write-through (sc1) output stores (0 | 1)   (SSC_STORE_WT)
    {"tile_gm", &g_tile_gm},         // tile rows per group of the tile order (8; 0 = row-major)   (SSC_TILE_GM)
    {"dec_dedup", &ssc_g_dec_dedup},           // decode: parent-state products on the distinct parents of a beam group (1 | 0)   (SSC_DEC_DEDUP)
    {"beam_reg", &ssc_g_beam_reg},             // decode: beam selection with the vocabulary row in registers (1 | 0)              (SSC_BEAM_REG)
    {"img_mfma", &ssc_g_img_mfma},             // decode: the image cell's table contraction on the fp32 matrix cores (1 | 0 = VALU form)   (SSC_IMG_MFMA)
    {"dec_ungathered", &ssc_g_dec_ungathered}, // decode: states left in the previous step's row order, read through the parent lists (1 | 0)   (SSC_DEC_UNGATHERED)
    {"dec_planes", &ssc_g_dec_planes},         // decode: the 2xFP16 products of a large call read states and weights pre-split into fp16 pieces (1 | 0)   (SSC_DEC_PLANES)
    {"dec_parts", &ssc_g_dec_parts},           // decode: the vocabulary head of a one-state search leaves per-tile records instead of logits (1 | 0)   (SSC_DEC_PARTS)
    {"dec_att_table", &ssc_g_dec_att_table},   // decode: attended-feature term of the decoder gates from a per-image table (1 | 0)   (SSC_DEC_ATT_TABLE)
    {"f16_npw", &g_f16_npw},         // 2xFP16 kernel: producer waves (8 | 4 = two workgroups per CU)   (SSC_F16_NPW)
    {"gemm_f16", &g_gemm_f16},       // op-level products (ssc_gemm outside a sequence-level call): 1 = the wave-specialised 128x128 NT form takes the 2xFP16 split (what ssc_model_cfg.gemm_mode 3 selects per call)
    {"big_min_m", &g_big_min_m},     // rows from which a product with N >= 512 takes 128x128 tiles (65; 512 = the behaviour until late in round 2)   (SSC_BIG_MIN_M)
};
}  // namespace

extern "C" int ssc_debug_set(const char* key, int value) {
  if (!key) return SSC_EINVAL;
  for (const DebugKey& k : g_debug_keys)
    if (!strcmp(k.name, key)) { *k.var = value; return SSC_OK; }
  return SSC_EINVAL;
}
extern "C" int ssc_debug_get(const char* key, int* value) {
  if (!key || !value) return SSC_EINVAL;
  for (const DebugKey& k : g_debug_keys)
    if (!strcmp(k.name, key)) { *value = *k.var; return SSC_OK; }
  return SSC_EINVAL;
}

// diagnostic: resident workgroups per CU the runtime reports for the GEMM kernels (tools/, not used by the product path)
#ifdef SSC_X3W_STAMP
// diagnostics build only: device buffer of 256 u64 and the workgroup (flat blockIdx.x of the grouped launch) that fills it
extern "C" int ssc_debug_stamp_setup(void* buf, int wg) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_ptr), &buf, sizeof(buf)) != hipSuccess) return SSC_EHIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_wg), &wg, sizeof(wg)) != hipSuccess) return SSC_EHIP;
  return SSC_OK;
}
#endif

extern "C" int ssc_debug_gemm_occupancy(int* out4) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_x3_kernel<2, 1, 2>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[0] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_x3_kernel<2, 1, 1>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[1] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_kernel<true, true, 1, 1, 4, true>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[2] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_kernel<true, false, 1, 2, 2, true>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[3] = n;
  return SSC_OK;
}

extern "C" int ssc_debug_gemm_occupancy_x3b(int* out4) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_x3b_kernel<true, true, false>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[0] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_x3b_kernel<true, false, false>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[1] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_x3b_kernel<false, false, false>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[2] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_x3b_kernel<false, false, true>, 256, 0) != hipSuccess) return SSC_EHIP;
  out4[3] = n;
  return SSC_OK;
}

// Compiled finite-state machines for constrained beam search on gfx950.
//
// Reference: ConstrainedBeamSearch.search, the per-target-state loop (updown-baseline/updown/modules/cbs.py:157-250), over the
// dense adjacency tensors FiniteStateMachineBuilder produces (updown-baseline/updown/utils/constraints.py:328-478).
//
// The dense form costs one scan of a row's V log-probs and V mask bytes PER TARGET STATE (decode.hip: beam_row_topk*_kernel on a
// (rows, S) grid).  The reference's machines send nearly every token of a from-state to one target set - the self-loop of a main
// state, the reset state of a sub-state (constraints.py:448-476) - and only the word forms of the constraint words go elsewhere.
// fsm_compile_kernel finds, per (machine, from-state), that default target set and the list of exception tokens;
// beam_row_fsm_kernel then reads a row ONCE: the top per_node NON-exception tokens by one k-pass selection, the exception tokens'
// log-probs by a gather, and per target state a selection among <= E + per_node candidates.  Bit-identical to the dense kernels:
// for target i the masked row is  x[w] if bit i of T(s,w) else -1e20;  with N = the non-exception tokens, T = D on all of N, so
//   i in D:      top-k(row) = top-k( exceptions  U  top-k of N by (x desc, w asc) )
//   i not in D:  every token of N is worth -1e20: top-k(row) = top-k( exceptions  U  the k smallest tokens of N )
// under the one total order (value descending, token ascending) every selection of this library uses.
#include "beam_common.h"
#include "ssc_common.h"

namespace {

struct FsmT {
  const int* ok;           // (M*S) 1: default + exceptions describes this from-state
  const uint32_t* dflt;    // (M*S) default target set
  const uint32_t* reach;   // (M*S) default target set | every exception's target set: the states this from-state can reach at all
  const int* nexc;         // (M*S)
  const int* etok;         // (M*S, E) ascending
  const uint32_t* emask;   // (M*S, E)
  const int* fill;         // (M*S, P) smallest non-exception tokens (-1 past the end)
  const uint32_t* bits;    // (M*S, 256, NW): bit (u & 31) of word (t, u >> 5) <-> token t + 256 u is an exception (the ownership of
                           // the row kernels: thread t holds tokens t, t + 256, ...)
  int E, P, NW;
};
struct FsmLayout {
  size_t ok, dflt, reach, nexc, etok, emask, fill, bits, total;   // in 4-byte words
  int NW;
};
inline size_t r64w(size_t x) { return (x + 63) & ~(size_t)63; }
FsmLayout fsm_layout(const ssc_fsm_dims& d) {
  FsmLayout l;
  const size_t ms = (size_t)d.M * d.S;
  l.NW = (ssc_cdiv(d.V, 256) + 31) / 32;
  size_t o = 0;
  l.ok = o; o += r64w(ms);
  l.dflt = o; o += r64w(ms);
  l.reach = o; o += r64w(ms);
  l.nexc = o; o += r64w(ms);
  l.etok = o; o += r64w(ms * d.E);
  l.emask = o; o += r64w(ms * d.E);
  l.fill = o; o += r64w(ms * d.P);
  l.bits = o; o += r64w(ms * 256 * l.NW);
  l.total = o;
  return l;
}
bool dims_ok(const ssc_fsm_dims& d) {
  return d.M > 0 && d.S > 0 && d.S <= 32 && d.V > 0 && d.E >= 0 && d.E <= 4096 && d.P > 0 && d.P <= 64;
}
FsmT fsm_view(const void* tables, const ssc_fsm_dims& d) {
  const FsmLayout l = fsm_layout(d);
  const int* w = (const int*)tables;
  FsmT t;
  t.ok = w + l.ok; t.dflt = (const uint32_t*)(w + l.dflt); t.reach = (const uint32_t*)(w + l.reach); t.nexc = w + l.nexc; t.etok = w + l.etok;
  t.emask = (const uint32_t*)(w + l.emask); t.fill = w + l.fill; t.bits = (const uint32_t*)(w + l.bits);
  t.E = d.E; t.P = d.P; t.NW = l.NW;
  return t;
}

__device__ __forceinline__ int block_sum_int(int v, int* sh) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  int r = 0;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// one workgroup per (machine, from-state): fsm[m, s, :, :] (S target rows of V bytes) -> the compiled form
__global__ __launch_bounds__(256) void fsm_compile_kernel(const uint8_t* __restrict__ fsm, int S, int V, int E, int P, int NW,
                                                          int* __restrict__ ok, uint32_t* __restrict__ dflt,
                                                          uint32_t* __restrict__ reach, int* __restrict__ nexc,
                                                          int* __restrict__ etok,
                                                          uint32_t* __restrict__ emask, int* __restrict__ fill,
                                                          uint32_t* __restrict__ bits) {
  __shared__ int sh[4];
  __shared__ int scan[256];
  __shared__ uint32_t cand[3];
  const int ms = blockIdx.x, t = threadIdx.x;
  const uint8_t* base = fsm + (size_t)ms * S * V;
  auto T = [&](int w) -> uint32_t {   // target set of token w
    uint32_t r = 0;
    for (int i = 0; i < S; ++i) r |= base[(size_t)i * V + w] ? (1u << i) : 0u;
    return r;
  };
  uint32_t* bw = bits + (size_t)ms * 256 * NW;
  for (int j = t; j < 256 * NW; j += 256) bw[j] = 0u;
  // the default target set is the most common one; three probes (late tokens first: ids 0 / 1 are @@UNKNOWN@@ - where every
  // out-of-vocabulary word form lands - and @@BOUNDARY@@) and the first whose complement fits the exception capacity wins
  if (t == 0) { cand[0] = T(V - 1); cand[1] = T(V / 2); cand[2] = T(min(2, V - 1)); }
  __syncthreads();
  int chosen = -1, total = 0;
  for (int k = 0; k < 3 && chosen < 0; ++k) {
    if (k > 0 && (cand[k] == cand[k - 1] || cand[k] == cand[0])) continue;
    const uint32_t c = cand[k];
    int cnt = 0;
    for (int w = t; w < V; w += 256) cnt += T(w) != c;
    total = block_sum_int(cnt, sh);
    if (total <= E) chosen = k;
  }
  if (chosen < 0) {
    if (t == 0) { ok[ms] = 0; dflt[ms] = 0u; reach[ms] = 0xffffffffu; nexc[ms] = 0; }   // (dense scans: no shortcut)
    for (int j = t; j < P; j += 256) fill[(size_t)ms * P + j] = -1;
    return;
  }
  const uint32_t c = cand[chosen];
  // ordered compaction: thread t owns the contiguous token range [lo, hi)
  const int chunk = (V + 255) / 256, lo = min(t * chunk, V), hi = min(lo + chunk, V);
  int cnt = 0;
  for (int w = lo; w < hi; ++w) cnt += T(w) != c;
  scan[t] = cnt;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = t >= off ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  int pos = scan[t] - cnt;
  __shared__ uint32_t s_reach;
  if (t == 0) s_reach = c;
  __syncthreads();
  uint32_t my_reach = 0u;
  for (int w = lo; w < hi; ++w) {
    const uint32_t tw = T(w);
    if (tw != c) {
      my_reach |= tw;
      etok[(size_t)ms * E + pos] = w;
      emask[(size_t)ms * E + pos] = tw;
      const int u = w >> 8;
      atomicOr(&bw[(size_t)(w & 255) * NW + (u >> 5)], 1u << (u & 31));
      ++pos;
    }
  }
  if (my_reach) atomicOr(&s_reach, my_reach);
  __syncthreads();
  if (t == 0) {
    ok[ms] = 1; dflt[ms] = c; reach[ms] = s_reach; nexc[ms] = total;
    int e = 0, f = 0;
    for (int w = 0; w < V && f < P; ++w) {
      if (e < total && etok[(size_t)ms * E + e] == w) ++e;
      else fill[(size_t)ms * P + f++] = w;
    }
    for (; f < P; ++f) fill[(size_t)ms * P + f] = -1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// later steps, part A, one workgroup per source row g = (b, s, k): the masked top-`per_node` for EVERY target state
// (cbs.py:177-209).  REG: V <= 256 * ROW_NV, the row lives in registers (decode.hip: beam_row_topk_reg_kernel - same ownership,
// same log-sum-exp arithmetic, so x is the same bits).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int ROW_NV = 40;

template <bool NORM, bool REG>
__global__ __launch_bounds__(256) void beam_row_fsm_kernel(const float* __restrict__ lp, int ldlp, FsmT T,
                                                           const uint8_t* __restrict__ fsm, const int* __restrict__ mach,
                                                           const int64_t* __restrict__ last_pred,
                                                           const float* __restrict__ last_lp, int S, int V, int beam,
                                                           int per_node, int end_index, int skip_dead,
                                                           float* __restrict__ sval, int64_t* __restrict__ sidx) {
  extern __shared__ unsigned char smem[];
  __shared__ Cand sh2[2][4];     // (256 threads = 4 waves; two slots each: the one-barrier reductions of beam_common.h)
  __shared__ float shr2[2][4];
  int par = 0;
  const int g = blockIdx.x, t = threadIdx.x;
  const int b = g / (S * beam), s = (g / beam) % S, k = g % beam;
  const int m = mach ? mach[b] : b;
  const size_t ms = (size_t)m * S + s;
  const bool ended = last_pred[g] == end_index;                          // workgroup-uniform; an ended beam never looks at its row
  const bool junk = skip_dead && !ended && last_lp[g] <= -1e19f;         // no finite beam here: scored as all-zero log-probs
  const float* row = lp + (size_t)g * ldlp;
  const int U = (V + 255) >> 8;
  float x[REG ? ROW_NV : 1];
  float lse = 0.f;
  // ---- the row ----------------------------------------------------------------------------------------------------------
  if (REG) {
    if (!ended && !junk) {
#pragma unroll
      for (int u = 0; u < ROW_NV; ++u) x[u] = row[min(t + u * 256, V - 1)];
      if (NORM) {
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < ROW_NV; ++u)
          if (t + u * 256 < V) mx = fmaxf(mx, x[u]);
        mx = dec_block_reduce1(mx, shr2, par, true); par ^= 1;
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < ROW_NV; ++u)
          if (t + u * 256 < V) sum += expf(x[u] - mx);
        sum = dec_block_reduce1(sum, shr2, par, false); par ^= 1;
        lse = mx + logf(sum);
#pragma unroll
        for (int u = 0; u < ROW_NV; ++u) x[u] -= lse;
      }
    } else {
#pragma unroll
      for (int u = 0; u < ROW_NV; ++u) x[u] = junk ? 0.f : (t + u * 256 == end_index ? 0.f : -INFINITY);
    }
  } else if (NORM && !ended && !junk) {   // thread t owns v = t, t + 256, ...: the order of log_softmax_kernel
    float mx = -INFINITY;
    for (int v = t; v < V; v += 256) mx = fmaxf(mx, row[v]);
    mx = dec_block_reduce1(mx, shr2, par, true); par ^= 1;
    float sum = 0.f;
    for (int v = t; v < V; v += 256) sum += expf(row[v] - mx);
    sum = dec_block_reduce1(sum, shr2, par, false); par ^= 1;
    lse = mx + logf(sum);
  }
  auto xval = [&](int v) -> float {   // the cleaned log-prob of token v (cbs.py:177-186), any token
    if (ended) return v == end_index ? 0.f : -INFINITY;
    if (junk) return 0.f;
    return NORM ? row[v] - lse : row[v];
  };

  if (!T.ok[ms]) {
    // ---- dense form for a from-state that is not "default + <= E exceptions": S masked scans of the row held here --------
    for (int i = 0; i < S; ++i) {
      const uint8_t* mk = fsm + (ms * S + i) * V;
      const size_t base = ((((size_t)b * S + i) * S + s) * beam + k) * per_node;   // scratch layout (b, i, s, k, n)
      Cand prev{INFINITY, -1};
      for (int n = 0; n < per_node; ++n) {
        Cand best{-INFINITY, -1};
        // (the rare path: row and mask are re-read from memory - L2 - instead of widening the register form)
        for (int v = t; v < V; v += 256) {
          const float y = mk[v] ? xval(v) : -1e20f;
          if (after(y, v, prev) && (best.i < 0 || better(y, v, best))) best = Cand{y, v};
        }
        best = block_best1(best, sh2, par); par ^= 1;
        if (t == 0) { sval[base + n] = best.v; sidx[base + n] = best.i; }
        prev = best;
      }
    }
    return;
  }

  // ---- compiled form ------------------------------------------------------------------------------------------------------
  const int E = T.E, P = T.P, NW = T.NW;
  float* xe = reinterpret_cast<float*>(smem);                 // (E) cleaned log-probs of the exception tokens
  int* etok_s = reinterpret_cast<int*>(xe + E);               // (E)
  uint32_t* emask_s = reinterpret_cast<uint32_t*>(etok_s + E);   // (E)
  float* ntv = reinterpret_cast<float*>(emask_s + E);         // (per_node) best non-exception tokens: value,
  int* nti = reinterpret_cast<int*>(ntv + per_node);          //            token
  const int nx = T.nexc[ms];
  const int* fill = T.fill + ms * P;
  // exception tokens: their values come from a gather (the row has just been read: L2)
  for (int e = t; e < nx; e += 256) {
    const int tok = T.etok[ms * E + e];
    etok_s[e] = tok;
    emask_s[e] = T.emask[ms * E + e];
    xe[e] = xval(tok);
  }
  const uint32_t reach = T.reach[ms];
  if (junk) {   // all-zero row: the best non-exception tokens are the smallest ones
    if (t < per_node) { ntv[t] = 0.f; nti[t] = fill[t]; }
  } else {
    const uint32_t* bw = T.bits + (ms * 256 + t) * NW;
    uint32_t w0 = 0u, w1 = 0u;
    if (REG) { w0 = bw[0]; w1 = NW > 1 ? bw[1] : 0u; }
    Cand prev{INFINITY, -1};
    for (int n = 0; n < per_node; ++n) {
      Cand best{-INFINITY, -1};
      if (REG) {
#pragma unroll
        for (int u = 0; u < ROW_NV; ++u) {
          const int v = t + u * 256;
          const bool exc = u < 32 ? ((w0 >> u) & 1u) : ((w1 >> (u - 32)) & 1u);
          const float y = x[u];
          if (v < V && !exc && after(y, v, prev) && (best.i < 0 || better(y, v, best))) best = Cand{y, v};
        }
      } else {
        for (int u = 0; u < U; ++u) {
          const int v = t + u * 256;
          if (v >= V || ((bw[u >> 5] >> (u & 31)) & 1u)) continue;
          const float y = xval(v);
          if (after(y, v, prev) && (best.i < 0 || better(y, v, best))) best = Cand{y, v};
        }
      }
      best = block_best1(best, sh2, par); par ^= 1;
      if (t == 0) { ntv[n] = best.v; nti[n] = best.i; }
      prev = best;
    }
  }
  __syncthreads();
  // per target state (a wave each): top-per_node of  exceptions U (i in D ? ntop : fill at -1e20)
  const uint32_t D = T.dflt[ms];
  const int lane = t & 63, wave = t >> 6;
  const int ncand = nx + per_node;
  // A state this from-state cannot reach at all (no token leads there: most of the S targets of a constraint machine) sees an
  // all-forbidden row, every token worth -1e20: its answer is the first per_node tokens - no selection
  for (int i = wave; i < S; i += 4) {
    const bool in_d = (D >> i) & 1u;
    const size_t base = ((((size_t)b * S + i) * S + s) * beam + k) * per_node;   // scratch layout (b, i, s, k, n)
    if (!((reach >> i) & 1u)) {
      if (lane < per_node) { sval[base + lane] = -1e20f; sidx[base + lane] = lane; }
      continue;
    }
    Cand prev{INFINITY, -1};
    for (int n = 0; n < per_node; ++n) {
      Cand best{-INFINITY, -1};
      for (int c = lane; c < ncand; c += 64) {
        float y;
        int v;
        if (c < nx) {
          v = etok_s[c];
          y = ((emask_s[c] >> i) & 1u) ? xe[c] : -1e20f;
        } else if (in_d) {
          v = nti[c - nx];
          y = ntv[c - nx];
        } else {
          v = fill[c - nx];
          y = -1e20f;
        }
        if (v >= 0 && after(y, v, prev) && (best.i < 0 || better(y, v, best))) best = Cand{y, v};
      }
      best = wave_best(best);
      if (lane == 0) { sval[base + n] = best.v; sidx[base + n] = best.i; }
      prev = best;
    }
  }
}

// later steps, part A for the trivial machine when the vocabulary head left RECORDS instead of logits (ssc_gemm_desc.topk_part: per
// row and 128-column tile the maximum, sum exp(x - maximum) and the two best columns): one wave per row combines the tiles' partials
// into the row's log-sum-exp and picks the per_node (<= 2) best columns - the top-k of a row is the top-k of its tiles' top-k.
// Same order as everywhere: value descending, token ascending.  (cbs.py:177-209 with an all-ones mask.)
__global__ __launch_bounds__(256) void beam_rows_parts_kernel(const float* __restrict__ parts, int ntn,
                                                              const int64_t* __restrict__ last_pred, int G, int per_node,
                                                              int end_index, float* __restrict__ sval, int64_t* __restrict__ sidx) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= G) return;
  const size_t base = (size_t)g * per_node;   // scratch layout (b, i, s, k, n) with S = 1
  if (last_pred[g] == end_index) {   // an ended beam emits END at +0; everything else is -inf (lowest tokens first)
    if (lane < per_node) {
      int tok = end_index;
      if (lane > 0) { tok = lane - 1; if (tok >= end_index) ++tok; }
      sval[base + lane] = lane == 0 ? 0.f : -INFINITY;
      sidx[base + lane] = tok;
    }
    return;
  }
  const float* p = parts + (size_t)g * ntn * 6;
  float mx = -INFINITY;
  for (int t = lane; t < ntn; t += 64) mx = fmaxf(mx, p[t * 6]);
  mx = ssc_wave_max(mx);
  float se = 0.f;
  for (int t = lane; t < ntn; t += 64) se += p[t * 6 + 1] * expf(p[t * 6] - mx);
  se = ssc_wave_sum(se);
  const float lse = mx + logf(se);
  Cand prev{INFINITY, -1};
  for (int n = 0; n < per_node; ++n) {
    Cand best{-INFINITY, -1};
    for (int c = lane; c < 2 * ntn; c += 64) {
      const float v = p[(c >> 1) * 6 + 2 + 2 * (c & 1)];
      const int i = __float_as_int(p[(c >> 1) * 6 + 3 + 2 * (c & 1)]);
      if (i >= 0 && after(v, i, prev) && (best.i < 0 || better(v, i, best))) best = Cand{v, i};
    }
    best = wave_best(best);
    if (lane == 0) { sval[base + n] = best.v - lse; sidx[base + n] = best.i; }
    prev = best;
  }
}

// part B: per (b, target state i): top-`beam` over the S*beam*per_node summed candidates      cbs.py:210-234
// ctl (optional): early stop without a host round trip, see ssc_beam_desc.
__global__ __launch_bounds__(64) void beam_merge_kernel(const float* __restrict__ sval, const int64_t* __restrict__ sidx,
                                                        const float* __restrict__ last_lp, int S, int beam, int per_node,
                                                        int64_t* __restrict__ pred, float* __restrict__ lp_out,
                                                        int64_t* __restrict__ backptr, int end_index, int* __restrict__ ctl,
                                                        int step_index, int max_steps, int* __restrict__ host_flag) {
  const int b = blockIdx.x / S, i = blockIdx.x % S;
  const int lane = threadIdx.x;
  const bool stopped = ctl && ctl[0] <= step_index;   // (written by an EARLIER launch of this stream)
  int not_ended = 0;
  if (stopped) {
    // the search had ended before this step: END at +0 from the same beam, so that nothing moves
    for (int k = lane; k < beam; k += 64) {
      const size_t o = (size_t)blockIdx.x * beam + k;
      pred[o] = end_index;
      lp_out[o] = last_lp[o];
      backptr[o] = (int64_t)i * beam + k;
    }
  } else {
    const int ncand = S * beam * per_node;
    const float* sv = sval + (size_t)blockIdx.x * ncand;
    const int64_t* si = sidx + (size_t)blockIdx.x * ncand;
    const float* ll = last_lp + (size_t)b * S * beam;
    Cand prev{INFINITY, -1};
    for (int k = 0; k < beam; ++k) {
      Cand best{-INFINITY, -1};
      for (int cidx = lane; cidx < ncand; cidx += 64) {
        float x = sv[cidx] + ll[cidx / per_node];
        if (after(x, cidx, prev) && (best.i < 0 || better(x, cidx, best))) best = Cand{x, cidx};
      }
      best = wave_best(best);
      if (lane == 0) {
        size_t o = (size_t)blockIdx.x * beam + k;
        const int64_t tok = si[best.i];
        pred[o] = tok;
        lp_out[o] = best.v;
        backptr[o] = best.i / per_node;
        not_ended += tok != end_index;
      }
      prev = best;
    }
  }
  if (ctl && lane == 0) {
    int* cnt = ctl + 2 + step_index;
    int* ticket = ctl + 2 + max_steps + step_index;
    if (not_ended) atomicAdd(cnt, not_ended);
    __threadfence();
    const int done = atomicAdd(ticket, 1);
    if (done == (int)gridDim.x - 1) {   // the last workgroup of this step
      if (!stopped) {
        __threadfence();
        const int left = atomicAdd(cnt, 0);
        if (left == 0) {
          atomicMin(ctl, step_index + 1);
          if (host_flag) __hip_atomic_store(host_flag, step_index + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      // progress word: the host of ssc_decode_search queues step t only once step t - 2 has got here (its run-ahead bound)
      if (host_flag) __hip_atomic_store(host_flag + 1, step_index, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// first step with early-stop accounting: nothing to select, only "have all beams ended after step 0?" (cbs.py:167 at timestep 0)
__global__ void beam_first_ctl_kernel(const int64_t* __restrict__ pred, int n, int end_index, int* __restrict__ ctl,
                                      int* __restrict__ host_flag) {
  __shared__ int sh[4];
  int c = 0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) c += pred[j] != end_index;
  c = block_sum_int(c, sh);
  if (threadIdx.x == 0 && c == 0) {
    atomicMin(ctl, 1);
    if (host_flag) __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void beam_backtrace_ctl_kernel(const int64_t* __restrict__ preds, const int64_t* __restrict__ backptrs,
                                          const int* __restrict__ ctl, int max_steps, int B, int SB, int end_index,
                                          int64_t* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B * SB) return;
  const int steps = min(max(ctl[0], 1), max_steps);
  int b = j / SB;
  int64_t idx = j % SB;
  size_t plane = (size_t)B * SB;
  int64_t* o = out + (size_t)j * max_steps;
  for (int t = max_steps - 1; t >= steps; --t) o[t] = end_index;
  for (int t = steps - 1; t >= 0; --t) {
    o[t] = preds[(size_t)t * plane + (size_t)b * SB + idx];
    if (t > 0) idx = backptrs[(size_t)(t - 1) * plane + (size_t)b * SB + idx];
  }
}

}  // namespace

int ssc_beam_merge(const float* sval, const int64_t* sidx, const float* last_lp, int B, int S, int beam, int per_node,
                   int64_t* pred, float* lp_out, int64_t* backptr, int end_index, int* ctl, int step_index, int max_steps,
                   int* host_flag, hipStream_t st) {
  SSC_LAUNCH(beam_merge_kernel, dim3(B * S), dim3(64), 0, st, sval, sidx, last_lp, S, beam, per_node, pred, lp_out, backptr,
             end_index, ctl, step_index, max_steps, host_flag);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" size_t ssc_fsm_tables_bytes(const ssc_fsm_dims* d) {
  if (!d || !dims_ok(*d)) return 0;
  return fsm_layout(*d).total * sizeof(int);
}

extern "C" int ssc_fsm_compile(const uint8_t* fsm, const ssc_fsm_dims* d, void* tables, size_t tables_bytes, void* stream) {
  if (!fsm || !d || !tables || !dims_ok(*d)) return SSC_EINVAL;
  const FsmLayout l = fsm_layout(*d);
  if (tables_bytes < l.total * sizeof(int)) return SSC_EWORKSPACE;
  int* w = (int*)tables;
  SSC_LAUNCH(fsm_compile_kernel, dim3(d->M * d->S), dim3(256), 0, (hipStream_t)stream, fsm, d->S, d->V, d->E, d->P, l.NW,
             w + l.ok, (uint32_t*)(w + l.dflt), (uint32_t*)(w + l.reach), w + l.nexc, w + l.etok, (uint32_t*)(w + l.emask), w + l.fill,
             (uint32_t*)(w + l.bits));
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_beam_first_fsm(const ssc_beam_desc* d, void* stream) {
  if (!d) return SSC_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  SSC_TRY(ssc_beam_first_dense(d->raw_logits != 0, d->scores, d->ld, d->fsm, d->mach, d->B, d->dims.S, d->dims.V, d->beam, d->pred,
                               d->lp_out, st));
  if (d->ctl) {
    if (d->max_steps <= 0) return SSC_EINVAL;
    SSC_LAUNCH(beam_first_ctl_kernel, dim3(1), dim3(256), 0, st, d->pred, d->B * d->dims.S * d->beam, d->end_index, d->ctl,
               d->host_flag);
    SSC_CHECK_LAUNCH();
  }
  return SSC_OK;
}

extern "C" int ssc_beam_step_fsm(const ssc_beam_desc* d, void* stream) {
  if (!d || !d->scores || !d->last_pred || !d->last_lp || !d->pred || !d->lp_out || !d->backptr || !d->scratch_val ||
      !d->scratch_idx)
    return SSC_EINVAL;
  const int B = d->B, S = d->dims.S, V = d->dims.V, beam = d->beam, per_node = d->per_node;
  if (B <= 0 || S <= 0 || V <= 0 || beam <= 0 || per_node <= 0 || per_node > V || d->ld < V || d->end_index < 0 ||
      d->end_index >= V || beam > S * beam * per_node || (!d->fsm && S != 1))
    return SSC_EINVAL;
  if (d->ctl && (d->max_steps <= 0 || d->step_index <= 0 || d->step_index >= d->max_steps)) return SSC_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool norm = d->raw_logits != 0;
  if (d->tables) {
    if (!dims_ok(d->dims) || per_node > d->dims.P || !d->fsm) return SSC_EINVAL;
    const FsmT T = fsm_view(d->tables, d->dims);
    const size_t lds = (size_t)d->dims.E * 12 + (size_t)per_node * 8;
    if (lds > 60 * 1024) return SSC_EINVAL;
    const dim3 grid(B * S * beam), blk(256);
#define SSC_ROW_FSM(NORM_, REG_)                                                                                            \
  SSC_LAUNCH((beam_row_fsm_kernel<NORM_, REG_>), grid, blk, lds, st, d->scores, d->ld, T, d->fsm, d->mach, d->last_pred,     \
             d->last_lp, S, V, beam, per_node, d->end_index, d->skip_dead, d->scratch_val, d->scratch_idx)
    if (V <= 256 * ROW_NV) {
      if (norm) SSC_ROW_FSM(true, true); else SSC_ROW_FSM(false, true);
    } else {
      if (norm) SSC_ROW_FSM(true, false); else SSC_ROW_FSM(false, false);
    }
#undef SSC_ROW_FSM
    SSC_CHECK_LAUNCH();
  } else {
    if (d->skip_dead) return SSC_EINVAL;   // (the dense kernels read every row)
    SSC_TRY(ssc_beam_rows_dense(norm, d->scores, d->ld, d->fsm, d->mach, d->last_pred, B, S, V, beam, per_node, d->end_index,
                                d->scratch_val, d->scratch_idx, st));
  }
  return ssc_beam_merge(d->scratch_val, d->scratch_idx, d->last_lp, B, S, beam, per_node, d->pred, d->lp_out, d->backptr,
                        d->end_index, d->ctl, d->step_index, d->max_steps, d->host_flag, st);
}

extern "C" int ssc_beam_step_parts(const ssc_beam_desc* d, const float* parts, void* stream) {
  if (!d || !parts || !d->last_pred || !d->last_lp || !d->pred || !d->lp_out || !d->backptr || !d->scratch_val || !d->scratch_idx)
    return SSC_EINVAL;
  const int B = d->B, V = d->dims.V, beam = d->beam, per_node = d->per_node;
  if (d->dims.S != 1 || d->fsm || d->tables || B <= 0 || V <= 0 || beam <= 0 || per_node <= 0 || per_node > 2 || d->end_index < 0 ||
      d->end_index >= V || beam > beam * per_node)
    return SSC_EINVAL;
  if (d->ctl && (d->max_steps <= 0 || d->step_index <= 0 || d->step_index >= d->max_steps)) return SSC_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int G = B * beam, ntn = ssc_cdiv(V, 128);
  SSC_LAUNCH(beam_rows_parts_kernel, dim3(ssc_cdiv(G, 4)), dim3(256), 0, st, parts, ntn, d->last_pred, G, per_node, d->end_index,
             d->scratch_val, d->scratch_idx);
  SSC_CHECK_LAUNCH();
  return ssc_beam_merge(d->scratch_val, d->scratch_idx, d->last_lp, B, 1, beam, per_node, d->pred, d->lp_out, d->backptr,
                        d->end_index, d->ctl, d->step_index, d->max_steps, d->host_flag, st);
}

extern "C" int ssc_beam_backtrace_ctl(const int64_t* preds, const int64_t* backptrs, const int* ctl, int max_steps, int B,
                                      int SB, int end_index, int64_t* out, void* stream) {
  if (!preds || !out || !ctl || max_steps <= 0 || B <= 0 || SB <= 0 || (max_steps > 1 && !backptrs)) return SSC_EINVAL;
  SSC_LAUNCH(beam_backtrace_ctl_kernel, dim3(ssc_cdiv(B * SB, 64)), dim3(64), 0, (hipStream_t)stream, preds, backptrs, ctl,
             max_steps, B, SB, end_index, out);
  SSC_CHECK_LAUNCH();
  return SSC_OK;
}

extern "C" int ssc_host_device_ptr(void* host_ptr, void** device_ptr) {
  if (!host_ptr || !device_ptr) return SSC_EINVAL;
  hipError_t e = hipHostGetDevicePointer(device_ptr, host_ptr, 0);
  if (e != hipSuccess) {
    ssc_tls_hip_error = (int)e;
    (void)hipGetLastError();
    return SSC_EHIP;
  }
  return SSC_OK;
}

// The whole constrained beam search of one diverse-decode call as ONE sequence-level entry point (ssc_decode_search):
// first step, state enlargement, then per step  decode step -> selection -> back-pointers  with the early-stop bookkeeping on the
// device - launched from here like ssc_train_fwd launches the training time loop: no Python and no framework glue between the
// steps (round 3's driver issued ~100 copy / fill / elementwise launches per call from torch around the ~25 library launches of a
// step).
// Reference: ConstrainedBeamSearch.search (updown-baseline/updown/modules/cbs.py:59-277) driving
// UpDownCaptioner._decode_step in eval mode (var_updown/var_updown/models/updown_captioner.py:371-455), as the reference's
// inference loop does per image and latent sample (var_updown/scripts/inference.py:117-189).
#include <algorithm>
#include <thread>

#include "ssc_common.h"

namespace {

inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

// The vocabulary head of the later steps leaves per-tile records instead of logits (ssc_decode_step_desc.topk_part) when the machine
// is the trivial one, at most two candidates per row are wanted and the head is one aligned 3xBF16 / 2xFP16 product.
bool search_uses_parts(const ssc_model_cfg* cfg, const ssc_search_desc* d) {
  const long G = (long)d->nimg * d->n_samples * d->S * d->beam;
  return d->S == 1 && !d->fsm && !d->tables && d->per_node <= 2 && !cfg->tied && cfg->gemm_mode != 2 && cfg->H % 4 == 0 && G >= 512 &&
         ssc_decode_parts_enabled();
}

struct SearchLayout {
  size_t st[2][4];     // h1, c1, hd, cd: two generations of (G,H)
  size_t pl[2][2];     // 2xFP16 numerics: the fp16 pieces of h1, hd - two generations of (G, Hk) words (ssc_decode_step_desc.h1_planes ...)
  size_t tokens0;      // (B) int64 start tokens
  size_t sent_rows;    // (G) float
  size_t preds;        // (max_steps, B, SB) int64
  size_t backs;        // (max_steps-1, B, SB) int64
  size_t parent0;      // (B, SB) int64 zeros: every beam of the first expanded step descends from the one start row
  size_t lp[2];        // (B, S, beam) float
  size_t sval, sidx;   // B*S*SB*per_node
  size_t alpha;        // (G, R)
  size_t logits;       // (G, V); (B, V) when the later steps leave records
  size_t parts;        // (G, ceil(V / 128), 6) records
  size_t stepws;       // ssc_decode_step workspace
  size_t stepws_bytes;
  size_t total;
};

SearchLayout search_layout(const ssc_model_cfg* cfg, const ssc_search_desc* d) {
  SearchLayout l;
  const size_t B = (size_t)d->nimg * d->n_samples, SB = (size_t)d->S * d->beam, G = B * SB;
  const size_t H = cfg->H;
  size_t o = 0;
  for (int g = 0; g < 2; ++g)
    for (int k = 0; k < 4; ++k) { l.st[g][k] = o; o += a256(G * H * 4); }
  const size_t prow = cfg->gemm_mode == 3 && !cfg->tied && G >= 512 ? (size_t)ssc_decode_planes_ld(cfg) : 0;
  for (int g = 0; g < 2; ++g)
    for (int k = 0; k < 2; ++k) { l.pl[g][k] = o; o += a256(G * prow * 4); }
  l.tokens0 = o; o += a256(B * 8);
  l.sent_rows = o; o += a256(G * 4);
  l.preds = o; o += a256((size_t)d->max_steps * G * 8);
  l.backs = o; o += a256((size_t)std::max(d->max_steps - 1, 1) * G * 8);
  l.parent0 = o; o += a256(G * 8);
  for (int g = 0; g < 2; ++g) { l.lp[g] = o; o += a256(G * 4); }
  l.sval = o; o += a256(B * d->S * SB * d->per_node * 4);
  l.sidx = o; o += a256(B * d->S * SB * d->per_node * 8);
  l.alpha = o; o += a256(G * (size_t)d->R * 4);
  const bool parts = search_uses_parts(cfg, d);
  l.logits = o; o += a256((parts ? B : G) * (size_t)cfg->V * 4);
  l.parts = o; o += a256(parts ? G * (size_t)ssc_cdiv(cfg->V, 128) * 6 * 4 : 0);
  l.stepws_bytes = ssc_decode_step_workspace_bytes(cfg, (int)G, d->R);
  l.stepws = o; o += a256(l.stepws_bytes);
  l.total = o;
  return l;
}

bool desc_ok(const ssc_model_cfg* cfg, const ssc_search_desc* d) {
  if (!cfg || !d) return false;
  if (d->nimg <= 0 || d->R <= 0 || d->n_samples <= 0 || d->S <= 0 || d->S > 32 || d->beam <= 0 || d->per_node <= 0 ||
      d->max_steps <= 0 || d->end_index < 0 || d->end_index >= cfg->V)
    return false;
  const long G = (long)d->nimg * d->n_samples * d->S * d->beam;
  if (G <= 0 || G > (1L << 24)) return false;
  if (!d->feats || !d->imgbuf || !d->eps0 || (d->max_steps > 1 && !d->eps) || !d->predictions || !d->log_probs || !d->ctl) return false;
  if (d->S > 1 && !d->fsm) return false;
  if (d->skip_dead && !d->tables && (d->S != 1 || d->fsm)) return false;   // (the one-state machine has no fill-only rows: skip_dead then only leaves ENDED beams out of the steps)
  if (cfg->kld_mode == 2 ? !d->obj_atts : ((cfg->S || cfg->pm_scale != 0.f) && !d->sentiment)) return false;
  return true;
}

__global__ void fill_i64_kernel(int64_t* __restrict__ p, size_t n, int64_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void ctl_init_kernel(int* __restrict__ ctl, int n, int max_steps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ctl[i] = i == 0 ? max_steps : 0;
}
// dst row g <- src row g / rep   (cbs.py:10-17: the start row's state for every (state, beam) of its batch entry)
__global__ void expand_rows_kernel(const float* __restrict__ src, int Wd, int rep, size_t rows, float* __restrict__ dst) {
  const size_t row = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (row < rows && x < Wd) dst[row * Wd + x] = src[(row / rep) * Wd + x];
}

}  // namespace

extern "C" size_t ssc_decode_search_workspace_bytes(const ssc_model_cfg* cfg, const ssc_search_desc* d) {
  if (!cfg || !d || d->nimg <= 0 || d->n_samples <= 0 || d->S <= 0 || d->beam <= 0 || d->per_node <= 0 || d->max_steps <= 0 ||
      d->R <= 0)
    return 0;
  return search_layout(cfg, d).total;
}

extern "C" int ssc_decode_search(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_search_desc* d, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  SscGemmModeScope mode_scope(cfg);   // the numerics mode of this cfg, for every product the call issues
  if (!p || !workspace || !desc_ok(cfg, d)) return SSC_EINVAL;
  const SearchLayout l = search_layout(cfg, d);
  if (workspace_bytes < l.total) return SSC_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* W = (char*)workspace;
  const int B = d->nimg * d->n_samples, S = d->S, beam = d->beam, SB = S * beam, G = B * SB, H = cfg->H, V = cfg->V, Z = cfg->Z;
  float* stt[2][4];
  for (int g = 0; g < 2; ++g)
    for (int k = 0; k < 4; ++k) stt[g][k] = (float*)(W + l.st[g][k]);
  int64_t* tokens0 = (int64_t*)(W + l.tokens0);
  float* sent_rows = (float*)(W + l.sent_rows);
  int64_t* preds = (int64_t*)(W + l.preds);
  int64_t* backs = (int64_t*)(W + l.backs);
  int64_t* parent0 = (int64_t*)(W + l.parent0);
  float* lp[2] = {(float*)(W + l.lp[0]), (float*)(W + l.lp[1])};
  float* alpha = (float*)(W + l.alpha);
  float* logits = (float*)(W + l.logits);
  float* parts = (float*)(W + l.parts);
  const bool use_parts = search_uses_parts(cfg, d);
  const size_t plane = (size_t)G;
  const int nctl = 2 + 2 * d->max_steps;

  SSC_LAUNCH(ctl_init_kernel, dim3(ssc_cdiv(nctl, 256)), dim3(256), 0, st, d->ctl, nctl, d->max_steps);
  SSC_CHECK_LAUNCH();
  SSC_LAUNCH(fill_i64_kernel, dim3(ssc_cdiv(B, 256)), dim3(256), 0, st, tokens0, (size_t)B, (int64_t)d->end_index);
  SSC_CHECK_LAUNCH();
  if (hipMemsetAsync(parent0, 0, (size_t)G * 8, st) != hipSuccess) return SSC_EHIP;
  for (int k = 0; k < 4; ++k)   // zero start states (cbs.py: start_state None -> updown_cell.py:131-141)
    if (hipMemsetAsync(stt[1][k], 0, (size_t)B * H * 4, st) != hipSuccess) return SSC_EHIP;

  // which form the steps take is decided once, from the extents (the decisions DecodeEngine.step makes per call)
  auto table_mode = [&](int rows, int rpi) { return (d->R <= 128 && rows >= 512 && rpi >= 16 && ssc_decode_att_table_enabled()) ? 1 : 0; };
  bool table_ready = false;
  ssc_decode_step_desc sd{};
  sd.R = d->R; sd.feats = d->feats; sd.imgbuf = d->imgbuf; sd.alpha = alpha; sd.log_probs = logits; sd.raw_logits = 1;
  sd.obj_atts = d->obj_atts;
  // ---- first step: one row per batch entry (cbs.py:127) ------------------------------------------------------------------
  sd.G = B; sd.rows_per_image = d->n_samples; sd.tokens = tokens0; sd.sentiment = d->sentiment; sd.eps = d->eps0;
  sd.h1 = stt[1][0]; sd.c1 = stt[1][1]; sd.hd = stt[1][2]; sd.cd = stt[1][3];
  sd.h1_out = stt[0][0]; sd.c1_out = stt[0][1]; sd.hd_out = stt[0][2]; sd.cd_out = stt[0][3];
  sd.att_table = table_mode(B, d->n_samples) ? 2 : 0;
  table_ready = sd.att_table != 0;
  SSC_TRY(ssc_decode_step(cfg, p, &sd, W + l.stepws, l.stepws_bytes, st));
  ssc_beam_desc bd{};
  bd.scores = logits; bd.ld = V; bd.raw_logits = 1;
  bd.fsm = d->fsm; bd.tables = d->tables; bd.dims = d->dims; bd.mach = d->mach;
  if (!d->tables) { bd.dims.M = d->mach ? 0 : B; bd.dims.S = S; bd.dims.V = V; bd.dims.E = 0; bd.dims.P = 1; }
  if (bd.dims.S != S || bd.dims.V != V) return SSC_EINVAL;
  bd.B = B; bd.beam = beam; bd.per_node = d->per_node; bd.end_index = d->end_index;
  bd.pred = preds; bd.lp_out = lp[0];
  bd.ctl = d->early_stop ? d->ctl : nullptr; bd.max_steps = d->max_steps; bd.host_flag = d->early_stop ? d->host_flag : nullptr;
  bd.scratch_val = (float*)(W + l.sval); bd.scratch_idx = (int64_t*)(W + l.sidx);
  SSC_TRY(ssc_beam_first_fsm(&bd, st));
  // ---- enlarge the states to (B, S, beam) rows (cbs.py:152-155) --------------------------------------------------------------
  if (d->max_steps > 1) {
    for (int k = 0; k < 4; ++k) {
      SSC_LAUNCH(expand_rows_kernel, dim3(ssc_cdiv(H, 256), G), dim3(256), 0, st, stt[0][k], H, SB, (size_t)G, stt[1][k]);
      SSC_CHECK_LAUNCH();
    }
    if (d->sentiment) {
      SSC_LAUNCH(expand_rows_kernel, dim3(1, G), dim3(64), 0, st, d->sentiment, 1, SB, (size_t)G, sent_rows);
      SSC_CHECK_LAUNCH();
    }
  }
  int cur = 1, a = 0;
  const int rpi = d->n_samples * SB;
  const int tmode = table_mode(G, rpi);
  const bool ung = SB > 1 && ssc_decode_ungathered_ok(cfg, d->nimg, G, SB, tmode) != 0;
  bool ungathered = false;
  sd.G = G; sd.rows_per_image = rpi; sd.sentiment = d->sentiment ? sent_rows : nullptr; sd.group = SB;
  bd.skip_dead = d->skip_dead && d->tables ? 1 : 0;
  // Early stop and the host's run-ahead.  The host queues a step in a fraction of the time the device needs for it, so a host that
  // only polls the flag has long queued every step by the time the device writes it (measured: captions that all end at step 2 still
  // cost all 20 steps).  The run-ahead is therefore bounded: step t is queued only once step t - RUN_AHEAD has completed - the last
  // workgroup of a step's merge kernel notes the step in the second pinned word (ssc_beam_desc.host_flag[1]) and the host reads it;
  // no event, no synchronisation call (an event per step cost 1.7 % of a search that never stops).  The device always has RUN_AHEAD
  // steps of work queued, and at most that many surplus steps run.  Not under stream capture (a captured search queues every step).
  constexpr int RUN_AHEAD = 2;
  bool bounded = false;
  if (d->early_stop && d->host_flag_host) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    bounded = cs == hipStreamCaptureStatusNone;
  }
  auto wait_for_step = [&](int step) {   // until the device has completed `step` (or stopped, or the stream has drained: an error upstream)
    const volatile int* hf = (const volatile int*)d->host_flag_host;
    for (unsigned spin = 1; hf[1] < step && hf[0] == 0; ++spin) {
      if ((spin & 1023u) == 0 && hipStreamQuery(st) != hipErrorNotReady) { (void)hipGetLastError(); break; }
      std::this_thread::yield();
    }
  };
  for (int t = 1; t < d->max_steps; ++t) {
    // cbs.py:167: the device notes the step after which every beam had ended and turns later steps into no-ops (ssc_beam_desc.ctl);
    // the host stops QUEUEING once it sees the flag the device wrote - a plain read of pinned memory
    if (bounded && t > RUN_AHEAD) wait_for_step(t - RUN_AHEAD);
    if (d->early_stop && d->host_flag_host && *(volatile const int*)d->host_flag_host != 0) break;
    const int64_t* last = preds + (size_t)(t - 1) * plane;
    sd.tokens = last; sd.eps = d->eps + (size_t)(t - 1) * G * Z;
    sd.h1 = stt[cur][0]; sd.c1 = stt[cur][1]; sd.hd = stt[cur][2]; sd.cd = stt[cur][3];
    sd.h1_out = stt[1 - cur][0]; sd.c1_out = stt[1 - cur][1]; sd.hd_out = stt[1 - cur][2]; sd.cd_out = stt[1 - cur][3];
    sd.parent = t == 1 ? parent0 : backs + (size_t)(t - 2) * plane;
    sd.att_table = tmode ? (table_ready ? 1 : 2) : 0;
    table_ready = table_ready || tmode;
    sd.ungathered = ungathered ? 1 : 0;
    if (l.pl[1][0] != l.pl[0][0]) {   // the states' fp16 pieces travel with the un-gathered states (a re-ordered state is split again by its step)
      sd.h1_planes_out = W + l.pl[1 - cur][0]; sd.hd_planes_out = W + l.pl[1 - cur][1];
      sd.h1_planes = ungathered ? W + l.pl[cur][0] : nullptr; sd.hd_planes = ungathered ? W + l.pl[cur][1] : nullptr;
    }
    sd.row_lp = d->skip_dead ? lp[a] : nullptr; sd.end_index = d->end_index;
    if (use_parts) { sd.log_probs = nullptr; sd.topk_part = parts; }
    SSC_TRY(ssc_decode_step(cfg, p, &sd, W + l.stepws, l.stepws_bytes, st));
    bd.last_pred = last; bd.last_lp = lp[a]; bd.pred = preds + (size_t)t * plane; bd.lp_out = lp[1 - a];
    bd.backptr = backs + (size_t)(t - 1) * plane; bd.step_index = t;
    if (use_parts) SSC_TRY(ssc_beam_step_parts(&bd, parts, st));
    else SSC_TRY(ssc_beam_step_fsm(&bd, st));
    a = 1 - a;
    if (ung) {   // the next step reads these outputs through the back-pointers (ssc_decode_step_desc.ungathered)
      cur = 1 - cur;
      ungathered = true;
    } else {     // cbs.py:236-250: re-order the states by back-pointer (into the generation the step has just consumed)
      for (int k = 0; k < 4; ++k)
        SSC_TRY(ssc_gather_rows(stt[1 - cur][k], H, bd.backptr, B, SB, H, stt[cur][k], st));
    }
  }
  if (d->early_stop) {
    SSC_TRY(ssc_beam_backtrace_ctl(preds, backs, d->ctl, d->max_steps, B, SB, d->end_index, d->predictions, st));
  } else {
    SSC_TRY(ssc_beam_backtrace(preds, backs, d->max_steps, B, SB, d->predictions, st));
  }
  if (hipMemcpyAsync(d->log_probs, lp[a], (size_t)G * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return SSC_EHIP;
  return SSC_OK;
}

/*
 * ssc.h - C ABI of libssc_hip.so: the MI355X (gfx950) Style-SeqCVAE hot path.
 *
 * The reference (visinf/style-seqcvae) has NO native boundary on this path: its hot path is stock
 * PyTorch ops behind the Python module API var_updown.models.UpDownCaptioner /
 * var_updown.modules.UpDownCell.  This header is the boundary the build creates
 * underneath that API; each entry point cites the reference code it replaces (paths relative to
 * /root/reference).
 *
 * Conventions
 *  - plain C: raw device pointers, ints, floats; no torch / C++ types.
 *  - all tensors fp32 row-major unless noted; token ids int64; `ld*` = leading dimension in floats.
 *  - every buffer is caller-allocated device memory (e.g. torch tensor .data_ptr()), kept alive by
 *    the caller until `stream` is synchronised.  The library allocates nothing and is re-entrant per
 *    stream and safe under hipGraph capture.  Process-wide state: the numerics mode of NT products
 *    (ssc_set_gemm_mode, below) and the opt-in diagnostics / tuning switches of ssc_debug.h, which are
 *    not part of this ABI.
 *  - `stream` is a hipStream_t passed as void* (0 = default stream).
 *  - return value: 0 on success, negative SSC_E* otherwise; never throws.
 */
#ifndef SSC_H
#define SSC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSC_OK 0
#define SSC_EINVAL (-1)   /* bad shape / argument */
#define SSC_EALIGN (-2)   /* pointer or leading dimension violates an alignment requirement */
#define SSC_EHIP (-3)     /* a HIP runtime call failed (see ssc_last_hip_error) */
#define SSC_EWORKSPACE (-4) /* workspace too small */

#define SSC_MAX_SEG 6

int ssc_version(void);              /* ABI version (this header: 4) */
int ssc_last_hip_error(void);       /* last hipError_t observed by this thread */
const char* ssc_arch(void);         /* "gfx950" */

/* ------------------------------------------------------------------------------------------------
 * GEMM: C[M,N] (+)= sum_s op(A_s)[M,K_s] * op(B_s)[K_s,N] (+ bias[N])     exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32).  Replaces aten::mm / aten::addmm under nn.LSTMCell, nn.Linear
 * (var_updown/var_updown/modules/updown_cell.py:146,192,196-197,227;
 *  updown-baseline/updown/modules/attention.py:69,125; var_updown/.../updown_captioner.py:444-445)
 * and their autograd backward.  K is segmented so that torch.cat inputs (updown_cell.py:143,178,211)
 * are never materialised: segment s multiplies a column block of the weight with its own source.
 *   a_kc=1: A_s is (M,K_s) row-major (lda>=K_s)   a_kc=0: A_s is (K_s,M) row-major (A given transposed)
 *   b_kc=1: B_s is (N,K_s) row-major (weights as stored, out x in)   b_kc=0: B_s is (K_s,N) row-major
 * splits>1 splits the K loop over `splits` workgroups per tile; partial slabs go to `workspace`
 * (>= splits*M*N floats) and are reduced (deterministically) by a second kernel.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* A; const float* B;
  int lda, ldb, K;
  /* Optional, read by the 2xFP16 form only (k-contiguous operands): the SAME operand already split into its two fp16 pieces by
   * ssc_split_f16 with the product's a_scale / b_scale ("planes"; ld in 4-byte words, >= K rounded up to 32, a multiple of 4;
   * 16-byte aligned).  The kernel then copies the pieces instead of forming them once per tile that reads the operand - bit-identical
   * results.  A / B stay mandatory: every other form of the product reads them.  NULL = split inside the kernel. */
  const void* A16; const void* B16;
  int lda16, ldb16;
} ssc_gemm_seg;

typedef struct {
  ssc_gemm_seg seg[SSC_MAX_SEG];
  int nseg;
  int M, N;
  int a_kc, b_kc;
  float* C; int ldc;
  const float* bias;   /* optional (N) */
  int accumulate;      /* 1: C += result */
  int splits;          /* >=1; 0 = choose automatically */
  float* workspace; size_t workspace_floats;
  /* Optional device-side row compaction (all NULL = off).  Counts and index lists live in device memory, so the host
   * needs no synchronisation to know how many rows are active; the launch covers the full M / K and workgroups beyond
   * the device count exit.  Only with 16-B aligned operands (else SSC_EALIGN).
   *   m_count: M = min(M, *m_count); a_rows: row r of the k-contiguous A operand is read from row a_rows[r];
   *   c_rows: row r of the result is written to row c_rows[r]   (NT / NN products)
   *   k_count: K = min(K, *k_count); ka_rows / kb_rows: k-row k of A / B is read from row k?_rows[k]
   *            (single-segment TN products, a_kc = b_kc = 0) */
  const int* m_count; const int* a_rows; const int* c_rows;
  const int* k_count; const int* ka_rows; const int* kb_rows;
  /* Optional, used by the 2xFP16 form only (ssc_model_cfg.gemm_mode 3): device scalars holding POWERS OF TWO the A / B operands are
   * multiplied with before they are split into two fp16 pieces (the result is multiplied with the exact inverse of their product).
   * Choose them so that the operands' largest magnitudes land near 2^6 .. 2^13 (ssc_pow2_scale): fp16 overflows at 65504, and an
   * entry whose lo piece is below 2^-14 keeps an absolute precision of 2^-25 only.  NULL = 1. */
  const float* a_scale; const float* b_scale;
  /* Optional: instead of C (which may then be NULL), every (row, 128-column tile) leaves a record of 6 floats at
   * topk_part[(row * ceil(N / 128) + tile) * 6]: the tile's maximum, sum of exp(x - maximum), best value, its column (int bits),
   * second best, its column; x includes the bias; ties go to the lower column; c_rows applies to the record's row.  The
   * vocabulary head of a decode step then never writes its (rows, V) logits: ssc_beam_step_parts selects from the records.
   * NT products with 16-byte aligned operands under the 3xBF16 / 2xFP16 numerics only (SSC_EINVAL otherwise). */
  float* topk_part;
} ssc_gemm_desc;

/* out[0] = 2^(target_log2 - ceil(log2(max |x|))) over the rows x cols block x (ld), a power of two that brings the block's largest
 * magnitude to [2^(target_log2 - 1), 2^target_log2] (1 when the block is all zero).  combine = 1: out[0] = min(out[0], that).
 * scratch: one float of device memory. */
int ssc_pow2_scale(const float* x, size_t rows, int cols, size_t ld, int target_log2, float* out, int combine, float* scratch,
                   void* stream);

/* The two fp16 pieces of x * scale[0] (rows x K, ld ldx; scale: device scalar, a power of two, NULL = 1) in the plane layout of
 * ssc_gemm_seg.A16 / B16: row r occupies ldo 4-byte words; per 32-k block 32 hi halfs (x truncated to fp16) then 32 lo halfs
 * (x - hi truncated); columns K .. roundup(K, 32) are zero.  ldo >= roundup(K, 32), ldo % 4 == 0, out 16-byte aligned.
 * rows / row_count (optional, device): only the listed rows. */
int ssc_split_f16(const float* x, int rows, int K, int ldx, const float* scale, void* out, int ldo, const int* row_list,
                  const int* row_count, void* stream);

int ssc_gemm(const ssc_gemm_desc* d, void* stream);

/* Numerics of NT products (a_kc = b_kc = 1, 16-B aligned operands): mode 1 (default) splits every fp32 operand exactly
 * into three bf16 pieces and sums the six partial products of order <= 2 on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (error ~ one fp32 rounding per product); mode 0 uses the exact-fp32 MFMA v_mfma_f32_32x32x2_f32.
 * Process-wide; returns the previous mode.  Environment default: SSC_GEMM_MODE=x3|f32. */
int ssc_set_gemm_mode(int mode);
int ssc_gemm_auto_splits(int M, int N, int ksteps); /* the split count ssc_gemm picks for splits=0 */

/* ------------------------------------------------------------------------------------------------
 * Per-sequence precompute
 * ---------------------------------------------------------------------------------------------- */
/* region mask + masked mean: UpDownCell._average_image_features (updown_cell.py:233-270) +
 * allennlp masked_mean.  feats (B,R,F) -> mask (B,R) float {0,1}, avg (B,F). */
int ssc_feat_prep(const float* feats, int B, int R, int F, float* mask, float* avg, void* stream);

/* boundary tokens + loss weights: allennlp add_sentence_boundary_token_ids as called at
 * updown_captioner.py:265-278.  caps (B,L) int64 -> tokens_tm (L+2,B) int64 time-major,
 * w_tm (T=L+1,B) float = [tokens[b,t+1] != pad], nvalid (B) float = sum_t w. */
int ssc_prep_tokens(const int64_t* caps, int B, int L, int pad, int boundary, int64_t* tokens_tm, float* w_tm,
                    float* nvalid, void* stream);

/* nn.Embedding forward (updown_captioner.py:430): out[i,:] = table[ids[i],:]  (n rows, E cols). */
int ssc_embed_gather(const float* table, int ldt, const int64_t* ids, int n, int E, float* out, int ldo, void* stream);
/* nn.Embedding backward: dtable[ids[i],:] += d[i,:] for ids[i] != pad (padding_idx row gets no grad). */
int ssc_embed_scatter_add(float* dtable, int ldt, const int64_t* ids, int n, int E, const float* d, int ldd, int pad,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * LSTM cell epilogue: torch.nn.LSTMCell pointwise part (gate order i,f,g,o), fused with the split-K
 * slab reduction, the hoisted time-invariant gate terms, both biases and the rank-1 sentiment
 * column (updown_cell.py:146-148,192-194,227-229).
 *   pre[b,n] = sum_{s<nslab} slabs[s][b,n] + add0[row0(b),n] + add1[row1(b),n] + b_ih[n] + b_hh[n]
 *              + sent[b]*wcol[n*ldwcol]
 *   gates_out (B,4H) = activated (i,f,g,o);  c_out = f*c_prev + i*g;  h_out = o*tanh(c_out)
 * add1 is indexed by b / rows_per_add1 (decode: per-image hoisted term).  Any optional pointer may be 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int B, H;
  const float* slabs; int nslab; size_t slab_stride; /* each slab (B,4H), ld 4H */
  const float* add0; int ld_add0;
  const float* add1; int ld_add1; int rows_per_add1;
  const float* b_ih; const float* b_hh;
  const float* sent; const float* wcol; int ldwcol;
  const float* c_prev; int ld_cprev;
  float* gates_out;            /* (B,4H) activated, ld 4H; may be 0 */
  float* c_out; int ld_cout;
  float* h_out; int ld_hout;
  const int64_t* add0_rows;    /* optional: add0 is indexed by add0_rows[b] instead of b (decode: a per-token table of the
                                * embedding's gate contribution, row = the beam's last token) */
  /* decode, beams that share their parent (ssc_decode_step_desc.parent): row b's slab values are read from row slab_rows[b] of
   * `slabs` (a product formed on the distinct parents only); slabs2: a second slab list with its own row index (0 = row b) */
  const int* slab_rows;
  const float* slabs2; int nslab2; size_t slab2_stride; const int* slab2_rows;
  const int* c_prev_rows;      /* optional: row b's previous cell state is row c_prev_rows[b] of c_prev (decode: states left in the
                                * previous step's row order, ssc_decode_step_desc.ungathered) */
  const int* rows; const int* row_count;   /* optional (ssc_lstm_fwd only): a device-side list of the rows to compute, *row_count of them
                                * (decode: the rows that hold a finite, unfinished beam); the other rows' outputs are left as they are */
  /* optional (ssc_lstm_fwd, ssc_lstm_fwd_img): h_out * planes_scale[0] also leaves the cell split into its two fp16 pieces, in the
   * plane layout of ssc_split_f16 (ld_hplanes 4-byte words per row, >= H rounded up to 32; the padding columns are zeroed) - the
   * operand of the next 2xFP16 product (ssc_gemm_seg.A16) without a pass of its own.  planes_scale NULL = 1.  SSC_EINVAL from
   * ssc_lstm_fwd_z / _p and from the VALU form of ssc_lstm_fwd_img (H % 4 != 0), which do not write them. */
  void* h_planes; int ld_hplanes; const float* planes_scale;
} ssc_lstm_fwd_desc;
int ssc_lstm_fwd(const ssc_lstm_fwd_desc* d, void* stream);
/* The same with one more addend formed inside the kernel: pre[b,n] += z[b,:Z] . wz[n,:Z]  (z (B,Z) ld ldz; wz (4H,Z) ld ldwz;
 * exact-fp32 MFMA).  Used for the latent block of the decoder LSTM input (updown_cell.py:211-229): z exists only after the
 * latent head of the same step, the rest of the gate product does not wait for it. */
int ssc_lstm_fwd_z(const ssc_lstm_fwd_desc* d, const float* z, int ldz, const float* wz, int ldwz, int Z, void* stream);
/* ssc_lstm_fwd that also leaves partial products of its output with an nn.Linear weight wp (NP <= 256 rows of H, ld ldwp):
 *   pout[s][b,n] = sum_{j in [16 s, 16 s + 16)} h_out[b,j] wp[n,j],   s < ceil(H/16), each slab (B,NP) ld NP
 * (the encoder LSTM's h feeds fc_mean | fc_log_var in the same step, updown_cell.py:196-197: ssc_latent_fwd sums the slabs). */
int ssc_lstm_fwd_p(const ssc_lstm_fwd_desc* d, const float* wp, int ldwp, int NP, float* pout, void* stream);
/* ssc_lstm_fwd for rows that share per-image operands (decode), with one more addend from a per-image table:
 *   pre[b,n] += sum_r alpha[b,r] P[(img(b) R + r) 4H + n],  img(b) = b / rows_per_image;  alpha (B,R) ld ldalpha; R <= 128.
 * With P[img,r,:] = W_ih^dec[:, :F] v_{img,r} (ssc_decode_prepare) this IS the attended-feature segment of the decoder gate
 * product (updown_cell.py:156-158,211-229), by linearity of the product in att = sum_r alpha_r v_r. */
int ssc_lstm_fwd_img(const ssc_lstm_fwd_desc* d, const float* alpha, int ldalpha, const float* P, int R, int rows_per_image,
                     void* stream);

/* LSTMCell pointwise backward (SURVEY Appendix A.4 "LSTM^-1"):
 *   dh (B,H) (+ dh2 optional second addend), dc_in (B,H), gates (activated), c_prev, c_new
 *   -> dG (B,4H) pre-activation grads, dc_prev (B,H).  If dgsum != 0: dgsum += dG. */
typedef struct {
  int B, H;
  const float* dh; int ld_dh;
  const float* dh2; int ld_dh2;
  const float* dc_in; int ld_dcin;
  const float* gates;
  const float* c_prev; int ld_cprev;
  const float* c_new; int ld_cnew;
  float* dG;                   /* (B,4H) ld 4H */
  float* dc_prev; int ld_dcprev;
  float* dgsum;                /* optional (B,4H) running sum over time */
  /* optional extra addends of dh given as split-K partial slabs ((B,H), ld H each) of the GEMMs that produce them:
   * dh += sum_s slabsA[s] + sum_s slabsB[s]  (fixed summation order) */
  const float* slabsA; int nA; size_t strideA;
  const float* slabsB; int nB; size_t strideB;
} ssc_lstm_bwd_desc;
int ssc_lstm_bwd(const ssc_lstm_bwd_desc* d, void* stream);
/* The same with one more addend of dh formed inside the kernel: dh[b,j] += sum_k x[b,k] w[k,j]  (x (B,K) ld ldx; w (K,H) ld ldw;
 * exact-fp32 MFMA; K <= 768: its LDS images take (48 (K+4) + 2176) floats).  BPTT of the
 * encoder LSTM: x = (dmu | dlv), w = [W_mu ; W_lv] (fc_mean / fc_log_var, updown_cell.py:196-197); of the attention LSTM:
 * x = dq, w = Wq (the query projection, attention.py:69). */
int ssc_lstm_bwd_x(const ssc_lstm_bwd_desc* d, const float* x, int ldx, const float* w, int ldw, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bottom-up top-down attention step: BottomUpTopDownAttention.forward after the q projection
 * (attention.py:78-95) + allennlp masked_softmax + the weighted sum (updown_cell.py:156-158).
 *   logit[g,r] = wa . tanh(q[g] + pv[img(g),r]);  alpha = masked_softmax(logit, mask[img(g)])
 *   att[g,:]  = sum_r alpha[g,r] feats[img(g),r,:]          img(g) = g / rows_per_image
 * ---------------------------------------------------------------------------------------------- */
int ssc_attn_logits(const float* q, int ldq, const float* pv, const float* wa, int G, int R, int A, int rows_per_image,
                    float* logits, void* stream);
/* logits: scratch (G,R); alpha (G,R); R <= 256 */
int ssc_attn_fwd(const float* q, int ldq, const float* pv, const float* wa, const float* mask, const float* feats,
                 int G, int R, int A, int F, int rows_per_image, float* logits, float* alpha, float* att, int ldatt,
                 void* stream);

/* out (G,D) = sum_r alpha[g,r] x[img(g),r,:]: attention pooling of a per-region tensor x (nimg,R,D) with the step's weights -
 * the grounded style prior of SENTIMENT_VAE = 2 (updown_cell.py:160-163: per-region attribute means obj_atts, D = 150). */
/* the attention weights alone (no weighted feature sum): decode consumes alpha through ssc_lstm_fwd_img */
int ssc_attn_weights(const float* q, int ldq, const float* pv, const float* wa, const float* mask, int G, int R, int A,
                     int rows_per_image, float* logits, float* alpha, void* stream);

int ssc_attn_pool(const float* alpha, const float* x, int G, int R, int D, int rows_per_image, float* out, int ldo,
                  void* stream);
/* ssc_attn_fwd that also pools a second per-region tensor with the same weights, in the same launch:
 *   pool[g, :D] = sum_r alpha[g,r] obj[img(g),r,:D]   (obj (nimg,R,D); pool (G, ldpool), columns D..ldpool-1 set to 0) */
int ssc_attn_fwd_pool(const float* q, int ldq, const float* pv, const float* wa, const float* mask, const float* feats,
                      int G, int R, int A, int F, int rows_per_image, float* logits, float* alpha, float* att, int ldatt,
                      const float* obj, int D, float* pool, int ldpool, void* stream);

/* Attention backward: datt (G,F) -> dq (G,A), dpv_acc (G,R,A) += dpre,
 * dwa_acc (G,A) += sum_r dl_r u_r (caller sums over G at the end).  Training only (rows_per_image=1). */
int ssc_attn_bwd(const float* datt, int lddatt, const float* q, int ldq, const float* pv, const float* wa,
                 const float* alpha, const float* feats, int G, int R, int A, int F, float* dq, int lddq,
                 float* dpv_acc, float* dwa_acc, float* scratch_dalpha, void* stream);
/* The same when the attention weights also pooled obj (ssc_attn_fwd_pool): dalpha[g,r] += (dpool_a[g,:Da] + dpool_b[g,:D]) . obj[g,r,:D]
 * (two addends of the pooled tensor's gradient, e.g. the LSTM input gradients and the KL term; dpool_b may be 0; Da <= D: dpool_a
 * covers the first Da entries only - the conditioning block is the whole pooled vector or its first entry, updown_cell.py:169-172). */
int ssc_attn_bwd_pool(const float* datt, int lddatt, const float* q, int ldq, const float* pv, const float* wa,
                      const float* alpha, const float* feats, int G, int R, int A, int F, float* dq, int lddq,
                      float* dpv_acc, float* dwa_acc, float* scratch_dalpha, const float* obj, int D, const float* dpool_a,
                      int lddpa, int Da, const float* dpool_b, int lddpb, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Latent head epilogue: fc_mean / fc_log_var bias add, reparameterised sample and closed-form KL
 * (updown_cell.py:196-208, updown_captioner.py:295-303).
 *   mulv (B,2Z) raw [h_e Wmu^T | h_e Wlv^T] (ld ldmulv);  mu = mulv[:, :Z]+bmu;  lv = mulv[:, Z:]+blv
 *   z = eps*exp(lv/2)+mu;  kld_t[b] per formula (mode 0: vs N(0,1); mode 1: vs N(prior_mean, prior_var+1e-5))
 *   kld_acc[b] += w[b]*kld_t[b].   prior_mean per row = pm_scale*sent[b] (or 0), prior_var scalar.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int B, Z;
  const float* mulv; int ldmulv; int nslab; size_t slab_stride;
  const float* bmu; const float* blv;
  const float* eps; int ldeps;
  int kld_mode;                 /* 0: SENTIMENT_VAE==0 ; 1: otherwise */
  const float* sent; float pm_scale; float prior_var;
  const float* w;               /* (B) step weights w_bt */
  float* mu; float* lv; float* z; int ldz; /* mu, lv, z: (B, ldz) */
  float* kld_acc;               /* (B) */
  const float* pm; int ldpm;    /* optional per-row, per-dimension prior mean (B, ldpm) instead of pm_scale*sent: SENTIMENT_VAE = 2,
                                 * the attention-pooled attribute means of the step (updown_cell.py:160-163; kld_mode 1) */
} ssc_latent_fwd_desc;
int ssc_latent_fwd(const ssc_latent_fwd_desc* d, void* stream);

/* eval-mode sample: z = eps*sqrt(prior_var) + prior_mean (updown_cell.py:200-208). */
int ssc_latent_prior_sample(const float* eps, int ldeps, const float* sent, float pm_scale, float prior_var, int G, int Z,
                            float* z, int ldz, void* stream);

/* the same with a per-row, per-dimension prior mean pm (G, ldpm) - SENTIMENT_VAE = 2, where the prior mean of a step is the
 * attention-pooled attribute means (updown_cell.py:160-163,200-208) - and / or variance pv (G, ldpv); a NULL pm / pv falls back to
 * pm_scale * sent / prior_var. */
int ssc_latent_prior_sample_pm(const float* eps, int ldeps, const float* pm, int ldpm, const float* pv, int ldpv, const float* sent,
                               float pm_scale, float prior_var, int G, int Z, float* z, int ldz, void* stream);

/* latent backward (Appendix A.4): dz -> dmulv (B,2Z) = [dmu | dlv];  k[b] = gk[b]*w[b]. */
typedef struct {
  int B, Z;
  const float* dz; int lddz;
  const float* eps; int ldeps;
  const float* mu; const float* lv; int ldz;
  int kld_mode; const float* sent; float pm_scale; float prior_var;
  const float* w; const float* gk; /* (B) step weights, (B) upstream grad of kld_b */
  float* dmulv; int lddmulv;
  int nslab; size_t slab_stride; /* nslab > 1: dz is the first of nslab split-K slabs ((B, lddz) each) to be summed */
  const float* pm; int ldpm;     /* optional per-element prior mean (see ssc_latent_fwd_desc) ... */
  float* dpm; int lddpm;         /* ... and its gradient from the KL term, WRITTEN: dpm = -k (mu - pm) / (prior_var + 1e-5) */
} ssc_latent_bwd_desc;
int ssc_latent_bwd(const ssc_latent_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Vocabulary cross-entropy: allennlp sequence_cross_entropy_with_logits(average=None) times the
 * target length (updown_captioner.py:457-466).  logits (T*B, V) time-major rows (row = t*B+b).
 *   fwd: nll[row] = lse - logits[row,target];  loss[b] = n_b * sum_t w*nll / (n_b + 1e-13)
 *        `lse` must hold 2*T*B floats: [lse | w*nll]
 *   bwd (in place): logits[row,:] <- (softmax - onehot) * gl[b] * w[row] * n_b/(n_b+1e-13)
 * ---------------------------------------------------------------------------------------------- */
int ssc_ce_fwd(const float* logits, int ldl, const int64_t* targets, const float* w, const float* nvalid, int T, int B,
               int V, float* lse, float* loss, void* stream);
int ssc_ce_bwd(float* logits, int ldl, const int64_t* targets, const float* w, const float* nvalid, const float* lse,
               const float* gl, int T, int B, int V, void* stream);
/* row-wise log_softmax (updown_captioner.py:450, nn.LogSoftmax(dim=1)); in place allowed. */
int ssc_log_softmax(const float* logits, int ldl, int rows, int V, float* out, int ldo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small reductions / elementwise
 * ---------------------------------------------------------------------------------------------- */
/* out[n] (+)= sum_rows wrow[row]*X[row,n]  (bias grads: wrow=0 -> weights 1; sentiment column grads) */
int ssc_colsum(const float* X, int ldx, int rows, int N, const float* wrow, float* out, int out_stride, int accumulate,
               void* stream);
/* same, two-stage for many rows: scratch >= 64*N floats; out2 (optional, stride 1) receives a second copy
 * (nn.LSTMCell's bias_ih / bias_hh gradients are the same vector). */
int ssc_colsum2(const float* X, int ldx, int rows, int N, const float* wrow, float* out, int out_stride, float* out2,
                int accumulate, float* scratch, void* stream);
/* dst[i] = src[i*stride]: one weight column made contiguous (the rank-1 sentiment column, updown_cell.py:181-184) */
int ssc_copy_strided(const float* src, size_t stride, int n, float* dst, void* stream);
/* y = tanh(x + bias) (tied output projection, updown_captioner.py:115-117) and its backward dy*(1-y^2) */
int ssc_bias_tanh(float* x, int ldx, int rows, int N, const float* bias, void* stream);
int ssc_tanh_bwd(float* dy, int lddy, const float* y, int ldy, int rows, int N, void* stream);
int ssc_fill(float* p, size_t n, float v, void* stream);

/* clip_grad_norm_ + SGD(momentum, weight_decay) (var_updown/scripts/train.py:126-131,173-175) on flat buffers.
 *   ssc_sq_norm: partial sums of g^2 -> out (1 float, accumulated deterministically via 2 passes; scratch>=1024 floats)
 *   ssc_sgd_step: g' = g*gscale*min(1, max_norm/(sqrt(*sqnorm)*gscale+1e-6)); d = g' + wd*p;
 *                 buf = first ? d : mom*buf + d;  p -= lr*buf.     gscale folds the 1/world_size of DP. */
int ssc_sq_norm(const float* g, size_t n, float* scratch, float* out, void* stream);
int ssc_sgd_step(float* p, const float* g, float* buf, size_t n, const float* sqnorm, float gscale, float max_norm,
                 float lr, float momentum, float weight_decay, int first, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Model / sequence level: the T-step teacher-forced training forward and its BPTT
 * (UpDownCaptioner.forward training branch, updown_captioner.py:228-323; _decode_step :371-455;
 *  UpDownCell.forward updown_cell.py:86-231; backward = what autograd derives, Appendix A.4).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int V, E, H, A, F, Z;
  int S;            /* conditioning columns on the language LSTMs (updown_cell.py:47-81): 0, 1 (the sentiment column) or, with
                     * kld_mode 2 (SENTIMENT_VAE = 2: the attention-pooled attribute means, Z wide), Z (LATENT_EMBEDDING "glove") or
                     * 1 ("senti_word_net": their first entry, updown_cell.py:169-172) */
  int tied;         /* 1: frozen tied embedding + Linear/Tanh projection (updown_captioner.py:112-119) */
  int kld_mode;     /* 0: SENTIMENT_VAE==0 formula; 1: otherwise (updown_captioner.py:298-303); 2: the formula of 1 with the
                     * prior mean of step t = sum_r alpha_tr obj_atts_r (SENTIMENT_VAE = 2, updown_cell.py:160-163) */
  float pm_scale;   /* prior_mean = pm_scale * sentiment (0 for SENTIMENT_VAE 0 / SIMPLE_VAE) */
  float prior_var;  /* PRIOR_STD^2 */
  int pad, boundary;
  int gemm_mode;    /* numerics of the 16-byte aligned NT / NN / TN products issued by the sequence-level calls made with THIS cfg
                     * (ssc_train_*, ssc_decode_*): 0 = the process default (ssc_set_gemm_mode), 1 = 3xBF16 (three bf16 pieces per
                     * fp32 operand, six partial products on the bf16 matrix cores, fp32 accumulate), 2 = exact-fp32 MFMA */
} ssc_model_cfg;

/* Parameter (or gradient) table: device pointers + leading dimensions of 2-D weights. */
typedef struct {
  float* emb; int ld_emb;                         /* _embedding_layer.weight (V,E) */
  float* att_w_ih; int ld_att_w_ih;               /* (4H, E+F+2H) */
  float* att_w_hh; int ld_att_w_hh;               /* (4H, H) */
  float* att_b_ih; float* att_b_hh;               /* (4H) */
  float* wq; int ld_wq;                           /* (A,H) */
  float* wv; int ld_wv;                           /* (A,F) */
  float* wa;                                      /* (A) */
  float* enc_w_ih; int ld_enc_w_ih;               /* (4H, F+2H+S) */
  float* enc_w_hh; int ld_enc_w_hh;
  float* enc_b_ih; float* enc_b_hh;
  float* dec_w_ih; int ld_dec_w_ih;               /* (4H, F+2H+S+Z) */
  float* dec_w_hh; int ld_dec_w_hh;
  float* dec_b_ih; float* dec_b_hh;
  float* fc_mean_w; int ld_fc_mean_w;             /* (Z,H) */
  float* fc_mean_b;
  float* fc_lv_w; int ld_fc_lv_w;
  float* fc_lv_b;
  float* out_w; int ld_out_w;                     /* untied: _output_layer.weight (V,H); tied: unused */
  float* out_b;                                   /* untied: (V) */
  float* proj_w; int ld_proj_w;                   /* tied: _output_projection.0.weight (E,H) */
  float* proj_b;                                  /* tied: (E) */
} ssc_params;

typedef struct {
  int B, R, L;                 /* minibatch rows, regions, caption length (T = L+1 steps) */
  const float* feats;          /* (B,R,F) */
  const int64_t* caps;         /* (B,L) 0-padded, no boundary tokens */
  const float* sentiment;      /* (B) (ignored when cfg.S==0 and pm_scale==0) */
  const float* eps;            /* (T,B,Z) standard normal noise, ld Z */
  const float* obj_atts;       /* (B,R,S) per-region attribute means; kld_mode 2 only (else ignored, may be 0) */
} ssc_batch;

size_t ssc_train_workspace_bytes(const ssc_model_cfg* cfg, int B, int R, int L);

/* forward: fills loss (B), kld (B); keeps activations in `workspace` for ssc_train_bwd. */
int ssc_train_fwd(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* batch, void* workspace,
                  size_t workspace_bytes, float* loss, float* kld, void* stream);

/* backward: gl (B), gk (B) = upstream grads of loss_b, kld_b.  Gradients are WRITTEN (not
 * accumulated) into `g` (same layout as p); a null pointer in `g` skips that parameter's gradient
 * (frozen decoder LSTM, train.py:156-161; frozen tied embedding).  */
int ssc_train_bwd(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* batch, void* workspace,
                  size_t workspace_bytes, const float* gl, const float* gk, const ssc_params* g, void* stream);

/* The same backward in stream-ordered phases (bit mask): 1 = vocabulary head + BPTT time loop, or its two halves
 * 16 = vocabulary head (output-head gradients final) and 32 = BPTT time loop; 2 = embedding / attention-LSTM /
 * attention-projection gradients (or its halves 64 = the embedding gradient alone, 128 = the rest), 4 = encoder-LSTM and
 * latent-head gradients, 8 = decoder-LSTM gradients; 16 then 32 first, then 2 (64, 128) / 4 / 8 in any order.  Lets the caller overlap the RCCL all-reduce of a finished gradient range with the next phases. */
int ssc_train_bwd_phases(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_batch* batch, void* workspace,
                         size_t workspace_bytes, const float* gl, const float* gk, const ssc_params* g, unsigned phases,
                         void* stream);

/* read-back of saved per-step activations for tests: which = 0:h1 1:c1 2:h_enc 3:c_enc 4:h_dec 5:c_dec
 * (each (T+1,B,H), index 0 = initial zeros), 6: alpha (T,B,R), 7: mu (T,B,Zp), 8: lv (T,B,Zp), 9: logits (T*B,V),
 * 10: tokens (L+2,B) int64, 11: att (T,B,F).  Returns pointer into the workspace (and its ld) or 0. */
void* ssc_train_workspace_view(const ssc_model_cfg* cfg, int B, int R, int L, void* workspace, int which, int* ld);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange without RCCL: direct reduce-scatter + all-gather over peer-mapped buffers (hipIpc), all
 * xGMI links of a GPU busy at once (replaces nn.DataParallel's reduce-add, var_updown/scripts/train.py:123-124).
 * buf[j] / flags[j]: rank j's flat fp32 buffer and its flag block (3*SSC_XGMI_MAX_RANKS uint32, zero-initialised) as mapped in
 * THIS process (entry `rank` = the local ones).  ssc_xgmi_allreduce enqueues, on `stream`, the in-place sum over all ranks of
 * floats [lo, hi) (multiples of 4): every rank calls it with the same (lo, hi, seq), seq strictly increasing from call to call.
 * Waits are bounded by `timeout` polls (0 = default); a rank that gives up writes a non-zero stage number to *err (device int).
 * ---------------------------------------------------------------------------------------------- */
#define SSC_XGMI_MAX_RANKS 8
typedef struct {
  int world, rank;
  float* buf[SSC_XGMI_MAX_RANKS];
  unsigned* flags[SSC_XGMI_MAX_RANKS];
} ssc_xgmi_comm;
int ssc_xgmi_enable_peer(int peer_device);   /* hipDeviceEnablePeerAccess from the current device (idempotent) */
/* hipIpc plumbing: export the allocation containing `ptr` (64-byte handle + ptr's offset in it); open a handle in another
 * process UNDER THE CALLER'S CURRENT DEVICE (the device whose kernels will read the mapping); close it again. */
int ssc_xgmi_ipc_export(const void* ptr, void* handle64, size_t* offset);
int ssc_xgmi_ipc_open(const void* handle64, void** base_out);
int ssc_xgmi_ipc_close(void* base);
int ssc_xgmi_peek(const void* src, void* dst_host, size_t bytes);   /* hipMemcpy D2H from a (peer-mapped) device pointer */
int ssc_xgmi_allreduce(const ssc_xgmi_comm* c, size_t lo, size_t hi, unsigned seq, unsigned timeout, int* err, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Eval-mode decode step (UpDownCaptioner._decode_step with training=False, updown_captioner.py:371-455;
 * UpDownCell.forward training=False branch, updown_cell.py:200-229).  G rows, batch-major:
 * row g belongs to image g / rows_per_image; per-image terms (mask, avg, pv, hoisted gate term)
 * come from ssc_decode_prepare and are computed once per image instead of per step.
 * ---------------------------------------------------------------------------------------------- */
size_t ssc_decode_image_bytes(const ssc_model_cfg* cfg, int nimg, int R);
int ssc_decode_prepare(const ssc_model_cfg* cfg, const ssc_params* p, const float* feats, int nimg, int R, void* imgbuf,
                       size_t imgbuf_bytes, void* stream);
/* The same; `prev_imgbuf` (optional) is the image buffer of an EARLIER context (prev_nimg images, prev_R regions) prepared with
 * the SAME parameter values - the caller's promise: weights fixed over an inference run -; what depends on the weights alone (the
 * per-token gate table, V x 4H) is copied from it instead of being formed again. */
int ssc_decode_prepare_from(const ssc_model_cfg* cfg, const ssc_params* p, const float* feats, int nimg, int R, void* imgbuf,
                            size_t imgbuf_bytes, const void* prev_imgbuf, int prev_nimg, int prev_R, void* stream);

typedef struct {
  int G, R, rows_per_image;
  const float* feats;        /* (nimg,R,F) */
  const void* imgbuf;        /* from ssc_decode_prepare */
  const int64_t* tokens;     /* (G) previous predictions */
  const float* sentiment;    /* (G) per-row sentiment */
  const float* eps;          /* (G,Z) ld Z */
  /* states in (G,H) each, ld H; h_encoder / c_encoder are carried untouched by the caller */
  const float* h1; const float* c1; const float* hd; const float* cd;
  float* h1_out; float* c1_out; float* hd_out; float* cd_out;
  float* alpha;              /* (G,R) */
  float* log_probs;          /* (G,V) ld V; NULL: stop after the cell (UpDownCell.forward) */
  int raw_logits;            /* 1: leave the vocabulary logits un-normalised in log_probs (for ssc_beam_*_logits) */
  int emb_override;          /* 1: p->emb of THIS call is not the embedding ssc_decode_prepare saw (a caller that hands token
                              * embeddings instead of ids, UpDownCell.forward): the per-token gate table of the image context
                              * is not used, the embedding goes through the gate product */
  const int64_t* parent;     /* optional (G): after a beam re-ordering, parent[g] = index WITHIN row g's group of `group` consecutive rows of
                              * the beam it descends from (the back-pointer of cbs.py:231).  Rows of a group with equal parent[] hold
                              * identical recurrent states, so the products fed only by h1 / hd of the previous step are formed on
                              * the distinct parents (device-side row lists) and every beam reads its parent's row: at beam 5 /
                              * per-node 2 a third of the rows of those products go away.  0 = every row on its own */
  int group;                 /* rows per group (S * beam); used with `parent` */
  int att_table;             /* attended-feature term of the decoder gates (updown_cell.py:156-158,211-229): 0 = weighted feature sum +
                              * K = F segment of the gate product; 1 = from the per-image table P[img,r,:] = W_ih^dec[:, :F] v_r in the
                              * image buffer (ssc_lstm_fwd_img, K = R; R <= 128); 2 = form that table first (once per image
                              * context, by the first step that uses it), then as 1.  Pays from ~500 rows with >= 16 rows per image */
  int ungathered;            /* 1: h1 / c1 / hd / cd are the previous step's OUTPUTS in that step's row order - the caller has not
                              * re-ordered them by back-pointer (cbs.py:236-250); row g's previous state is row
                              * (g - g % group) + parent[g].  Every reader goes through the row lists the parent sharing builds
                              * anyway, so the four (G,H) gathers per step go away.  Only where ssc_decode_ungathered_ok() says
                              * so (parent sharing and the attended-feature table both in use); SSC_EINVAL otherwise */
  const float* row_lp;       /* optional (G): the running log-prob of every row's beam.  A row with row_lp <= -1e19 (no finite beam: what
                              * ssc_beam_desc.skip_dead scores without its logits) or whose token is end_index (an ended beam re-emits
                              * END whatever its logits are, cbs.py:177-181) is SKIPPED: no product is formed for it, its outputs
                              * (states, alpha, log_probs row) are unspecified.  Honoured where parent sharing and the attended-feature
                              * table are in use (untied head); elsewhere every row is computed */
  int end_index;             /* used with row_lp */
  const float* obj_atts;     /* cfg->kld_mode 2 (SENTIMENT_VAE = 2): per-region attribute means (nimg, R, Z) of the image context.  The step's
                              * prior mean is their attention-weighted sum (updown_cell.py:160-163), z = eps sqrt(prior_var) + that
                              * (:200-208), and its leading cfg->S entries (all Z: LATENT_EMBEDDING "glove"; 1: "senti_word_net",
                              * :169-172) condition the decoder LSTM (:219-222) */
  float* prior_mean_out;     /* optional (G, Z) ld Z: the pooled prior mean of every row (what the cell returns, updown_cell.py:231) */
  const float* prior_mean;   /* optional (G, Z) ld Z: the caller's own prior mean instead of pm_scale * sentiment (_decode_step's
                              * prior_mean argument, updown_captioner.py:371-381); ignored with kld_mode 2, whose cell replaces it */
  const float* prior_var;    /* optional (G, Z) ld Z: the caller's own prior variance instead of cfg->prior_var */
  float* topk_part;          /* optional (G, ceil(V / 128), 6): the vocabulary head leaves per-tile records (ssc_gemm_desc.topk_part) here INSTEAD
                              * of writing log_probs (which may then be NULL); untied head, 16-byte aligned H, not under the exact-fp32
                              * numerics (SSC_EINVAL otherwise).  For ssc_beam_step_parts */
  /* 2xFP16 numerics, large calls (cfg->gemm_mode 3, parent sharing + attended-feature table in use): the step's products read the
   * recurrent states already split into their fp16 pieces (ssc_split_f16 layout, ssc_decode_planes_ld(cfg) words per row, G rows).
   * h1_planes_out / hd_planes_out (optional): where the cells leave the pieces of h1_out / hd_out (default: the step workspace);
   * h1_planes / hd_planes (optional): the pieces of h1 / hd, i.e. what the PREVIOUS step left in its *_planes_out - without them the
   * step splits h1 / hd itself.  Ignored by every other form of the step. */
  const void* h1_planes; const void* hd_planes;
  void* h1_planes_out; void* hd_planes_out;
} ssc_decode_step_desc;
/* words (4 bytes) per row of the state pieces above: H rounded up to 32 */
int ssc_decode_planes_ld(const ssc_model_cfg* cfg);
/* 1 if a step of G rows in groups of `group` (0: group size not known yet - any divisor of G above 1 will do) over an image context of
 * nimg images with this att_table mode can take un-gathered states */
int ssc_decode_ungathered_ok(const ssc_model_cfg* cfg, int nimg, int G, int group, int att_table);
size_t ssc_decode_step_workspace_bytes(const ssc_model_cfg* cfg, int G, int R);
int ssc_decode_step(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_decode_step_desc* d, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Constrained beam search bookkeeping on device (updown-baseline/updown/modules/cbs.py:59-277).
 *   ssc_beam_first : first step (:127-145): per (b, state s) top-`beam` of log_probs[b,:] masked by fsm[b,0,s,:]
 *   ssc_beam_step  : one later step (:170-234): ended beams forced to end_index, per target state i the
 *                    masked (-1e20) top-`per_node` per row, + running log-prob, top-`beam` over S*beam*per_node;
 *                    backpointer = idx / per_node (floor).  Ties resolve to the lowest index.
 *   ssc_gather_rows: state re-ordering by backpointer (:236-250).
 * fsm (B,S,S,V) uint8, or NULL with S = 1 for the trivial one-state machine (every transition allowed; what
 * MAX_GIVEN_CONSTRAINTS: 0 produces).  Row order (batch, fsm_state, beam).
 * ---------------------------------------------------------------------------------------------- */
int ssc_beam_first(const float* log_probs, int ldlp, const uint8_t* fsm, int B, int S, int V, int beam,
                   int64_t* pred, float* lp_out, void* stream);
int ssc_beam_step(const float* log_probs, int ldlp, const uint8_t* fsm, const int64_t* last_pred, const float* last_lp,
                  int B, int S, int V, int beam, int per_node, int end_index, int64_t* pred, float* lp_out,
                  int64_t* backptr, float* scratch_val, int64_t* scratch_idx, void* stream);
/* Same selections from UN-normalised vocabulary logits: every row's log-sum-exp is taken inside the selection kernel
 * (row staged in LDS, same arithmetic and summation order as ssc_log_softmax followed by the calls above, so the results
 * are bit-identical) - saves a full write + read of the (G,V) log-probability matrix per step. */
int ssc_beam_first_logits(const float* logits, int ldlp, const uint8_t* fsm, int B, int S, int V, int beam,
                          int64_t* pred, float* lp_out, void* stream);
int ssc_beam_step_logits(const float* logits, int ldlp, const uint8_t* fsm, const int64_t* last_pred, const float* last_lp,
                         int B, int S, int V, int beam, int per_node, int end_index, int64_t* pred, float* lp_out,
                         int64_t* backptr, float* scratch_val, int64_t* scratch_idx, void* stream);
int ssc_gather_rows(const float* src, int ld, const int64_t* backptr, int B, int rows_per_batch, int W, float* dst,
                    void* stream);
/* back-trace (cbs.py:252-277): preds (steps,B,SB) int64, backptrs (steps-1,B,SB) int64 -> out (B,SB,steps). */
int ssc_beam_backtrace(const int64_t* preds, const int64_t* backptrs, int steps, int B, int SB, int64_t* out,
                       void* stream);


/* ------------------------------------------------------------------------------------------------
 * Compiled finite-state machines for constrained beam search.
 * The reference keeps a machine as a dense adjacency tensor fsm[from, to, token] (uint8, (S,S,V):
 * updown-baseline/updown/utils/constraints.py:328-478) and its search scans every row's V log-probs and V mask
 * bytes once per TARGET state (cbs.py:157-250: `for i in range(num_fsm_states)` masked_fill + topk).
 * The machines the reference builds send almost every token of a from-state to ONE target set (the self-loop
 * of a main state, the reset state of a sub-state); only the word forms of the constraint words differ.
 * ssc_fsm_compile turns each (machine, from-state) into
 *     default target set (bit i = state i) | <= E exception tokens, ascending, each with its target set |
 *     an "is exception" bitmap over V | the P smallest non-exception tokens,
 * and ssc_beam_step_fsm makes ONE scan per row - the top per_node non-exception tokens - and then selects, per
 * target state, among those, the exception tokens and the -1e20 fill.  The result is the dense kernels' bit for
 * bit (same order: value descending, token ascending).  A from-state that is not "default + <= E exceptions" is
 * flagged and its rows take the dense per-target scans inside the same launch.  S <= 32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int M;   /* machines */
  int S;   /* states per machine (<= 32) */
  int V;   /* vocabulary */
  int E;   /* exception capacity per (machine, from-state) */
  int P;   /* fill tokens kept per (machine, from-state): >= the per_node of every later ssc_beam_step_fsm */
} ssc_fsm_dims;
size_t ssc_fsm_tables_bytes(const ssc_fsm_dims* d);
/* fsm (M,S,S,V) uint8 -> tables (device, ssc_fsm_tables_bytes) */
int ssc_fsm_compile(const uint8_t* fsm, const ssc_fsm_dims* d, void* tables, size_t tables_bytes, void* stream);

/* One descriptor for both search steps.  Batch entry b uses machine mach[b] (NULL: machine b), so the N_Z latent
 * samples of an image share its machine instead of carrying a copy each. */
typedef struct {
  const float* scores; int ld;   /* first: (B,V); step: (B*S*beam, V); log-probs, or un-normalised logits with raw_logits = 1 */
  int raw_logits;
  const uint8_t* fsm;            /* dense (M,S,S,V); NULL only for the trivial machine (S = 1, every transition allowed) */
  const void* tables;            /* from ssc_fsm_compile, or NULL: dense scans */
  ssc_fsm_dims dims;             /* of `tables` / `fsm` (M, S, V used when tables is NULL) */
  const int* mach;               /* (B) or NULL; every entry in [0, dims.M) - the kernels index the machines with it unchecked */
  int B, beam, per_node, end_index;
  const int64_t* last_pred;      /* step: (B, S*beam) previous predictions */
  const float* last_lp;          /* step: (B, S, beam) running log-probs */
  int64_t* pred; float* lp_out;  /* (B, S*beam), (B, S, beam) */
  int64_t* backptr;              /* step: (B, S*beam) */
  float* scratch_val; int64_t* scratch_idx;   /* step: B*S*S*beam*per_node each */
  int skip_dead;                 /* step: a row whose running log-prob is <= -1e19 (no finite beam: only -1e20 fills led to it) is
                                  * scored as if all its log-probs were 0 and its `scores` row is NOT read.  Every log-prob of the
                                  * search and every beam with a finite log-prob stay bit-identical (x + (-1e20) == -1e20 in fp32);
                                  * only the token ids of beams that are themselves <= -1e19 can differ.  Lets the caller skip the
                                  * decode step for such rows (ssc_decode_step_desc.live_*) */
  /* early stop without a host round trip (cbs.py:167 asks `(last_predictions == end).all()` before every step):
   * ctl = device int32[2 + 2*max_steps], zero-filled by the caller before the first step except ctl[0] = max_steps;
   * step t (1-based index of the column it writes, first = 0) counts the beams that have not ended, and the last workgroup of
   * the step to finish sets ctl[0] = min(ctl[0], t + 1) when there are none and writes t + 1 to host_flag[0]; it also notes its
   * own completion, host_flag[1] = t (host_flag: a device-visible pointer to TWO ints of pinned host memory, zeroed by the caller,
   * optional - ssc_decode_search bounds the host's run-ahead with [1]).  A step that finds ctl[0] <= t (the search had already stopped) emits END at +0
   * for every beam with the identity back-pointer, so that surplus steps queued by a host that polls *host_flag late change
   * nothing: columns [0, ctl[0]) of the back-trace ARE the reference's output. */
  int* ctl; int step_index; int max_steps; int* host_flag;
} ssc_beam_desc;
int ssc_beam_first_fsm(const ssc_beam_desc* d, void* stream);
int ssc_beam_step_fsm(const ssc_beam_desc* d, void* stream);
/* ssc_beam_step_fsm for the trivial machine (S = 1, fsm NULL) and per_node <= 2 from the RECORDS a vocabulary head launched with
 * ssc_gemm_desc.topk_part left (rows B*beam, ceil(V / 128) tiles of 6 floats each) instead of from the (rows, V) logits: the
 * row's log-sum-exp is combined from the tiles' partials, its best tokens from the tiles' best two.  Same selections as
 * ssc_beam_step_fsm on the logits unless two candidates collide after the subtraction of the log-sum-exp (which here is
 * summed in tile order); log-probs agree to rounding (~1e-6).  d->scores is not read. */
int ssc_beam_step_parts(const ssc_beam_desc* d, const float* parts, void* stream);
/* back-trace with the step count taken from ctl[0] on the device: out (B,SB,max_steps); columns >= ctl[0] are filled with end_index */
int ssc_beam_backtrace_ctl(const int64_t* preds, const int64_t* backptrs, const int* ctl, int max_steps, int B, int SB,
                           int end_index, int64_t* out, void* stream);
/* device-visible address of a pinned (hipHostMalloc / torch pin_memory) host word, for ssc_beam_desc.host_flag */
int ssc_host_device_ptr(void* host_ptr, void** device_ptr);

/* ------------------------------------------------------------------------------------------------
 * One diverse-decode call = ONE entry point: the whole constrained beam search over nimg images x n_samples latent
 * samples (batch entry b = (image, sample), rows (b, fsm state, beam)), every step launched from the library:
 * ConstrainedBeamSearch.search (updown-baseline/updown/modules/cbs.py:59-277) around the eval _decode_step
 * (var_updown/var_updown/models/updown_captioner.py:371-455), what var_updown/scripts/inference.py:117-189 runs per image
 * and sample.  Noise is handed in for all steps (the reference draws (rows, Z) per step, updown_cell.py:206), so the
 * random state a call consumes does not depend on where the search stops.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int nimg, R, n_samples;        /* image context (ssc_decode_prepare over nimg images of R regions); B = nimg * n_samples */
  int S, beam, per_node, max_steps, end_index;
  const float* feats;            /* (nimg, R, F) */
  const void* imgbuf;            /* from ssc_decode_prepare */
  const float* sentiment;        /* (B) or NULL */
  const float* eps0;             /* (B, Z): noise of the first step */
  const float* eps;              /* (max_steps - 1, B*S*beam, Z): noise of the later steps */
  const float* obj_atts;         /* cfg->kld_mode 2: (nimg, R, Z) per-region attribute means (ssc_decode_step_desc.obj_atts); else NULL */
  const uint8_t* fsm;            /* (M, S, S, V) dense machines; NULL with S = 1: the trivial machine */
  const void* tables;            /* ssc_fsm_compile of `fsm`, or NULL: dense scans */
  ssc_fsm_dims dims;             /* of `tables` */
  const int* mach;               /* (B) machine of every batch entry (each in [0, M): not checked on the device), or NULL: machine b */
  int skip_dead;                 /* ssc_beam_desc.skip_dead + ssc_decode_step_desc.row_lp: rows without a finite beam (needs `tables`) and rows
                                  * whose beam has ended (any machine; with the trivial machine these are the only ones) are neither
                                  * stepped nor scored from logits */
  int early_stop;                /* cbs.py:167 */
  int64_t* predictions;          /* out (B, S*beam, max_steps): columns [0, ctl[0]) are the search's; the rest holds end_index */
  float* log_probs;              /* out (B, S, beam) */
  int* ctl;                      /* device int32[2 + 2*max_steps], initialised by the call; ctl[0] = number of columns (steps) */
  int* host_flag;                /* optional: device-visible address of TWO pinned host ints (ssc_host_device_ptr), zeroed by the caller:
                                  * [0] = the stop flag, [1] = the last step the device has completed */
  const int* host_flag_host;     /* the same words' host address: read between steps - the host stops queueing once [0] is set and
                                  * stays at most two steps ahead of [1] (plain memory reads, no event, no synchronisation call) */
} ssc_search_desc;
size_t ssc_decode_search_workspace_bytes(const ssc_model_cfg* cfg, const ssc_search_desc* d);
int ssc_decode_search(const ssc_model_cfg* cfg, const ssc_params* p, const ssc_search_desc* d, void* workspace,
                      size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSC_H */

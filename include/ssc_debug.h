/*
 * ssc_debug.h - diagnostics, profiling and tuning switches of libssc_hip.so.  NOT part of the product ABI (ssc.h):
 * nothing here is needed to run the hot path, and everything here is process-global and not thread-safe.
 * Used by bench.py (roofline leg), tools/ and the kernel-form tests.
 */
#ifndef SSC_DEBUG_H
#define SSC_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* In-situ GEMM profiling: while enabled every GEMM launch is bracketed by a hipEvent pair on its stream.
 * ssc_prof_collect synchronises the device and writes up to max_records x 8 floats
 * {kind (0 NT, 1 NN, 3 TN), M, N, sum K, splits, milliseconds, algorithmic bytes, flops}; returns the record count.  A grouped
 * launch is ONE record: its bytes (every operand and result of every member once) and flops (2 M N K per member) are exact,
 * its N / K are nominal (widest member, summed K). */
int ssc_prof_enable(int on);
int ssc_prof_collect(float* out, int max_records);

/* Time-loop timing of ssc_train_fwd / ssc_train_bwd: while enabled a hipEvent pair brackets the T-step loop itself
 * (precompute, hoisted products, vocabulary head and the weight-gradient products are outside).  ssc_prof_loop_ms
 * synchronises the events of the LAST forward / backward call and returns the loop times in milliseconds
 * (negative when that call has not run since the switch was turned on). */
int ssc_prof_loop_enable(int on);
int ssc_prof_loop_ms(float* fwd_loop_ms, float* bwd_loop_ms);

/* Named switches between kernel forms that compute the same product (A/B measurements, kernel-form tests).
 * Keys (default; environment variable read once at load):
 *   "large_form"  large products: 0 64-wide kernels | 3 4-wave 128x128 3xBF16 kernel | 2 wave-specialised 128x128 |
 *                 1 by grid size: wave-specialised from 768 workgroups on (default)               (SSC_X3B)
 *   "x3w_skinny"  minibatch products on the wave-specialised 64x256 kernel: 0 | 1 NT+NN (default) | 2 NN only (SSC_X3W_SKINNY)
 *   "x3w_min_n"   ... from this output width on (1024)                                             (SSC_X3W_MIN_N)
 *   "x3_wide" (0), "x3_nbuf" (1), "x3_pf" (2)   forms of the 64-wide 3xBF16 kernel
 *   "wide_min_n"  exact-fp32 kernels: 64x128 tile for M <= 64 from this width on (1024)
 *   "gemm_group"  grouped launches of independent minibatch products (1)                          (SSC_GEMM_GROUP)
 *   "dw_group"    grouped launches of the weight-gradient products (1)                            (SSC_DW_GROUP)
 *   "x3w_npw" (8), "x3w_big_npw" (8)   producer waves of the wave-specialised 64x256 / 128x128 kernels: 4 | 8
 *                                                                                    (SSC_X3W_NPW, SSC_X3W_BIG_NPW)
 *   "x3w_pf"      k-steps in flight in the producers' registers of the 64x256 kernels: 2 (default) | 3   (SSC_X3W_PF)
 *   "store_wt"    write-through (sc1) output stores of the wave-specialised kernels (1)            (SSC_STORE_WT)
 *   "tile_gm"     tile rows per group of the XCD-aware tile order (8; 0 = row-major)               (SSC_TILE_GM)
 *   "big_min_m"   rows from which a product with N >= 512 takes 128x128 tiles (65); below 512 rows always in the
 *                 wave-specialised form                                                            (SSC_BIG_MIN_M)
 *   decode (ssc_decode_step and the beam kernels; every form gives the same captions):
 *   "dec_att_table"  attended-feature term of the decoder gates from the per-image table (1)      (SSC_DEC_ATT_TABLE)
 *   "dec_dedup"      products fed only by the parent's states on the distinct parents (1)         (SSC_DEC_DEDUP)
 *   "dec_ungathered" ssc_decode_ungathered_ok() may say yes: states read through the parent lists (1)   (SSC_DEC_UNGATHERED)
 *   "img_mfma"       table contraction of ssc_lstm_fwd_img on the fp32 matrix cores (1 | 0 = VALU form)  (SSC_IMG_MFMA)
 *   "beam_reg"       beam selection with the vocabulary row in registers for V <= 10240 (1)       (SSC_BEAM_REG)
 * The environment variables are honoured only with SSC_DEBUG=1.  Returns SSC_EINVAL for an unknown key. */
int ssc_debug_set(const char* key, int value);
int ssc_debug_get(const char* key, int* value);

/* resident workgroups per CU the runtime reports for the GEMM kernels (tools/occ.py) */
int ssc_debug_gemm_occupancy(int* out4);
int ssc_debug_gemm_occupancy_x3b(int* out4);

#ifdef __cplusplus
}
#endif
#endif

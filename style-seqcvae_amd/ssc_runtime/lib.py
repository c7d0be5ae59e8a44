"""ctypes binding of libssc_hip.so - mirrors include/ssc.h one to one."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product always loads the in-tree library.  Tools that A/B two builds opt in with SSC_DEBUG=1 (then SSC_LIB_PATH is honoured);
# without it a stray SSC_LIB_PATH is ignored.
DEBUG_ENV = os.environ.get("SSC_DEBUG", "") == "1"
LIB_PATH = (DEBUG_ENV and os.environ.get("SSC_LIB_PATH")) or os.path.join(os.path.dirname(_HERE), "libssc_hip.so")
SSC_MAX_SEG = 6

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)
vp = C.c_void_p


class SscError(RuntimeError):
    CODES = {-1: "SSC_EINVAL", -2: "SSC_EALIGN", -3: "SSC_EHIP", -4: "SSC_EWORKSPACE"}

    def __init__(self, fn, rc, hip=0):
        super().__init__(f"{fn} failed: {self.CODES.get(rc, rc)}" + (f" (hipError {hip})" if rc == -3 else ""))
        self.rc = rc


class GemmSeg(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("lda", C.c_int), ("ldb", C.c_int), ("K", C.c_int),
                ("A16", vp), ("B16", vp), ("lda16", C.c_int), ("ldb16", C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("seg", GemmSeg * SSC_MAX_SEG), ("nseg", C.c_int), ("M", C.c_int), ("N", C.c_int), ("a_kc", C.c_int),
                ("b_kc", C.c_int), ("C", vp), ("ldc", C.c_int), ("bias", vp), ("accumulate", C.c_int),
                ("splits", C.c_int), ("workspace", vp), ("workspace_floats", C.c_size_t), ("m_count", vp), ("a_rows", vp),
                ("c_rows", vp), ("k_count", vp), ("ka_rows", vp), ("kb_rows", vp), ("a_scale", vp), ("b_scale", vp), ("topk_part", vp)]


class LstmFwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("slabs", vp), ("nslab", C.c_int), ("slab_stride", C.c_size_t),
                ("add0", vp), ("ld_add0", C.c_int), ("add1", vp), ("ld_add1", C.c_int), ("rows_per_add1", C.c_int),
                ("b_ih", vp), ("b_hh", vp), ("sent", vp), ("wcol", vp), ("ldwcol", C.c_int), ("c_prev", vp),
                ("ld_cprev", C.c_int), ("gates_out", vp), ("c_out", vp), ("ld_cout", C.c_int), ("h_out", vp),
                ("ld_hout", C.c_int), ("add0_rows", vp), ("slab_rows", vp), ("slabs2", vp), ("nslab2", C.c_int),
                ("slab2_stride", C.c_size_t), ("slab2_rows", vp), ("c_prev_rows", vp), ("rows", vp), ("row_count", vp),
                ("h_planes", vp), ("ld_hplanes", C.c_int), ("planes_scale", vp)]


class LstmBwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("dh", vp), ("ld_dh", C.c_int), ("dh2", vp), ("ld_dh2", C.c_int),
                ("dc_in", vp), ("ld_dcin", C.c_int), ("gates", vp), ("c_prev", vp), ("ld_cprev", C.c_int),
                ("c_new", vp), ("ld_cnew", C.c_int), ("dG", vp), ("dc_prev", vp), ("ld_dcprev", C.c_int), ("dgsum", vp),
                ("slabsA", vp), ("nA", C.c_int), ("strideA", C.c_size_t), ("slabsB", vp), ("nB", C.c_int),
                ("strideB", C.c_size_t)]


class LatentFwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("Z", C.c_int), ("mulv", vp), ("ldmulv", C.c_int), ("nslab", C.c_int),
                ("slab_stride", C.c_size_t), ("bmu", vp), ("blv", vp), ("eps", vp), ("ldeps", C.c_int),
                ("kld_mode", C.c_int), ("sent", vp), ("pm_scale", C.c_float), ("prior_var", C.c_float), ("w", vp),
                ("mu", vp), ("lv", vp), ("z", vp), ("ldz", C.c_int), ("kld_acc", vp), ("pm", vp), ("ldpm", C.c_int)]


class LatentBwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("Z", C.c_int), ("dz", vp), ("lddz", C.c_int), ("eps", vp), ("ldeps", C.c_int),
                ("mu", vp), ("lv", vp), ("ldz", C.c_int), ("kld_mode", C.c_int), ("sent", vp), ("pm_scale", C.c_float),
                ("prior_var", C.c_float), ("w", vp), ("gk", vp), ("dmulv", vp), ("lddmulv", C.c_int),
                ("nslab", C.c_int), ("slab_stride", C.c_size_t), ("pm", vp), ("ldpm", C.c_int), ("dpm", vp), ("lddpm", C.c_int)]


class ModelCfg(C.Structure):
    _fields_ = [("V", C.c_int), ("E", C.c_int), ("H", C.c_int), ("A", C.c_int), ("F", C.c_int), ("Z", C.c_int),
                ("S", C.c_int), ("tied", C.c_int), ("kld_mode", C.c_int), ("pm_scale", C.c_float),
                ("prior_var", C.c_float), ("pad", C.c_int), ("boundary", C.c_int), ("gemm_mode", C.c_int)]


# (field, has_ld) in the exact order of ssc_params
PARAM_FIELDS = [("emb", True), ("att_w_ih", True), ("att_w_hh", True), ("att_b_ih", False), ("att_b_hh", False),
                ("wq", True), ("wv", True), ("wa", False), ("enc_w_ih", True), ("enc_w_hh", True), ("enc_b_ih", False),
                ("enc_b_hh", False), ("dec_w_ih", True), ("dec_w_hh", True), ("dec_b_ih", False), ("dec_b_hh", False),
                ("fc_mean_w", True), ("fc_mean_b", False), ("fc_lv_w", True), ("fc_lv_b", False), ("out_w", True),
                ("out_b", False), ("proj_w", True), ("proj_b", False)]


def _param_fields():
    f = []
    for name, has_ld in PARAM_FIELDS:
        f.append((name, vp))
        if has_ld:
            f.append(("ld_" + name, C.c_int))
    return f


class Params(C.Structure):
    _fields_ = _param_fields()


class Batch(C.Structure):
    _fields_ = [("B", C.c_int), ("R", C.c_int), ("L", C.c_int), ("feats", vp), ("caps", vp), ("sentiment", vp),
                ("eps", vp), ("obj_atts", vp)]


SSC_XGMI_MAX_RANKS = 8


class XgmiComm(C.Structure):
    _fields_ = [("world", C.c_int), ("rank", C.c_int), ("buf", vp * SSC_XGMI_MAX_RANKS), ("flags", vp * SSC_XGMI_MAX_RANKS)]


class DecodeStepDesc(C.Structure):
    _fields_ = [("G", C.c_int), ("R", C.c_int), ("rows_per_image", C.c_int), ("feats", vp), ("imgbuf", vp),
                ("tokens", vp), ("sentiment", vp), ("eps", vp), ("h1", vp), ("c1", vp), ("hd", vp), ("cd", vp),
                ("h1_out", vp), ("c1_out", vp), ("hd_out", vp), ("cd_out", vp), ("alpha", vp), ("log_probs", vp),
                ("raw_logits", C.c_int), ("emb_override", C.c_int), ("parent", vp), ("group", C.c_int), ("att_table", C.c_int),
                ("ungathered", C.c_int), ("row_lp", vp), ("end_index", C.c_int), ("obj_atts", vp), ("prior_mean_out", vp), ("prior_mean", vp), ("prior_var", vp), ("topk_part", vp),
                ("h1_planes", vp), ("hd_planes", vp), ("h1_planes_out", vp), ("hd_planes_out", vp)]


class FsmDims(C.Structure):
    _fields_ = [("M", C.c_int), ("S", C.c_int), ("V", C.c_int), ("E", C.c_int), ("P", C.c_int)]


class BeamDesc(C.Structure):
    _fields_ = [("scores", vp), ("ld", C.c_int), ("raw_logits", C.c_int), ("fsm", vp), ("tables", vp), ("dims", FsmDims),
                ("mach", vp), ("B", C.c_int), ("beam", C.c_int), ("per_node", C.c_int), ("end_index", C.c_int),
                ("last_pred", vp), ("last_lp", vp), ("pred", vp), ("lp_out", vp), ("backptr", vp), ("scratch_val", vp),
                ("scratch_idx", vp), ("skip_dead", C.c_int), ("ctl", vp), ("step_index", C.c_int), ("max_steps", C.c_int),
                ("host_flag", vp)]


class SearchDesc(C.Structure):
    _fields_ = [("nimg", C.c_int), ("R", C.c_int), ("n_samples", C.c_int), ("S", C.c_int), ("beam", C.c_int), ("per_node", C.c_int),
                ("max_steps", C.c_int), ("end_index", C.c_int), ("feats", vp), ("imgbuf", vp), ("sentiment", vp), ("eps0", vp),
                ("eps", vp), ("obj_atts", vp), ("fsm", vp), ("tables", vp), ("dims", FsmDims), ("mach", vp), ("skip_dead", C.c_int),
                ("early_stop", C.c_int), ("predictions", vp), ("log_probs", vp), ("ctl", vp), ("host_flag", vp),
                ("host_flag_host", vp)]


# name -> (restype, argtypes).  Every symbol include/ssc.h declares is listed; tests check they all resolve.
_i, _f, _sz = C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "ssc_version": (_i, []),
    "ssc_last_hip_error": (_i, []),
    "ssc_arch": (C.c_char_p, []),
    "ssc_gemm": (_i, [C.POINTER(GemmDesc), vp]),
    "ssc_gemm_auto_splits": (_i, [_i, _i, _i]),
    "ssc_pow2_scale": (_i, [vp, _sz, _i, _sz, _i, vp, _i, vp, vp]),
    "ssc_split_f16": (_i, [vp, _i, _i, _i, vp, vp, _i, vp, vp, vp]),
    "ssc_decode_planes_ld": (_i, [vp]),
    "ssc_set_gemm_mode": (_i, [_i]),
    "ssc_feat_prep": (_i, [vp, _i, _i, _i, vp, vp, vp]),
    "ssc_prep_tokens": (_i, [vp, _i, _i, _i, _i, vp, vp, vp, vp]),
    "ssc_embed_gather": (_i, [vp, _i, vp, _i, _i, vp, _i, vp]),
    "ssc_embed_scatter_add": (_i, [vp, _i, vp, _i, _i, vp, _i, _i, vp]),
    "ssc_lstm_fwd": (_i, [C.POINTER(LstmFwdDesc), vp]),
    "ssc_lstm_fwd_z": (_i, [C.POINTER(LstmFwdDesc), vp, _i, vp, _i, _i, vp]),
    "ssc_lstm_fwd_p": (_i, [C.POINTER(LstmFwdDesc), vp, _i, _i, vp, vp]),
    "ssc_lstm_fwd_img": (_i, [C.POINTER(LstmFwdDesc), vp, _i, vp, _i, _i, vp]),
    "ssc_lstm_bwd_x": (_i, [C.POINTER(LstmBwdDesc), vp, _i, vp, _i, _i, vp]),
    "ssc_lstm_bwd": (_i, [C.POINTER(LstmBwdDesc), vp]),
    "ssc_attn_logits": (_i, [vp, _i, vp, vp, _i, _i, _i, _i, vp, vp]),
    "ssc_attn_fwd": (_i, [vp, _i, vp, vp, vp, vp, _i, _i, _i, _i, _i, vp, vp, vp, _i, vp]),
    "ssc_attn_weights": (_i, [vp, _i, vp, vp, vp, _i, _i, _i, _i, vp, vp, vp]),
    "ssc_attn_pool": (_i, [vp, vp, _i, _i, _i, _i, vp, _i, vp]),
    "ssc_attn_fwd_pool": (_i, [vp, _i, vp, vp, vp, vp, _i, _i, _i, _i, _i, vp, vp, vp, _i, vp, _i, vp, _i, vp]),
    "ssc_attn_bwd_pool": (_i, [vp, _i, vp, _i, vp, vp, vp, vp, _i, _i, _i, _i, vp, _i, vp, vp, vp, vp, _i, vp, _i, _i, vp, _i, vp]),
    "ssc_attn_bwd": (_i, [vp, _i, vp, _i, vp, vp, vp, vp, _i, _i, _i, _i, vp, _i, vp, vp, vp, vp]),
    "ssc_latent_fwd": (_i, [C.POINTER(LatentFwdDesc), vp]),
    "ssc_latent_prior_sample": (_i, [vp, _i, vp, _f, _f, _i, _i, vp, _i, vp]),
    "ssc_latent_prior_sample_pm": (_i, [vp, _i, vp, _i, vp, _i, vp, _f, _f, _i, _i, vp, _i, vp]),
    "ssc_latent_bwd": (_i, [C.POINTER(LatentBwdDesc), vp]),
    "ssc_ce_fwd": (_i, [vp, _i, vp, vp, vp, _i, _i, _i, vp, vp, vp]),
    "ssc_ce_bwd": (_i, [vp, _i, vp, vp, vp, vp, vp, _i, _i, _i, vp]),
    "ssc_log_softmax": (_i, [vp, _i, _i, _i, vp, _i, vp]),
    "ssc_colsum": (_i, [vp, _i, _i, _i, vp, vp, _i, _i, vp]),
    "ssc_colsum2": (_i, [vp, _i, _i, _i, vp, vp, _i, vp, _i, vp, vp]),
    "ssc_copy_strided": (_i, [vp, _sz, _i, vp, vp]),
    "ssc_bias_tanh": (_i, [vp, _i, _i, _i, vp, vp]),
    "ssc_tanh_bwd": (_i, [vp, _i, vp, _i, _i, _i, vp]),
    "ssc_fill": (_i, [vp, _sz, _f, vp]),
    "ssc_sq_norm": (_i, [vp, _sz, vp, vp, vp]),
    "ssc_sgd_step": (_i, [vp, vp, vp, _sz, vp, _f, _f, _f, _f, _f, _i, vp]),
    "ssc_train_workspace_bytes": (_sz, [C.POINTER(ModelCfg), _i, _i, _i]),
    "ssc_train_fwd": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), C.POINTER(Batch), vp, _sz, vp, vp, vp]),
    "ssc_train_bwd": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), C.POINTER(Batch), vp, _sz, vp, vp, C.POINTER(Params), vp]),
    "ssc_train_bwd_phases": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), C.POINTER(Batch), vp, _sz, vp, vp, C.POINTER(Params),
                                  C.c_uint, vp]),
    "ssc_train_workspace_view": (vp, [C.POINTER(ModelCfg), _i, _i, _i, vp, _i, C.POINTER(C.c_int)]),
    "ssc_xgmi_enable_peer": (_i, [_i]),
    "ssc_xgmi_ipc_export": (_i, [vp, vp, C.POINTER(C.c_size_t)]),
    "ssc_xgmi_ipc_open": (_i, [vp, C.POINTER(vp)]),
    "ssc_xgmi_ipc_close": (_i, [vp]),
    "ssc_xgmi_peek": (_i, [vp, vp, _sz]),
    "ssc_xgmi_allreduce": (_i, [C.POINTER(XgmiComm), _sz, _sz, C.c_uint, C.c_uint, vp, vp]),
    "ssc_decode_image_bytes": (_sz, [C.POINTER(ModelCfg), _i, _i]),
    "ssc_decode_prepare": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), vp, _i, _i, vp, _sz, vp]),
    "ssc_decode_prepare_from": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), vp, _i, _i, vp, _sz, vp, _i, _i, vp]),
    "ssc_decode_step_workspace_bytes": (_sz, [C.POINTER(ModelCfg), _i, _i]),
    "ssc_decode_step": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), C.POINTER(DecodeStepDesc), vp, _sz, vp]),
    "ssc_decode_ungathered_ok": (_i, [C.POINTER(ModelCfg), _i, _i, _i, _i]),
    "ssc_beam_first": (_i, [vp, _i, vp, _i, _i, _i, _i, vp, vp, vp]),
    "ssc_beam_step": (_i, [vp, _i, vp, vp, vp, _i, _i, _i, _i, _i, _i, vp, vp, vp, vp, vp, vp]),
    "ssc_beam_first_logits": (_i, [vp, _i, vp, _i, _i, _i, _i, vp, vp, vp]),
    "ssc_beam_step_logits": (_i, [vp, _i, vp, vp, vp, _i, _i, _i, _i, _i, _i, vp, vp, vp, vp, vp, vp]),
    "ssc_gather_rows": (_i, [vp, _i, vp, _i, _i, _i, vp, vp]),
    "ssc_beam_backtrace": (_i, [vp, vp, _i, _i, _i, vp, vp]),
    "ssc_fsm_tables_bytes": (_sz, [C.POINTER(FsmDims)]),
    "ssc_fsm_compile": (_i, [vp, C.POINTER(FsmDims), vp, _sz, vp]),
    "ssc_beam_first_fsm": (_i, [C.POINTER(BeamDesc), vp]),
    "ssc_beam_step_fsm": (_i, [C.POINTER(BeamDesc), vp]),
    "ssc_beam_step_parts": (_i, [C.POINTER(BeamDesc), vp, vp]),
    "ssc_beam_backtrace_ctl": (_i, [vp, vp, vp, _i, _i, _i, _i, vp, vp]),
    "ssc_host_device_ptr": (_i, [vp, C.POINTER(vp)]),
    "ssc_decode_search_workspace_bytes": (_sz, [C.POINTER(ModelCfg), C.POINTER(SearchDesc)]),
    "ssc_decode_search": (_i, [C.POINTER(ModelCfg), C.POINTER(Params), C.POINTER(SearchDesc), vp, _sz, vp]),
}

# include/ssc_debug.h (diagnostics / profiling / tuning switches: not part of the product ABI)
DEBUG_SYMBOLS = {
    "ssc_prof_enable": (_i, [_i]),
    "ssc_prof_collect": (_i, [vp, _i]),
    "ssc_prof_loop_enable": (_i, [_i]),
    "ssc_prof_loop_ms": (_i, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ssc_debug_set": (_i, [C.c_char_p, _i]),
    "ssc_debug_get": (_i, [C.c_char_p, C.POINTER(C.c_int)]),
    "ssc_debug_gemm_occupancy": (_i, [vp]),
    "ssc_debug_gemm_occupancy_x3b": (_i, [vp]),
}

_lib = None


class _Lib:
    def __init__(self, cdll):
        self._cdll = cdll
        for name, (res, args) in list(SYMBOLS.items()) + list(DEBUG_SYMBOLS.items()):
            fn = getattr(cdll, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_raw_" + name, fn)

    def __getattr__(self, name):
        raw = self.__dict__.get("_raw_" + name)
        if raw is None:
            raise AttributeError(name)
        if raw.restype is not C.c_int or name in ("ssc_version", "ssc_last_hip_error", "ssc_gemm_auto_splits", "ssc_prof_collect", "ssc_set_gemm_mode"):
            return raw

        def checked(*a):
            rc = raw(*a)
            if rc != 0:
                raise SscError(name, rc, self._raw_ssc_last_hip_error())
            return rc

        return checked


def load():
    """Load libssc_hip.so.  Raises (never falls back) when the extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python style-seqcvae_amd/build.py` "
                              "(the HIP extension is mandatory; there is no CPU fallback)")
        # PyTorch-ROCm ships its own HIP runtime.  It must be in the process BEFORE this library is loaded, so that both
        # resolve to the same runtime instance; loaded the other way round the library binds the system runtime and every
        # launch on a torch stream fails with hipErrorNoDevice (100).
        import torch  # noqa: F401
        _lib = _Lib(C.CDLL(LIB_PATH))
    return _lib


def ptr(t):
    """Device (or host) pointer of a torch tensor as c_void_p (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

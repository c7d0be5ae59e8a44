"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import ctypes as C

import torch

from ssc_runtime import lib as L   # (no oracle import here: tools/ reuse the GEMM helper and must stay oracle-free)
from ssc_runtime.engine import ModelDims, TrainEngine


def dims_from_cfg(cfg: "oracle.OracleConfig") -> ModelDims:
    sv1 = cfg.sentiment_vae == 1 and not cfg.simple_vae
    return ModelDims(V=cfg.vocab_size, E=cfg.embedding_size, H=cfg.hidden_size, A=cfg.attention_projection_size,
                     F=cfg.image_feature_size, Z=cfg.z_space, S=cfg.senti_cols, tied=cfg.tied,
                     kld_mode=0 if cfg.sentiment_vae == 0 else (2 if cfg.sentiment_vae == 2 and not cfg.simple_vae else 1),
                     pm_scale=cfg.senti_prior_multip if sv1 else 0.0, prior_var=cfg.prior_std ** 2,
                     pad=cfg.pad_index, boundary=cfg.boundary_index)


def engine_from(cfg, params) -> TrainEngine:
    eng = TrainEngine(dims_from_cfg(cfg), "cuda")
    eng.load_state_dict(params)
    return eng


def dev(t):
    return t.cuda().contiguous()


def gemm(segs, M, N, a_kc, b_kc, Cout, bias=None, accumulate=0, splits=0, ws=None, compact=None, check=True):
    """segs: list of (A, lda, B, ldb, K) with torch tensors (views allowed; pointer = data_ptr())."""
    lib = L.load()
    d = L.GemmDesc()
    d.nseg = len(segs)
    for i, (A, lda, B, ldb, K) in enumerate(segs):
        d.seg[i].A = A.data_ptr()
        d.seg[i].B = B.data_ptr()
        d.seg[i].lda, d.seg[i].ldb, d.seg[i].K = lda, ldb, K
    d.M, d.N, d.a_kc, d.b_kc = M, N, int(a_kc), int(b_kc)
    d.C = Cout.data_ptr()
    d.ldc = Cout.stride(0)
    d.bias = bias.data_ptr() if bias is not None else None
    d.accumulate = accumulate
    d.splits = splits
    if ws is not None:
        d.workspace = ws.data_ptr()
        d.workspace_floats = ws.numel()
    for k, v in (compact or {}).items():   # m_count / a_rows / c_rows / k_count / ka_rows / kb_rows: int32 device tensors
        setattr(d, k, v.data_ptr())
    rc = lib._raw_ssc_gemm(C.byref(d), L.stream_ptr()) if not check else lib.ssc_gemm(C.byref(d), L.stream_ptr())
    return rc


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()

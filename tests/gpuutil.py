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


def split_f16(x, K=None, scale=None, rows=None):
    """ssc_split_f16 of x[:, :K] -> int32 tensor (rows, roundup(K, 32)) in the plane layout of ssc_gemm_seg.A16 / B16."""
    lib = L.load()
    K = x.size(1) if K is None else K
    kp = (K + 31) // 32 * 32
    out = torch.full((x.size(0), kp), 0x7E007E00, dtype=torch.int32, device=x.device)   # (fp16 NaNs: an unwritten word shows)
    rl, rc = (None, None) if rows is None else (L.ptr(rows[0]), L.ptr(rows[1]))
    lib.ssc_split_f16(L.ptr(x), x.size(0), K, x.stride(0), L.ptr(scale) if scale is not None else None, L.ptr(out), kp, rl, rc,
                      L.stream_ptr())
    return out


def gemm(segs, M, N, a_kc, b_kc, Cout, bias=None, accumulate=0, splits=0, ws=None, compact=None, check=True, planes=None):
    """segs: list of (A, lda, B, ldb, K) with torch tensors (views allowed; pointer = data_ptr()); planes: per segment
    (A16 or None, B16 or None) int32 tensors from split_f16."""
    lib = L.load()
    d = L.GemmDesc()
    d.nseg = len(segs)
    for i, (A, lda, B, ldb, K) in enumerate(segs):
        d.seg[i].A = A.data_ptr()
        d.seg[i].B = B.data_ptr()
        d.seg[i].lda, d.seg[i].ldb, d.seg[i].K = lda, ldb, K
        if planes is not None:
            a16, b16 = planes[i]
            if a16 is not None:
                d.seg[i].A16, d.seg[i].lda16 = a16.data_ptr(), a16.stride(0)
            if b16 is not None:
                d.seg[i].B16, d.seg[i].ldb16 = b16.data_ptr(), b16.stride(0)
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

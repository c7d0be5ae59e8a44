"""One UpDownCell training-mode step composed from the op-level C ABI (no autograd): the stand-alone form of
var_updown/var_updown/modules/updown_cell.py:86-231 with training=True.  The differentiable training path is the fused
sequence kernel (ssc_train_fwd / ssc_train_bwd); this exists so that UpDownCell.forward / UpDownCaptioner._decode_step
keep the reference's call signature in training mode too (inspection, debugging, step-wise callers)."""
import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L
from .engine import P_ATT, P_BUTD, P_CELL, P_DEC, P_ENC, ModelDims


def _gemm(lib, segs, M, N, out, bias=None, ws=None):
    d = L.GemmDesc()
    d.nseg = len(segs)
    for i, (A, lda, B, ldb, K) in enumerate(segs):
        d.seg[i].A, d.seg[i].B, d.seg[i].lda, d.seg[i].ldb, d.seg[i].K = A, B, lda, ldb, K
    d.M, d.N, d.a_kc, d.b_kc = M, N, 1, 1
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.bias = bias.data_ptr() if bias is not None else None
    if ws is not None:
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    lib.ssc_gemm(C.byref(d), L.stream_ptr())


def _lstm(lib, pre, b_ih, b_hh, c_prev, sent=None, wcol=None, ldw=1):
    G, H4 = pre.shape
    H = H4 // 4
    h, c = torch.empty(G, H, device=pre.device), torch.empty(G, H, device=pre.device)
    d = L.LstmFwdDesc()
    d.B, d.H = G, H
    d.slabs, d.nslab, d.slab_stride = pre.data_ptr(), 1, 0
    d.b_ih, d.b_hh = b_ih.data_ptr(), b_hh.data_ptr()
    if sent is not None:
        d.sent, d.wcol, d.ldwcol = sent.data_ptr(), wcol, ldw
    d.c_prev, d.ld_cprev = c_prev.data_ptr(), H
    d.c_out, d.ld_cout, d.h_out, d.ld_hout = c.data_ptr(), H, h.data_ptr(), H
    lib.ssc_lstm_fwd(C.byref(d), L.stream_ptr())
    return h, c


def cell_train_step(dims: ModelDims, P: Dict[str, torch.Tensor], feats: torch.Tensor, emb: torch.Tensor,
                    states: Optional[Dict[str, torch.Tensor]], sentiment: Optional[torch.Tensor], eps: torch.Tensor,
                    obj_atts: Optional[torch.Tensor] = None, training: bool = True, prior_mean: Optional[torch.Tensor] = None,
                    prior_var: Optional[torch.Tensor] = None):
    """One UpDownCell step through the op-level C ABI -> (h_decoder, states, mean, log_var, alpha, pooled) with pooled = the
    attention-pooled attribute means (G,Z) when obj_atts is given (SENTIMENT_VAE = 2), else None.
    feats (G,R,F), emb (G,E), sentiment (G,) or None, eps (G,Z).
    dims.S selects the conditioning of the language LSTMs (updown_cell.py:47-81): 0 none, 1 the sentiment column,
    > 1 (SENTIMENT_VAE = 2: 150) the attention-pooled attribute means c = sum_r alpha_r obj_atts_r (updown_cell.py:160-163),
    which then is also the prior mean.  training=False: no encoder LSTM, z = eps sqrt(prior_var) + prior_mean (:200-208)."""
    lib = L.load()
    dev = feats.device
    G, R, F = feats.shape
    E, H, A, Z, S = dims.E, dims.H, dims.A, dims.Z, dims.S
    f32 = dict(dtype=torch.float32, device=dev)
    feats = feats.contiguous().float()
    emb = emb.contiguous().float()
    if states is None:
        states = {k: torch.zeros(G, H, **f32) for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")}
    st = {k: v.contiguous().float() for k, v in states.items()}
    ws = torch.empty(40 * G * 4 * H + 64, **f32)
    mask, avg = torch.empty(G, R, **f32), torch.empty(G, F, **f32)
    lib.ssc_feat_prep(L.ptr(feats), G, R, F, L.ptr(mask), L.ptr(avg), L.stream_ptr())
    w_att, w_hh = P[P_ATT + "weight_ih"], P[P_ATT + "weight_hh"]
    ld = w_att.stride(0)
    base = w_att.data_ptr()
    pre = torch.empty(G, 4 * H, **f32)
    _gemm(lib, [(emb.data_ptr(), E, base, ld, E), (avg.data_ptr(), F, base + 4 * E, ld, F),
                (st["h1"].data_ptr(), H, base + 4 * (E + F), ld, H), (st["h_decoder"].data_ptr(), H, base + 4 * (E + F + H), ld, H),
                (st["h1"].data_ptr(), H, w_hh.data_ptr(), w_hh.stride(0), H)], G, 4 * H, pre, ws=ws)
    h1, c1 = _lstm(lib, pre, P[P_ATT + "bias_ih"], P[P_ATT + "bias_hh"], st["c1"])
    wq, wv, wa = (P[P_BUTD + n + ".weight"] for n in ("_query_vector_projection_layer", "_image_features_projection_layer",
                                                      "_attention_layer"))
    q, pv = torch.empty(G, A, **f32), torch.empty(G * R, A, **f32)
    _gemm(lib, [(h1.data_ptr(), H, wq.data_ptr(), wq.stride(0), H)], G, A, q, ws=ws)
    _gemm(lib, [(feats.data_ptr(), F, wv.data_ptr(), wv.stride(0), F)], G * R, A, pv, ws=ws)
    logits, alpha, att = torch.empty(G, R, **f32), torch.empty(G, R, **f32), torch.empty(G, F, **f32)
    lib.ssc_attn_fwd(L.ptr(q), A, L.ptr(pv), L.ptr(wa), L.ptr(mask), L.ptr(feats), G, R, A, F, 1, L.ptr(logits), L.ptr(alpha),
                     L.ptr(att), F, L.stream_ptr())
    sent = sentiment.reshape(G).contiguous().float() if (sentiment is not None and (S == 1 or dims.pm_scale != 0.0)) else None
    hd_prev = st["h_decoder"]
    cond = None      # (G, S) conditioning block of the language LSTMs when S > 1
    pooled = None    # (G, Z) attention-pooled attribute means: the prior mean of SENTIMENT_VAE = 2
    if S > 1 and obj_atts is None:
        raise ValueError("SENTIMENT_VAE = 2 needs obj_atts (G, R, %d): per-region attribute means" % S)
    if obj_atts is not None:
        oa = obj_atts.to(dev, torch.float32).contiguous()
        assert tuple(oa.shape) == (G, R, Z), (oa.shape, (G, R, Z))
        pooled = torch.empty(G, Z, **f32)
        lib.ssc_attn_pool(L.ptr(alpha), L.ptr(oa), G, R, Z, 1, L.ptr(pooled), Z, L.stream_ptr())
        if S > 1:        # LATENT_EMBEDDING "glove": the whole vector conditions the language LSTMs (S = Z = 150)
            assert S == Z
            cond = pooled
        elif S == 1:     # "senti_word_net": its first entry does (updown_cell.py:171-172)
            sent = pooled[:, 0].contiguous()
    csegs_e = csegs_d = []
    w_e, w_ehh = P[P_ENC + "weight_ih"], P[P_ENC + "weight_hh"]
    lde, be = w_e.stride(0), w_e.data_ptr()
    if cond is not None:
        csegs_e = [(cond.data_ptr(), S, be + 4 * (F + 2 * H), lde, S)]
    eps = eps.to(dev, torch.float32).contiguous()
    if not training:
        # eval (updown_cell.py:200-208): the encoder LSTM is skipped, mean / var are the prior's.  A per-row VECTOR prior mean
        # (SENTIMENT_VAE = 2, or one handed in) is applied with two elementwise torch ops - this stand-alone step is the
        # module-level API, not the hot path (the fused decode step handles the scalar priors of modes 0 / 1).
        pm = pooled if pooled is not None else (prior_mean.to(dev, torch.float32) if prior_mean is not None
                                                else torch.zeros(G, Z, **f32))
        pv = prior_var.to(dev, torch.float32) if prior_var is not None else torch.full((G, Z), float(dims.prior_var), **f32)
        mu, lv = pm, pv.log()
        z = (eps * pv.sqrt() + pm).contiguous()
        he, ce = st["h_encoder"], st["c_encoder"]
    else:
        _gemm(lib, [(att.data_ptr(), F, be, lde, F), (h1.data_ptr(), H, be + 4 * F, lde, H),
                    (hd_prev.data_ptr(), H, be + 4 * (F + H), lde, H)] + csegs_e +
              [(st["h_encoder"].data_ptr(), H, w_ehh.data_ptr(), w_ehh.stride(0), H)], G, 4 * H, pre, ws=ws)
        he, ce = _lstm(lib, pre, P[P_ENC + "bias_ih"], P[P_ENC + "bias_hh"], st["c_encoder"], sent if S == 1 else None,
                       be + 4 * (F + 2 * H), lde)
    mulv = torch.empty(G, 2 * Z, **f32)
    if training:
        wm, wl = P[P_CELL + "fc_mean.weight"], P[P_CELL + "fc_log_var.weight"]
        _gemm(lib, [(he.data_ptr(), H, wm.data_ptr(), wm.stride(0), H)], G, Z, mulv[:, :Z], ws=ws)
        _gemm(lib, [(he.data_ptr(), H, wl.data_ptr(), wl.stride(0), H)], G, Z, mulv[:, Z:], ws=ws)
        mu, lv, z = torch.empty(G, Z, **f32), torch.empty(G, Z, **f32), torch.empty(G, Z, **f32)
        kld = torch.zeros(G, **f32)
        ones = torch.ones(G, **f32)
        d = L.LatentFwdDesc()
        d.B, d.Z, d.mulv, d.ldmulv, d.nslab, d.slab_stride = G, Z, mulv.data_ptr(), 2 * Z, 1, 0
        d.bmu, d.blv = P[P_CELL + "fc_mean.bias"].data_ptr(), P[P_CELL + "fc_log_var.bias"].data_ptr()
        d.eps, d.ldeps, d.kld_mode = eps.data_ptr(), Z, dims.kld_mode
        d.sent = sent.data_ptr() if (sent is not None and dims.pm_scale != 0.0) else None
        d.pm_scale, d.prior_var, d.w = dims.pm_scale, dims.prior_var, ones.data_ptr()
        d.mu, d.lv, d.z, d.ldz, d.kld_acc = mu.data_ptr(), lv.data_ptr(), z.data_ptr(), Z, kld.data_ptr()
        lib.ssc_latent_fwd(C.byref(d), L.stream_ptr())   # (the KL term it also accumulates belongs to the captioner; unused here)
    w_d, w_dhh = P[P_DEC + "weight_ih"], P[P_DEC + "weight_hh"]
    ldd, bd = w_d.stride(0), w_d.data_ptr()
    if cond is not None:
        csegs_d = [(cond.data_ptr(), S, bd + 4 * (F + 2 * H), ldd, S)]
    _gemm(lib, [(att.data_ptr(), F, bd, ldd, F), (h1.data_ptr(), H, bd + 4 * F, ldd, H),
                (hd_prev.data_ptr(), H, bd + 4 * (F + H), ldd, H)] + csegs_d +
          [(z.data_ptr(), Z, bd + 4 * (F + 2 * H + S), ldd, Z),
           (hd_prev.data_ptr(), H, w_dhh.data_ptr(), w_dhh.stride(0), H)], G, 4 * H, pre, ws=ws)
    hd, cd = _lstm(lib, pre, P[P_DEC + "bias_ih"], P[P_DEC + "bias_hh"], st["c_decoder"], sent if S == 1 else None,
                   bd + 4 * (F + 2 * H), ldd)
    new = {"h1": h1, "c1": c1, "h_encoder": he, "c_encoder": ce, "h_decoder": hd, "c_decoder": cd}
    return hd, new, mu, lv, alpha, pooled

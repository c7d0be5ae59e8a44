"""GPU: the drop-in module API (var_updown.models.UpDownCaptioner mirror) through autograd, torch.optim.SGD and
clip_grad_norm_ exactly as var_updown/scripts/train.py:154-176 drives it, against reference goldens; the eval
decode step against the reference's _decode_step goldens; fused clip+SGD against the same fixture."""
import pytest
import torch

import oracle
from goldenlib import group, load
from gpuutil import dev, engine_from, maxdiff
from ssc_runtime.vocab import Vocabulary
from var_updown.models import UpDownCaptioner

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_model(cfg: "oracle.OracleConfig", params=None, beam=5, cls=UpDownCaptioner):
    m = cls(Vocabulary.synthetic(cfg.vocab_size), cfg.image_feature_size, cfg.embedding_size, cfg.hidden_size,
            cfg.attention_projection_size, max_caption_length=cfg.max_caption_length, beam_size=beam,
            use_cbs=cfg.tied, z_space=cfg.z_space, prior_std=cfg.prior_std, simple_vae=cfg.simple_vae,
            latent_embedding="glove", sentiment_vae=cfg.sentiment_vae, senti_prior_multip=cfg.senti_prior_multip,
            device=torch.device("cuda"))
    if params is not None:
        sd = dict(params)
        if cfg.tied:
            sd["_output_layer.weight"] = sd["_embedding_layer.weight"]
        m.load_state_dict(sd)
    return m.cuda()


def test_state_dict_keys_and_seeded_init_match_reference():
    d, cfgd = load("g1_train_sv1")
    cfg = oracle.OracleConfig(**cfgd)
    torch.manual_seed(2)  # the fixture's reference model was built under manual_seed(2)
    m = build_model(cfg)
    sd = m.state_dict()
    ref = group(d, "param/")
    assert set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert torch.equal(sd[k].cpu(), v), k
    assert hasattr(m._updown_cell, "_language_lstm_cell_decoder")


def test_dropin_training_loop_matches_reference():
    d, cfgd = load("g6_sgd")
    cfg = oracle.OracleConfig(**cfgd)
    m = build_model(cfg, group(d, "param/"))
    m.train()
    ins = group(d, "in/")
    feats, caps, senti = dev(ins["feats"]), dev(ins["caps"]), dev(ins["sentiment"])
    opt = torch.optim.SGD(m.parameters(), lr=0.015, momentum=0.9, weight_decay=0.001)
    for it in (1, 2):
        for p in m._updown_cell._language_lstm_cell_decoder.parameters():  # train.py:156-161
            p.requires_grad = it != 1
        opt.zero_grad()
        m._eps_override = ins["eps"]
        out = m(feats, None, None, caps, senti)  # positional, as train.py:167
        loss = out["loss"].mean() + out["kld"].mean() / 750.0
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
        opt.step()
        assert abs(float(norm) - float(d[f"out/norm{it}"])) < 1e-4
        sd = m.state_dict()
        for k, v in group(d, f"after{it}/").items():
            assert maxdiff(sd[k], v) < 1e-5, (it, k)


def test_fused_clip_sgd_matches_reference():
    d, cfgd = load("g6_sgd")
    cfg = oracle.OracleConfig(**cfgd)
    eng = engine_from(cfg, group(d, "param/"))
    ins = group(d, "in/")
    args = (dev(ins["feats"]), dev(ins["caps"]), dev(ins["sentiment"]), dev(ins["eps"]))
    for it in (1, 2):
        eng.train_step(*args, lr=0.015, kld_weight=750.0, momentum=0.9, weight_decay=0.001, max_norm=0.5,
                       decoder_frozen=(it == 1))
        sd = eng.state_dict()
        for k, v in group(d, f"after{it}/").items():
            assert maxdiff(sd[k], v) < 1e-5, (it, k)


@pytest.mark.parametrize("name", ["g5_decode_sv1", "g5b_decode_sv0"])
def test_eval_decode_step_matches_reference(name):
    d, cfgd = load(name)
    cfg = oracle.OracleConfig(**cfgd)
    m = build_model(cfg, group(d, "param/"))
    m.eval()
    ins = group(d, "in/")
    feats, senti = dev(ins["feats"]), dev(ins["sentiment"])
    m._eps_override = [ins["eps0"], ins["eps1"]]
    lp0, st0, pm, plv, al0 = m._decode_step(feats, None, dev(ins["tok0"]), None, sentiment=senti)
    assert maxdiff(lp0, torch.from_numpy(d["out/lp0"])) < TOL
    assert maxdiff(al0, torch.from_numpy(d["out/alpha0"])) < TOL
    for k, v in group(d, "out/st0/").items():
        assert maxdiff(st0[k], v) < TOL, k
    st_in = {k: dev(v) for k, v in group(d, "in/st1/").items()}
    lp1, st1, _, _, al1 = m._decode_step(feats, None, dev(ins["tok1"]), st_in, sentiment=senti)
    assert maxdiff(lp1, torch.from_numpy(d["out/lp1"])) < TOL
    assert maxdiff(al1, torch.from_numpy(d["out/alpha1"])) < TOL
    for k, v in group(d, "out/st1/").items():
        assert maxdiff(st1[k], v) < TOL, k
    assert pm.shape == (ins["tok0"].numel(), cfg.z_space) and plv.shape == pm.shape


def test_large_call_decode_step_matches_reference_on_reordered_states():
    """g15_decode_large: the reference's `_decode_step` at 8 images x 14 groups x beam 5 = 560 rows on states re-ordered by
    back-pointers (cbs.py:236-250).  At this size the build's step takes its large-call paths - per-token gate table, per-image
    attended-feature table contracted on the matrix cores, products over the distinct parents of each beam group - and is handed
    the states (a) re-ordered, with the back-pointers as `_parent`, (b) NOT re-ordered (`_ungathered`: read through the parent
    lists), (c) re-ordered without back-pointers (no sharing): all three equal the reference's result."""
    d, cfgd = load("g15_decode_large")
    cfg = oracle.OracleConfig(**cfgd)
    m = build_model(cfg, group(d, "param/"))
    m.eval()
    m._engine()
    dec = m._dec
    ins = group(d, "in/")
    feats = dev(ins["feats"])
    nimg = feats.size(0)
    G = ins["tok"].numel()
    parent = dev(ins["parent"])                       # (groups, beam)
    NG, beam = parent.shape
    base = {k: dev(v) for k, v in group(d, "in/base/").items()}
    H = base["h1"].size(1)
    gathered = {k: v.view(NG, beam, H).gather(1, parent.view(NG, beam, 1).expand(NG, beam, H)).reshape(G, H).contiguous()
                for k, v in base.items()}
    sent_rows = dev(ins["sentiment"]).view(nimg, 1).expand(nimg, G // nimg).reshape(G).contiguous()
    tok, eps = dev(ins["tok"]), dev(ins["eps"])
    ctx = dec.prepare(feats)
    assert dec.ungathered_ok(ctx, G, beam)            # the large-call paths are really in use at this size
    want_st = group(d, "out/st/")

    def check(states):
        lp, st, al = dec.step(ctx, tok, states, sent_rows, eps)
        assert maxdiff(lp, torch.from_numpy(d["out/lp"])) < TOL
        assert maxdiff(al, torch.from_numpy(d["out/alpha"])) < TOL
        for k in ("h1", "c1", "h_decoder", "c_decoder"):
            assert maxdiff(st[k], want_st[k]) < TOL, k

    a = dict(gathered)
    a["_parent"] = parent
    check(a)
    b = dict(base)
    b["_parent"] = parent
    b["_ungathered"] = True
    check(b)
    check(dict(gathered))


def test_tied_module_and_cpu_call_fails_loudly():
    d, cfgd = load("g3_train_tied")
    cfg = oracle.OracleConfig(**cfgd)
    params = group(d, "param/")

    class Tied(UpDownCaptioner):
        def _initialize_glove(self):
            return params["_embedding_layer.weight"].clone()

    m = build_model(cfg, params, cls=Tied)
    assert m._output_layer.weight is m._embedding_layer.weight and not m._embedding_layer.weight.requires_grad
    m.train()
    ins = group(d, "in/")
    m._eps_override = ins["eps"]
    out = m(dev(ins["feats"]), None, None, dev(ins["caps"]), dev(ins["sentiment"]))
    assert maxdiff(out["loss"], torch.from_numpy(d["out/loss"])) < TOL
    (out["loss"].mean() + out["kld"].mean() / 750.0).backward()
    for k, g in group(d, "grad/").items():
        if k.startswith("_output_layer") or k.startswith("_embedding"):
            continue
        assert maxdiff(dict(m.named_parameters())[k].grad, g) < TOL, k
    assert m._embedding_layer.weight.grad is None
    cpu_model = build_model.__wrapped__(cfg) if hasattr(build_model, "__wrapped__") else None
    m2 = Tied(Vocabulary.synthetic(cfg.vocab_size), cfg.image_feature_size, cfg.embedding_size, cfg.hidden_size,
              cfg.attention_projection_size, max_caption_length=cfg.max_caption_length, use_cbs=True,
              z_space=cfg.z_space, prior_std=1.0, latent_embedding="glove", sentiment_vae=1, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m2(ins["feats"], None, None, ins["caps"], ins["sentiment"])


def test_standalone_training_cell_step_matches_reference_step0_and_1():
    """UpDownCell.forward(training=True) / UpDownCaptioner._decode_step in training mode (op-level composition, no
    autograd) reproduce the reference's per-step states of the training fixture."""
    d, cfgd = load("g1_train_sv1")
    cfg = oracle.OracleConfig(**cfgd)
    params = group(d, "param/")
    m = build_model(cfg, params)
    m.train()
    ins = group(d, "in/")
    feats, caps, senti, eps = dev(ins["feats"]), ins["caps"], dev(ins["sentiment"]), ins["eps"]
    tokens, _ = oracle.add_sentence_boundary_token_ids(caps, caps != 0, 1, 1)
    states = None
    for t in (0, 1):
        m._eps_override = [eps[t]]
        logits, states, mean, log_var, pm, plv, alpha = m._decode_step(feats, None, tokens[:, t].cuda(), states, sentiment=senti)
        st = group(d, f"step{t}/")
        for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder"):
            assert maxdiff(states[k], st[k]) < TOL, (t, k)
        assert maxdiff(alpha, st["alpha"]) < TOL and maxdiff(mean, st["mean"]) < TOL and maxdiff(log_var, st["log_var"]) < TOL
        assert maxdiff(logits, st["logits"]) < TOL
    # the cell API itself
    emb = m._embedding_layer.weight.detach()[tokens[:, 0].cuda()]
    out = m._updown_cell(feats, None, emb, None, True, senti, None, None, None, eps=eps[0])
    assert maxdiff(out[0], group(d, "step0/")["h_decoder"]) < TOL and len(out) == 7


@pytest.mark.parametrize("tag,sv,Z", [("sv2", 2, 150), ("sv1", 1, 16), ("sv0", 0, 16)])
def test_bare_updown_cell_matches_reference_cell(tag, sv, Z):
    """A bare UpDownCell (no captioner) on the HIP path against the reference's own UpDownCell.forward
    (var_updown/var_updown/modules/updown_cell.py:86-231; fixture tests/golden/g11_cell.npz from the imported reference):
    training and eval steps, all three SENTIMENT_VAE modes - including 2, the attention-grounded style prior
    (c = sum_r alpha_r obj_atts_r as LSTM conditioning and prior mean, :160-163,185-188,219-222).  Tolerance 1e-4 fp32."""
    import numpy as np
    import os
    from var_updown.modules import UpDownCell
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_cell.npz"))
    t = lambda k: torch.from_numpy(z[f"{tag}/{k}"])
    torch.manual_seed(7)
    cell = UpDownCell(64, 40, 48, 32, Z, sv, False, torch.device("cuda"), "glove")
    sd = {k[len(tag) + 7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}/param/")}
    for k, v in cell.state_dict().items():
        assert torch.equal(v, sd[k]), k          # same constructor order -> same seeded default init as the reference
    cell = cell.cuda()
    st_in = {k: t("in/state/" + k) for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")}
    obj = t("in/obj_atts").cuda() if sv == 2 else None
    for mode, training in (("train", True), ("eval", False)):
        hd, st, mean, lv, pm, plv, al = cell(t("in/feats").cuda(), obj, t("in/emb").cuda(), {k: v.cuda() for k, v in st_in.items()},
                                             training, t("in/sentiment").cuda(), None, t("in/prior_mean").cuda(),
                                             t("in/prior_var").cuda(), eps=t(f"{mode}/eps"))
        assert maxdiff(al, t(f"{mode}/alpha")) < 1e-5
        assert maxdiff(hd, t(f"{mode}/h_dec")) < 1e-4
        for k in st_in:
            assert maxdiff(st[k], t(f"{mode}/state/{k}")) < 1e-4, (mode, k)
        assert maxdiff(mean, t(f"{mode}/mean")) < 1e-4 and maxdiff(lv, t(f"{mode}/log_var")) < 1e-4
        assert maxdiff(pm, t(f"{mode}/prior_mean")) < 1e-4 and maxdiff(plv, t(f"{mode}/prior_log_var")) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,Z,nslab", [(64, 1200, 128, 13), (5, 37, 13, 3), (70, 50, 6, 0)])
def test_lstm_fwd_z_equals_cell_with_precomputed_latent_block(B, H, Z, nslab):
    """ssc_lstm_fwd_z (decoder cell with the K = Z latent block z . Wz^T formed inside the kernel) against ssc_lstm_fwd fed the
    same block through add0, and against the LSTMCell pointwise formulas in float64 (updown_cell.py:211-229).  1e-5: the two
    kernels sum in a different order; the in-kernel product is exact-fp32 MFMA."""
    import ctypes as C
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H)
    dev = "cuda"
    Zp = (Z + 3) // 4 * 4
    slabs = (torch.randn(max(nslab, 1), B, 4 * H, generator=g) * 0.3).to(dev)
    z = torch.zeros(B, Zp)
    z[:, :Z] = torch.randn(B, Z, generator=g)
    wz = torch.full((4 * H, Zp), float("nan"))      # pad columns must never be read
    wz[:, :Z] = torch.randn(4 * H, Z, generator=g) * 0.2
    z, wz = z.to(dev), wz.to(dev)
    b_ih, b_hh = torch.randn(4 * H, generator=g).to(dev), torch.randn(4 * H, generator=g).to(dev)
    sent, wcol = torch.randn(B, generator=g).to(dev), torch.randn(4 * H, generator=g).to(dev)
    c_prev = torch.randn(B, H, generator=g).to(dev)

    def run(with_z):
        h, c, gates = (torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, 4 * H, device=dev))
        d = L.LstmFwdDesc()
        d.B, d.H = B, H
        if nslab:
            d.slabs, d.nslab, d.slab_stride = slabs.data_ptr(), nslab, B * 4 * H
        d.b_ih, d.b_hh = b_ih.data_ptr(), b_hh.data_ptr()
        d.sent, d.wcol, d.ldwcol = sent.data_ptr(), wcol.data_ptr(), 1
        d.c_prev, d.ld_cprev = c_prev.data_ptr(), H
        d.gates_out = gates.data_ptr()
        d.c_out, d.ld_cout, d.h_out, d.ld_hout = c.data_ptr(), H, h.data_ptr(), H
        if with_z:
            lib.ssc_lstm_fwd_z(C.byref(d), z.data_ptr(), Zp, wz.data_ptr(), Zp, Z, L.stream_ptr())
        else:
            blk = (z[:, :Z].double() @ wz[:, :Z].double().t()).float().contiguous()
            d.add0, d.ld_add0 = blk.data_ptr(), 4 * H
            lib.ssc_lstm_fwd(C.byref(d), L.stream_ptr())
        torch.cuda.synchronize()
        return h, c, gates

    hz, cz, gz = run(True)
    h0, c0, g0 = run(False)
    for a, b in ((hz, h0), (cz, c0), (gz, g0)):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() < 1e-5
    pre = (slabs[:nslab].double().sum(0) if nslab else 0) + z[:, :Z].double() @ wz[:, :Z].double().t() + b_ih.double() + b_hh.double() \
        + sent.double()[:, None] * wcol.double()[None, :]
    i, f, gg, o = pre.split(H, dim=1)
    cw = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    hw = torch.sigmoid(o) * torch.tanh(cw)
    assert (cz.double() - cw).abs().max().item() < 1e-5 and (hz.double() - hw).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,K,nA", [(64, 1200, 256, 5), (5, 37, 26, 0), (70, 50, 300, 9), (64, 1200, 768, 5)])
def test_lstm_bwd_x_equals_cell_backward_with_precomputed_addend(B, H, K, nA):
    """ssc_lstm_bwd_x (LSTM cell backward with the addend dh += x . w formed inside the kernel; BPTT of the encoder LSTM with
    x = (dmu | dlv), w = [W_mu ; W_lv], updown_cell.py:196-197) against ssc_lstm_bwd fed the same product through dh2."""
    import ctypes as C
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H + K)
    dev = "cuda"
    Hp = (H + 3) // 4 * 4
    x = torch.randn(B, K, generator=g).to(dev)
    w = torch.full((K, Hp), float("nan"))
    w[:, :H] = torch.randn(K, H, generator=g) * 0.1
    w = w.to(dev)
    gates = torch.rand(B, 4 * H, generator=g).to(dev)
    c_prev, c_new = torch.randn(B, H, generator=g).to(dev), torch.randn(B, H, generator=g).to(dev)
    dh_in, dc_in = torch.randn(B, H, generator=g).to(dev), torch.randn(B, H, generator=g).to(dev)
    slabs = (torch.randn(max(nA, 1), B, H, generator=g) * 0.3).to(dev)

    def run(fused):
        dG, dcp = torch.empty(B, 4 * H, device=dev), torch.empty(B, H, device=dev)
        d = L.LstmBwdDesc()
        d.B, d.H = B, H
        d.dh, d.ld_dh = dh_in.data_ptr(), H
        d.dc_in, d.ld_dcin = dc_in.data_ptr(), H
        d.gates = gates.data_ptr()
        d.c_prev, d.ld_cprev, d.c_new, d.ld_cnew = c_prev.data_ptr(), H, c_new.data_ptr(), H
        d.dG, d.dc_prev, d.ld_dcprev = dG.data_ptr(), dcp.data_ptr(), H
        if nA:
            d.slabsA, d.nA, d.strideA = slabs.data_ptr(), nA, B * H
        if fused:
            lib.ssc_lstm_bwd_x(C.byref(d), x.data_ptr(), K, w.data_ptr(), Hp, K, L.stream_ptr())
        else:
            add = (x.double() @ w[:, :H].double()).float().contiguous()
            d.dh2, d.ld_dh2 = add.data_ptr(), H
            lib.ssc_lstm_bwd(C.byref(d), L.stream_ptr())
        torch.cuda.synchronize()
        return dG, dcp

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.isfinite(u).all()
        assert (u - v).abs().max().item() < 2e-5 * max(1.0, float(v.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,NP,nslab", [(64, 1200, 256, 12), (5, 37, 26, 3), (70, 50, 12, 0)])
def test_lstm_fwd_p_partial_products_sum_to_the_linear_layer(B, H, NP, nslab):
    """ssc_lstm_fwd_p = ssc_lstm_fwd + partial products of its h with an nn.Linear weight (NP x H): the ceil(H/16) slabs sum to
    h @ wp^T (fc_mean | fc_log_var after the encoder LSTM, updown_cell.py:196-197); cell outputs equal ssc_lstm_fwd's bit for bit."""
    import ctypes as C
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H + NP)
    dev = "cuda"
    Hp = (H + 3) // 4 * 4
    slabs = (torch.randn(max(nslab, 1), B, 4 * H, generator=g) * 0.3).to(dev)
    wp = torch.full((NP, Hp), float("nan"))
    wp[:, :H] = torch.randn(NP, H, generator=g) * 0.2
    wp = wp.to(dev)
    b_ih, b_hh = torch.randn(4 * H, generator=g).to(dev), torch.randn(4 * H, generator=g).to(dev)
    c_prev = torch.randn(B, H, generator=g).to(dev)
    ns = (H + 15) // 16

    def run(fused):
        h, c, gates = (torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, 4 * H, device=dev))
        pout = torch.full((ns, B, NP), float("nan"), device=dev)
        d = L.LstmFwdDesc()
        d.B, d.H = B, H
        if nslab:
            d.slabs, d.nslab, d.slab_stride = slabs.data_ptr(), nslab, B * 4 * H
        d.b_ih, d.b_hh = b_ih.data_ptr(), b_hh.data_ptr()
        d.c_prev, d.ld_cprev = c_prev.data_ptr(), H
        d.gates_out = gates.data_ptr()
        d.c_out, d.ld_cout, d.h_out, d.ld_hout = c.data_ptr(), H, h.data_ptr(), H
        if fused:
            lib.ssc_lstm_fwd_p(C.byref(d), wp.data_ptr(), Hp, NP, pout.data_ptr(), L.stream_ptr())
        else:
            lib.ssc_lstm_fwd(C.byref(d), L.stream_ptr())
        torch.cuda.synchronize()
        return h, c, gates, pout

    hp, cp_, gp, pout = run(True)
    h0, c0, g0, _ = run(False)
    assert torch.equal(hp, h0) and torch.equal(cp_, c0) and torch.equal(gp, g0)
    want = hp.double() @ wp[:, :H].double().t()
    assert torch.isfinite(pout).all()
    assert (pout.double().sum(0) - want).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("G,R,A,F,rpi", [(64, 36, 768, 2048, 1), (128, 100, 768, 2048, 1), (3, 5, 32, 64, 1), (7, 70, 1024, 100, 1),
                                         (10, 36, 260, 4096, 5), (9, 130, 64, 2052, 3)])
def test_attention_forward_matches_float64(G, R, A, F, rpi):
    """ssc_attn_fwd (attention.py:69-95 + updown_cell.py:151-158) against float64: logits, allennlp masked softmax (masked logits
    zeroed, not -inf'd, then renormalised with 1e-13) and the weighted sum, with zero-padded regions, rows sharing an image
    (rpi > 1) and widths that are no multiple of the kernels' 256-column chunks.  2e-5 on values of order 1: hardware exp2 / rcp
    in the tanh (ssc_tanh_fast, <= 2 ulp) and the summation order."""
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(G * 131 + R)
    n_img = (G + rpi - 1) // rpi
    feats = torch.randn(n_img, R, F, generator=g)
    mask = torch.ones(n_img, R)
    if R > 4:
        mask[0, R - 3:] = 0.0
        mask[-1, 1] = 0.0
    feats = feats * mask[:, :, None]
    pv = torch.randn(n_img, R, A, generator=g) * mask[:, :, None]
    q, wa = torch.randn(G, A, generator=g), torch.randn(A, generator=g) * 0.3
    img = torch.arange(G) // rpi
    lg = (torch.tanh(q.double()[:, None, :] + pv.double()[img]) * wa.double()).sum(-1)
    m = mask.double()[img]
    p = torch.softmax(lg * m, dim=1) * m
    al_w = p / (p.sum(1, keepdim=True) + 1e-13)
    att_w = (al_w[:, :, None] * feats.double()[img]).sum(1)
    d = "cuda"
    q_d, pv_d, wa_d, mask_d, feats_d = q.to(d), pv.to(d), wa.to(d), mask.to(d), feats.to(d)
    logits, alpha = torch.full((G, R), float("nan"), device=d), torch.full((G, R), float("nan"), device=d)
    ldatt = F + 4
    att = torch.full((G, ldatt), float("nan"), device=d)
    lib.ssc_attn_fwd(L.ptr(q_d), A, L.ptr(pv_d), L.ptr(wa_d), L.ptr(mask_d), L.ptr(feats_d), G, R, A, F, rpi,
                     L.ptr(logits), L.ptr(alpha), L.ptr(att), ldatt, L.stream_ptr())
    torch.cuda.synchronize()
    assert (logits.double().cpu() - lg).abs().max().item() < 2e-5 * max(1.0, lg.abs().max().item())
    assert (alpha.double().cpu() - al_w).abs().max().item() < 2e-5
    assert (att[:, :F].double().cpu() - att_w).abs().max().item() < 2e-5 * max(1.0, att_w.abs().max().item())
    assert torch.isnan(att[:, F:]).all()          # nothing written past the F columns of a row
    if R > 4:
        assert float(alpha[0, R - 3:].abs().max()) == 0.0

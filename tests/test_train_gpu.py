"""GPU: fused T-step forward + BPTT (ssc_train_fwd / ssc_train_bwd through the C ABI) against
(a) the committed golden vectors produced by the reference itself and (b) the CPU oracle on seeded
inputs at medium size.  Tolerance (BASELINE.json north_star): 1e-4 abs, fp32."""
import pytest
import torch

import oracle
from goldenlib import group, load
from gpuutil import dev, engine_from, maxdiff

pytestmark = pytest.mark.gpu

TRAIN = ["g1_train_sv1", "g2_train_sv0", "g3_train_tied", "g4_train_prior", "g4b_train_simple", "g7_train_odd"]
TOL = 1e-4


@pytest.mark.parametrize("name", TRAIN)
def test_train_matches_reference_golden(name):
    d, cfgd = load(name)
    cfg = oracle.OracleConfig(**cfgd)
    params = group(d, "param/")
    ins = group(d, "in/")
    eng = engine_from(cfg, params)
    loss, kld = eng.forward(dev(ins["feats"]), dev(ins["caps"]), dev(ins["sentiment"]), dev(ins["eps"]))
    exp = group(d, "out/")
    assert maxdiff(loss, exp["loss"]) < TOL
    assert maxdiff(kld, exp["kld"]) < TOL
    T = ins["eps"].shape[0]
    names = {0: "h1", 1: "c1", 2: "h_encoder", 3: "c_encoder", 4: "h_decoder", 5: "c_decoder"}
    for t in (0, 1, T - 1):
        st = group(d, f"step{t}/")
        for which, key in names.items():
            assert maxdiff(eng.workspace_view(which)[t + 1], st[key]) < TOL, (t, key)
        assert maxdiff(eng.workspace_view(6)[t], st["alpha"]) < TOL
        assert maxdiff(eng.workspace_view(7)[t], st["mean"]) < TOL
        assert maxdiff(eng.workspace_view(8)[t], st["log_var"]) < TOL
        # logits exist only for rows with a real target (the padded (t, b) rows are skipped on the device; the reference
        # computes and then masks them): caption b is active at step t iff t <= its length
        act = (ins["caps"] != cfg.pad_index).sum(1) >= t
        assert maxdiff(eng.workspace_view(9)[t][act.cuda()], st["logits"][act]) < TOL
    B = loss.numel()
    gl = torch.full((B,), 1.0 / B, device="cuda")
    gk = torch.full((B,), 1.0 / (B * 750.0), device="cuda")
    eng.backward(gl, gk)
    got = eng.grad_dict()
    grads = group(d, "grad/")
    for k, g in grads.items():
        if k == "_output_layer.weight" and cfg.tied:
            continue
        assert maxdiff(got[k], g) < TOL, k
        scale = g.abs().max().item()
        assert maxdiff(got[k], g) <= 1e-4 * max(scale, 1e-3) + 2e-6, (k, scale)


@pytest.mark.parametrize("sv,B,R,dims", [
    (1, 8, 36, dict(V=1000, E=200, H=256, A=128, F=512, Z=64, L=12)),
    (0, 5, 10, dict(V=777, E=100, H=130, A=70, F=260, Z=30, L=9)),
])
def test_train_matches_oracle_medium(sv, B, R, dims):
    cfg = oracle.OracleConfig(vocab_size=dims["V"], image_feature_size=dims["F"], embedding_size=dims["E"],
                              hidden_size=dims["H"], attention_projection_size=dims["A"], z_space=dims["Z"],
                              max_caption_length=dims["L"], sentiment_vae=sv, senti_prior_multip=0.5)
    params = oracle.init_params(cfg, seed=5)
    g = torch.Generator().manual_seed(17)
    L, T = dims["L"], dims["L"] + 1
    feats = torch.randn(B, R, dims["F"], generator=g)
    feats[0, R - 3:] = 0
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(3, L + 1, (1,), generator=g))
        caps[b, :n] = torch.randint(2, dims["V"], (n,), generator=g)
    senti = torch.randint(-1, 2, (B, 1), generator=g).float()
    eps = torch.randn(T, B, dims["Z"], generator=g)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = oracle.train_forward(p, cfg, feats, caps, senti, eps)
    oracle.train_objective(out, cfg).backward()
    eng = engine_from(cfg, params)
    loss, kld = eng.forward(dev(feats), dev(caps), dev(senti), dev(eps))
    assert maxdiff(loss, out["loss"]) < TOL * 5  # loss ~ O(80): relative 1e-5
    assert maxdiff(kld, out["kld"]) < TOL * 5
    gl = torch.full((B,), 1.0 / B, device="cuda")
    gk = torch.full((B,), 1.0 / (B * cfg.kld_weight), device="cuda")
    eng.backward(gl, gk)
    got = eng.grad_dict()
    for k, v in p.items():
        assert maxdiff(got[k], v.grad) < TOL, k


def test_stress_shape_c5_like_regions_and_length():
    """BASELINE configs[4] geometry (R=100 regions, L=40 -> T=41, B=128) at reduced widths: parity vs the oracle."""
    cfg = oracle.OracleConfig(vocab_size=1500, image_feature_size=256, embedding_size=96, hidden_size=128,
                              attention_projection_size=64, z_space=32, max_caption_length=40, sentiment_vae=1,
                              senti_prior_multip=0.5)
    params = oracle.init_params(cfg, seed=9)
    g = torch.Generator().manual_seed(23)
    B, R, L = 128, 100, 40
    feats = torch.randn(B, R, 256, generator=g)
    feats[3, 60:] = 0
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(5, L + 1, (1,), generator=g))
        caps[b, :n] = torch.randint(2, 1500, (n,), generator=g)
    senti = torch.randint(-1, 2, (B, 1), generator=g).float()
    eps = torch.randn(L + 1, B, 32, generator=g)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = oracle.train_forward(p, cfg, feats, caps, senti, eps)
    oracle.train_objective(out, cfg).backward()
    eng = engine_from(cfg, params)
    loss, kld = eng.forward(dev(feats), dev(caps), dev(senti), dev(eps))
    assert maxdiff(loss, out["loss"]) < 1e-3      # loss ~ O(250): 4e-6 relative
    assert maxdiff(kld, out["kld"]) < 1e-3
    eng.backward(torch.full((B,), 1.0 / B, device="cuda"), torch.full((B,), 1.0 / (B * cfg.kld_weight), device="cuda"))
    got = eng.grad_dict()
    for k, v in p.items():
        assert maxdiff(got[k], v.grad) < TOL, k

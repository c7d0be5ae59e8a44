"""GPU: SENTIMENT_VAE = 2 (attention-grounded style prior) through the drop-in MODULE API - constructor with a
caller-supplied attribute table, forward(..., obj_atts) training, eval _decode_step - against fixtures made by the reference's own
code (tests/golden/make_golden.py::sv2_module_fixture, sv2_train_fixture): the reference UpDownCell under autograd / in eval mode
(var_updown/var_updown/modules/updown_cell.py:86-231) inside a thin restatement of the captioner around it (the reference captioner
itself cannot be constructed in this mode, updown_captioner.py:79,89).  Tolerance: 1e-4 (north_star)."""
import json

import numpy as np
import pytest
import torch

from goldenlib import load_raw
from gpuutil import maxdiff
from ssc_runtime.vocab import Vocabulary
from var_updown.models import UpDownCaptioner

pytestmark = pytest.mark.gpu
_G16 = load_raw("g16_sv2_module")
_G14 = load_raw("g14_train_sv2")


def _group(d, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


def _model(V, E, H, A, F, Z, L, latent, prior_std, params):
    m = UpDownCaptioner(Vocabulary.synthetic(V), F, E, H, A, max_caption_length=L, beam_size=3, z_space=Z, prior_std=prior_std,
                        simple_vae=False, latent_embedding=latent, latent_embedding_multip=1, sentiment_vae=2, senti_prior_multip=1.0,
                        device=torch.device("cuda"), mean_choice={})
    m.load_state_dict(params)
    return m.cuda()


@pytest.mark.parametrize("latent,Z", [("glove", 150), ("senti_word_net", 16)])
def test_eval_decode_steps_equal_the_reference_cell(latent, Z):
    """Four consecutive eval _decode_step calls (states fed back) on 3 images x 4 rows: log-probs, states, attention weights and the
    pooled prior mean the step returns, for both LATENT_EMBEDDING modes (150 conditioning columns / one)."""
    tag = "eval_" + latent
    V, E, H, A, F, R, nimg, rpi = 90, 40, 48, 32, 64, 5, 3, 4
    m = _model(V, E, H, A, F, Z, 8, latent, 0.8, _group(_G16, tag + "/param/")).eval()
    feats = torch.from_numpy(_G16[tag + "/in/feats"]).cuda()
    obj = torch.from_numpy(_G16[tag + "/in/obj_atts"]).cuda()
    G = nimg * rpi
    states = None
    for t in range(4):
        s = _group(_G16, f"{tag}/step{t}/")
        m._eps_override = [s["eps"].cuda()]
        if t == 0:
            states = {k: torch.zeros(G, H, device="cuda") for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")}
        with torch.no_grad():
            lp, states, pm, plv, alpha = m._decode_step(feats, obj, s["tok"].cuda(), states)
        assert maxdiff(lp, s["log_probs"]) < 1e-4, t
        assert maxdiff(alpha, s["alpha"]) < 1e-5, t
        assert maxdiff(pm, s["prior_mean"]) < 1e-5, t            # the attention-pooled attribute means (updown_cell.py:160-163)
        assert maxdiff(plv, s["prior_log_var"]) < 1e-6, t
        for k in ("h1", "c1", "h_decoder", "c_decoder"):
            assert maxdiff(states[k], s["state/" + k]) < 1e-4, (t, k)


def _train_case(params, grads, inp, out, latent, Z, dims):
    V, E, H, A, F, L = dims
    m = _model(V, E, H, A, F, Z, L, latent, 0.9, params).train()
    feats, caps, eps, obj = (inp[k].cuda() for k in ("feats", "caps", "eps", "obj_atts"))
    m._eps_override = eps
    res = m(feats, obj, None, caps, None)                          # positional, as train.py:167 calls the model
    assert maxdiff(res["loss"], out["loss"]) < 1e-4 * max(1.0, float(out["loss"].abs().max()))
    assert maxdiff(res["kld"], out["kld"]) < 1e-4 * max(1.0, float(out["kld"].abs().max()))
    (res["loss"].mean() + res["kld"].mean() / 750.0).backward()
    got = dict(m.named_parameters())
    for k, v in grads.items():
        scale = float(v.abs().max())
        assert maxdiff(got[k].grad, v) <= 1e-4 * max(scale, 1e-3) + 2e-6, (k, scale, maxdiff(got[k].grad, v))


def test_training_forward_backward_glove_through_the_module():
    """UpDownCaptioner(sentiment_vae=2, latent_embedding='glove').forward(feats, obj_atts, ..) + autograd backward == g14 (the
    reference cell driven through T teacher-forced steps under autograd): loss, KL, every gradient."""
    _train_case(_group(_G14, "param/"), _group(_G14, "grad/"), _group(_G14, "in/"), _group(_G14, "out/"), "glove", 150,
                (120, 40, 48, 32, 64, 6))


def test_training_forward_backward_senti_word_net_through_the_module():
    """The same with LATENT_EMBEDDING 'senti_word_net': the language LSTMs see ONE conditioning column, the first entry of the
    pooled attribute means (updown_cell.py:171-172), whose gradient flows back into the attention weights."""
    t = "train_swn/"
    _train_case(_group(_G16, t + "param/"), _group(_G16, t + "grad/"), _group(_G16, t + "in/"), _group(_G16, t + "out/"),
                "senti_word_net", 16, (120, 40, 48, 32, 64, 6))


@pytest.mark.parametrize("latent,Z", [("glove", 150), ("senti_word_net", 16)])
def test_large_decode_calls_of_mode_2_equal_the_small_call_paths(latent, Z):
    """SENTIMENT_VAE = 2 on the large-call decode paths (>= 512 rows: attended-feature table, parent lists, un-gathered states)
    against the same search with those paths switched off: identical captions."""
    from ssc_runtime import lib as L
    from ssc_runtime.inference import diverse_decode
    lib = L.load()
    V, E, H, A, F, R = 300, 40, 64, 24, 64, 6
    torch.manual_seed(5)
    m = UpDownCaptioner(Vocabulary.synthetic(V), F, E, H, A, max_caption_length=7, beam_size=5, z_space=Z, prior_std=0.9,
                        latent_embedding=latent, sentiment_vae=2, device=torch.device("cuda"), mean_choice={}).cuda().eval()
    m._engine()
    g = torch.Generator().manual_seed(4)
    nimg, ns, beam = 8, 16, 5
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    obj = (torch.randn(nimg, R, Z, generator=g) * 0.5).cuda()
    B = nimg * ns
    eps = [torch.randn(B, Z, generator=g).cuda()] + [torch.randn(B * beam, Z, generator=g).cuda() for _ in range(6)]
    outs = []
    for on in (1, 0):
        for key in (b"dec_dedup", b"dec_att_table", b"dec_ungathered"):
            lib.ssc_debug_set(key, on)
        try:
            outs.append(diverse_decode(m._dec, feats, None, ns, beam, 7, 1, eps_steps=[e.clone() for e in eps], early_stop=False,
                                       obj_means=obj)[0].clone())
        finally:
            for key in (b"dec_dedup", b"dec_att_table", b"dec_ungathered"):
                lib.ssc_debug_set(key, 1)
    assert torch.equal(outs[0], outs[1])


def test_harness_trains_mode_2_on_synthetic_attribute_means(tmp_path):
    """scripts/train.py with MODEL.SENTIMENT_VAE 2: the batch producer carries the per-region attribute means next to the features and
    both optimiser paths run (loss finite and equal between the fused step and the autograd module path for the first iteration)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tmp_path / "c.yaml"
    cfg.write_text("RANDOM_SEED: 3\nDATA:\n  MAX_CAPTION_LENGTH: 6\nMODEL:\n  IMAGE_FEATURE_SIZE: 64\n  EMBEDDING_SIZE: 40\n  HIDDEN_SIZE: 48\n"
                   "  ATTENTION_PROJECTION_SIZE: 32\n  Z_SPACE: 150\n  SENTIMENT_VAE: 2\n  LATENT_EMBEDDING: glove\n  PRIOR_STD: 0.9\n"
                   "OPTIM:\n  BATCH_SIZE: 8\n  NUM_ITERATIONS: 4\n")
    losses = []
    for extra in ([], ["--fused-optimizer"]):
        out = tmp_path / ("run" + str(len(losses)))
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train.py"), "--config", str(cfg), "--gpu-ids", "0",
                            "--serialization-dir", str(out), "--synthetic", "32", "--vocab-size", "200", "--num-boxes", "5", "--zero-eps"]
                           + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        recs = [json.loads(x) for x in open(out / "scalars.jsonl")]
        assert all(np.isfinite(x["3loss"]) for x in recs) and len(recs) >= 1
        losses.append(recs[0]["3loss"])
    assert abs(losses[0] - losses[1]) < 1e-3 * abs(losses[0])

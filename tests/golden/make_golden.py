#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Run in the build container only (``/root/reference`` does not travel to the GPU
box):  ``python tests/golden/make_golden.py``

What it does
------------
* Puts ``/root/reference/var_updown`` and ``/root/reference/updown-baseline`` on
  ``sys.path`` and imports the reference's unmodified ``UpDownCaptioner`` /
  ``UpDownCell`` / ``BottomUpTopDownAttention`` (nothing is copied).
* Third-party packages the reference needs but that are absent here
  (allennlp==0.8.4, torchtext, yacs) are provided as in-memory
  module objects: the four allennlp tensor utilities are the oracle's
  restatements (``oracle/seqcvae_oracle.py``); everything else is a bare symbol.
  Consequence: fixtures pin the oracle against the reference's OWN code;
  the allennlp boundary itself stays "parity unpinned".
* ``eps`` is injected by temporarily rebinding ``torch.randn`` around the
  reference call (the reference calls ``torch.randn(var.shape)`` exactly once
  per step, var_updown/var_updown/modules/updown_cell.py:206).
* A fresh ``image_features`` tensor object is passed on every forward (the
  reference lru-caches on tensor identity).

Fixtures are plain ``.npz`` (inputs, weights, expected outputs) - data only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import oracle  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from goldenlib import cbs_table_step  # noqa: E402  (the step function the g12 search cases are defined on)


class ToyVocabulary:
    """Duck-type of allennlp.data.Vocabulary as used by the hot path
    (get_vocab_size / get_token_index / get_token_from_index / get_token_to_index_vocabulary)."""

    def __init__(self, size):
        self._tokens = ["@@UNKNOWN@@", "@@BOUNDARY@@"] + [f"w{i}" for i in range(2, size)]
        self._index = {t: i for i, t in enumerate(self._tokens)}

    def get_vocab_size(self, namespace="tokens"):
        return len(self._tokens)

    def get_token_index(self, token, namespace="tokens"):
        return self._index.get(token, 0)

    def get_token_from_index(self, index, namespace="tokens"):
        return self._tokens[index]

    def get_token_to_index_vocabulary(self, namespace="tokens"):
        return dict(self._index)


def install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Sym:  # bare symbol; never instantiated on the exercised path (BeamSearch is constructed, unused)
        def __init__(self, *a, **k):
            pass

    mod("allennlp")
    mod("allennlp.data", Vocabulary=ToyVocabulary)
    mod("allennlp.nn")
    mod("allennlp.nn.util",
        masked_softmax=oracle.masked_softmax,
        masked_mean=lambda v, m, dim, keepdim=False, eps=1e-8: oracle.masked_mean(v, m, dim, eps),
        add_sentence_boundary_token_ids=oracle.add_sentence_boundary_token_ids,
        sequence_cross_entropy_with_logits=lambda lo, ta, w, average=None: oracle.sequence_cross_entropy_with_logits(lo, ta, w))
    mod("allennlp.nn.beam_search", BeamSearch=_Sym)
    mod("torchtext")
    mod("torchtext.vocab", GloVe=_Sym, Vectors=_Sym)
    mod("yacs")
    mod("yacs.config", CfgNode=_Sym)


def import_reference():
    install_standins()
    sys.path[:0] = [os.path.join(REF, "var_updown"), os.path.join(REF, "updown-baseline")]
    from var_updown.models import UpDownCaptioner  # noqa
    return UpDownCaptioner


class EpsInjector:
    def __init__(self, eps_list):
        self.eps = list(eps_list)
        self.k = 0

    def __enter__(self):
        self._orig = torch.randn

        def fake(*shape, **kw):
            e = self.eps[self.k]
            self.k += 1
            shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert tuple(e.shape) == shp, (e.shape, shp)
            return e.clone()

        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self._orig


def make_inputs(seed, B, R, F, L, V, Z, T, pad_region_row=True, sv=1, unk=False):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, R, F, generator=g)
    if pad_region_row:  # one image with two zero-padded regions (adaptive features)
        feats[1, R - 2:] = 0
    caps = torch.zeros(B, L, dtype=torch.long)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L  # one full-length caption
    for b in range(B):
        caps[b, : lens[b]] = torch.randint(2, V, (int(lens[b]),), generator=g)
    if unk:
        # in-caption @@UNKNOWN@@ (id 0 = the padding id; every out-of-vocabulary word in real data): the boundary END then
        # lands on column count_nonzero + 1 (over a real token) and the UNK step has loss weight 0 while LATER steps of the
        # same caption have weight 1 (updown_captioner.py:265-278)
        caps[0, 2] = 0                      # the full-length caption
        caps[1, 0] = 0                      # first word unknown
        if B > 2 and int(lens[2]) > 3:
            caps[2, 1] = 0
            caps[2, 3] = 0                  # two unknown words in one caption
        if B > 3:
            caps[3, int(lens[3]) - 1] = 0   # last word unknown (indistinguishable from a shorter caption)
    senti = torch.randint(-1, 2, (B, 1), generator=g).float()
    eps = torch.randn(T, B, Z, generator=g)
    return feats, caps, senti, eps


def build_reference_model(UpDownCaptioner, dims, sv, simple_vae=False, prior_std=1.0, multip=1.0, tied=False, seed=2):
    V, E, H, A, F, Z, L = dims
    torch.manual_seed(seed)
    cls = UpDownCaptioner
    if tied:
        class Tied(UpDownCaptioner):  # local seeded table instead of the GloVe download (:189,199)
            def _initialize_glove(self):
                t = torch.randn(self._vocabulary.get_vocab_size(), self.embedding_size,
                                generator=torch.Generator().manual_seed(77)) * 0.5
                t[0] = 0
                return t
        cls = Tied
    model = cls(ToyVocabulary(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                attention_projection_size=A, max_caption_length=L, beam_size=5, use_cbs=False if not tied else True,
                z_space=Z, prior_std=prior_std, simple_vae=simple_vae, latent_embedding="glove",
                sentiment_vae=sv, senti_prior_multip=multip, device=torch.device("cpu"))
    return model


def oracle_cfg(dims, sv, simple_vae=False, prior_std=1.0, multip=1.0, tied=False):
    V, E, H, A, F, Z, L = dims
    return dict(vocab_size=V, image_feature_size=F, embedding_size=E, hidden_size=H, attention_projection_size=A,
                z_space=Z, max_caption_length=L, sentiment_vae=sv, simple_vae=simple_vae, prior_std=prior_std,
                senti_prior_multip=multip, tied=tied)


def state_dict_np(model):
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        if k == "_output_layer.weight" and "_output_projection.0.weight" in sd:
            continue  # tied: same storage as the embedding
        out["param/" + k] = v.detach().numpy().copy()
    return out


def train_fixture(UpDownCaptioner, name, dims, sv, B=3, R=5, unk=False, **kw):
    V, E, H, A, F, Z, L = dims
    T = L + 1
    model = build_reference_model(UpDownCaptioner, dims, sv, **kw)
    model.train()
    feats, caps, senti, eps = make_inputs(1234, B, R, F, L, V, Z, T, unk=unk)
    # hook per-step states through _decode_step
    steps = []
    orig = model._decode_step

    def spy(*a, **k):
        out = orig(*a, **k)
        st = out[1]
        steps.append({**{kk: vv.detach().clone() for kk, vv in st.items()}, "alpha": out[6].detach().clone(),
                      "mean": out[2].detach().clone(), "log_var": out[3].detach().clone(),
                      "logits": out[0].detach().clone()})
        return out

    model._decode_step = spy
    with EpsInjector([eps[t] for t in range(T)]):
        out = model(feats.clone(), None, None, caps, senti)
    kld_weight = 750.0
    obj = out["loss"].mean() + out["kld"].mean() / kld_weight
    obj.backward()
    data = state_dict_np(model)
    data.update({"in/feats": feats.numpy(), "in/caps": caps.numpy(), "in/sentiment": senti.numpy(),
                 "in/eps": eps.numpy(), "out/loss": out["loss"].detach().numpy(),
                 "out/kld": out["kld"].detach().numpy()})
    for n, p in model.named_parameters():
        if p.grad is not None:
            data["grad/" + n] = p.grad.numpy().copy()
    for t in (0, 1, T - 1):
        for k, v in steps[t].items():
            data[f"step{t}/{k}"] = v.numpy()
    cfg = oracle_cfg(dims, sv, **kw)
    data["cfg"] = np.array(repr(cfg))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "loss", out["loss"].detach().numpy(), "kld", out["kld"].detach().numpy())
    return model, (feats, caps, senti, eps)


def sgd_fixture(UpDownCaptioner, name, dims, sv):
    """One clip + SGD(momentum, wd) update, decoder LSTM frozen (train.py:126-131,156-176)."""
    V, E, H, A, F, Z, L = dims
    T = L + 1
    model = build_reference_model(UpDownCaptioner, dims, sv)
    model.train()
    feats, caps, senti, eps = make_inputs(4321, 3, 5, F, L, V, Z, T)
    opt = torch.optim.SGD(model.parameters(), lr=0.015, momentum=0.9, weight_decay=0.001)
    data = state_dict_np(model)
    data.update({"in/feats": feats.numpy(), "in/caps": caps.numpy(), "in/sentiment": senti.numpy(),
                 "in/eps": eps.numpy()})
    for it in (1, 2):  # two iterations: exercises the momentum recurrence; decoder frozen at it=1, trained at it=2
        frozen = it == 1
        for p in model._updown_cell._language_lstm_cell_decoder.parameters():
            p.requires_grad = not frozen
        opt.zero_grad()
        with EpsInjector([eps[t] for t in range(T)]):
            out = model(feats.clone(), None, None, caps, senti)
        (out["loss"].mean() + out["kld"].mean() / 750.0).backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)  # small max-norm so clipping is active
        opt.step()
        data[f"out/norm{it}"] = np.array(float(norm))
        for k, v in model.state_dict().items():
            data[f"after{it}/" + k] = v.detach().numpy().copy()
    data["cfg"] = np.array(repr(oracle_cfg(dims, sv)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "norms", data["out/norm1"], data["out/norm2"])


def decode_fixture(UpDownCaptioner, name, dims, sv, B=2, R=5, beam=5):
    """Eval-mode ``_decode_step`` at G == B (first call) and G == beam*B (updown_captioner.py:371-455)."""
    V, E, H, A, F, Z, L = dims
    model = build_reference_model(UpDownCaptioner, dims, sv, multip=0.5)
    model.eval()
    g = torch.Generator().manual_seed(99)
    feats = torch.randn(B, R, F, generator=g)
    feats[1, R - 1:] = 0
    senti = torch.tensor([[1.0]] * B)  # equal per-row sentiment: tile-order quirk (:418-424) is then invisible
    prior_mean = senti.repeat(1, Z) * 0.5 if sv == 1 else torch.zeros(B, Z)
    prior_var = torch.ones(B, Z)
    data = state_dict_np(model)
    data.update({"in/feats": feats.numpy(), "in/sentiment": senti.numpy()})
    with torch.no_grad():
        tok0 = torch.full((B,), 1, dtype=torch.long)
        eps0 = torch.randn(B, Z, generator=g)
        with EpsInjector([eps0]):
            lp0, st0, _, _, al0 = model._decode_step(feats.clone(), None, tok0, None, senti, None, prior_mean, prior_var)
        G = B * beam
        tok1 = torch.randint(1, V, (G,), generator=g)
        st_in = {k: torch.randn(G, H, generator=g) * 0.3 for k in st0}
        eps1 = torch.randn(G, Z, generator=g)
        with EpsInjector([eps1]):
            lp1, st1, _, _, al1 = model._decode_step(feats.clone(), None, tok1, {k: v.clone() for k, v in st_in.items()},
                                                     senti, None, prior_mean, prior_var)
    data.update({"in/tok0": tok0.numpy(), "in/eps0": eps0.numpy(), "out/lp0": lp0.numpy(), "out/alpha0": al0.numpy(),
                 "in/tok1": tok1.numpy(), "in/eps1": eps1.numpy(), "out/lp1": lp1.numpy(), "out/alpha1": al1.numpy()})
    for k in st0:
        data["out/st0/" + k] = st0[k].numpy()
        data["in/st1/" + k] = st_in[k].numpy()
        data["out/st1/" + k] = st1[k].numpy()
    data["cfg"] = np.array(repr(oracle_cfg(dims, sv, multip=0.5)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "lp0[0,:3]", lp0[0, :3].numpy())


def decode_large_fixture(UpDownCaptioner, name="g15_decode_large", dims=(300, 40, 48, 32, 64, 16, 6), sv=1, nimg=8, groups=14, beam=5,
                         R=7):
    """One eval-mode ``_decode_step`` (updown_captioner.py:371-455) at a LARGE call: nimg images x groups x beam rows = 560 rows
    whose recurrent states look like those a beam search hands over after a re-ordering (cbs.py:236-250): the `beam` rows of a
    group hold the states of the parents the back-pointers name, so several rows of a group share their states.  The build's
    large-call decode paths (per-token gate table, per-image attended-feature table, products over the distinct parents, states
    read through the parent lists) are pinned against THIS, the reference's own step on the re-ordered states."""
    V, E, H, A, F, Z, L = dims
    model = build_reference_model(UpDownCaptioner, dims, sv, multip=0.5)
    model.eval()
    g = torch.Generator().manual_seed(1515)
    feats = torch.randn(nimg, R, F, generator=g)
    feats[3, R - 2:] = 0
    senti = torch.tensor([[1.0]] * nimg)  # equal per-row sentiment: the tile-order quirk (:418-424) is then invisible
    prior_mean = senti.repeat(1, Z) * 0.5 if sv == 1 else torch.zeros(nimg, Z)
    prior_var = torch.ones(nimg, Z)
    G = nimg * groups * beam
    NG = nimg * groups
    tok = torch.randint(1, V, (G,), generator=g)
    keys = ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")
    base = {k: torch.randn(NG, beam, H, generator=g) * 0.3 for k in keys}    # the previous step's outputs, in ITS row order
    parent = torch.randint(0, beam, (NG, beam), generator=g)                 # back-pointers of the beam step in between
    parent[0] = torch.tensor([0, 0, 0, 0, 0])                                # a group with one parent ...
    parent[1] = torch.arange(beam)                                           # ... and one with five
    st_in = {k: v.gather(1, parent.view(NG, beam, 1).expand(NG, beam, H)).reshape(G, H).contiguous() for k, v in base.items()}
    eps = torch.randn(G, Z, generator=g)
    data = state_dict_np(model)
    with torch.no_grad(), EpsInjector([eps]):
        lp, st, _, _, al = model._decode_step(feats.clone(), None, tok, {k: v.clone() for k, v in st_in.items()}, senti, None,
                                              prior_mean, prior_var)
    data.update({"in/feats": feats.numpy(), "in/sentiment": senti.numpy(), "in/tok": tok.numpy(), "in/eps": eps.numpy(),
                 "in/parent": parent.numpy(), "out/lp": lp.numpy(), "out/alpha": al.numpy()})
    for k in keys:
        data["in/base/" + k] = base[k].reshape(G, H).numpy()
    for k in st:
        data["out/st/" + k] = st[k].numpy()
    data["cfg"] = np.array(repr(oracle_cfg(dims, sv, multip=0.5)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "lp[0,:3]", lp[0, :3].numpy(), "distinct parents", sum(len(set(r.tolist())) for r in parent), "of", G)


def full_size_inputs(seed, B, R=36, F=2048, L=20, V=10000, Z=128, unk=False):
    """Inputs of the full-size fixtures, regenerated from the seed on both sides (make_golden.py here, tests/test_train_gpu.py
    on the GPU box): BASELINE.md 4 distributions; `unk` plants @@UNKNOWN@@ (id 0) inside some captions."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, R, F, generator=g)
    lens = torch.randint(8, L + 1, (B,), generator=g)
    ids = torch.randint(2, V, (B, L), generator=g)
    caps = torch.where(torch.arange(L).unsqueeze(0) < lens.unsqueeze(1), ids, torch.zeros_like(ids))
    if unk:
        for b in range(0, B, 3):
            caps[b, int(lens[b]) // 2] = 0
    senti = torch.randint(-1, 2, (B, 1), generator=g).float()
    eps = torch.randn(L + 1, B, Z, generator=g)
    return feats, caps, senti, eps


def full_size_fixture(UpDownCaptioner, name, B, unk):
    """BASELINE configs[0] / configs[1] at FULL size (V=10000, E=1000, H=1200, A=768, F=2048, Z=128, L=20, SENTIMENT_VAE=1):
    parameters = the reference's default init under manual_seed(2) (the mirror's init is bit-identical, so they are NOT
    stored - only per-tensor checksums that prove it), inputs regenerated from the seed; stored: loss, kld and per-gradient
    {norm, sum, the first 32 and 32 strided entries}.  < 100 KB."""
    dims = (10000, 1000, 1200, 768, 2048, 128, 20)
    V, E, H, A, F, Z, L = dims
    model = build_reference_model(UpDownCaptioner, dims, sv=1, multip=0.5, seed=2)
    model.train()
    feats, caps, senti, eps = full_size_inputs(4242 + B, B, unk=unk)
    with EpsInjector([eps[t] for t in range(L + 1)]):
        out = model(feats.clone(), None, None, caps, senti)
    (out["loss"].mean() + out["kld"].mean() / 750.0).backward()
    data = {"out/loss": out["loss"].detach().numpy(), "out/kld": out["kld"].detach().numpy(), "B": np.array(B),
            "unk": np.array(int(unk)), "seed": np.array(4242 + B)}
    for n, p in model.named_parameters():
        w = p.detach().double()
        data["psum/" + n] = np.array([float(w.sum()), float(w.abs().sum())])
        if p.grad is None:
            continue
        gflat = p.grad.detach().reshape(-1)
        stride = max(1, gflat.numel() // 32)
        data["gnorm/" + n] = np.array([float(gflat.double().norm()), float(gflat.double().sum()), float(gflat.abs().max())])
        data["ghead/" + n] = gflat[:32].numpy().copy()
        data["gstride/" + n] = gflat[::stride][:32].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "loss[:3]", out["loss"][:3].detach().numpy(), "kld[:3]", out["kld"][:3].detach().numpy())


def cell_fixture(name="g11_cell"):
    """The reference's UpDownCell on its own (var_updown/var_updown/modules/updown_cell.py:86-231), one training step and one
    eval step from given states, for SENTIMENT_VAE = 2 (attention-pooled obj_atts as conditioning and prior mean: the branch
    the captioner cannot reach as shipped) and for modes 0 / 1."""
    from var_updown.modules import UpDownCell
    F, E, H, A, R, G = 64, 40, 48, 32, 5, 4
    data = {}
    for tag, sv, Z in (("sv2", 2, 150), ("sv1", 1, 16), ("sv0", 0, 16)):
        torch.manual_seed(7)
        cell = UpDownCell(F, E, H, A, Z, sv, False, torch.device("cpu"), "glove")
        g = torch.Generator().manual_seed(40 + sv)
        feats = torch.randn(G, R, F, generator=g)
        feats[2, R - 2:] = 0
        emb = torch.randn(G, E, generator=g)
        obj = torch.randn(G, R, 150, generator=g) * 0.3 if sv == 2 else None
        senti = torch.randint(-1, 2, (G, 1), generator=g).float()
        st_in = {k: torch.randn(G, H, generator=g) * 0.3 for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")}
        pm = torch.zeros(G, Z) if sv != 1 else senti.repeat(1, Z) * 0.5
        pv = torch.full((G, Z), 0.81)
        for k, v in cell.state_dict().items():
            data[f"{tag}/param/{k}"] = v.numpy().copy()
        data.update({f"{tag}/in/feats": feats.numpy(), f"{tag}/in/emb": emb.numpy(), f"{tag}/in/sentiment": senti.numpy(),
                     f"{tag}/in/prior_mean": pm.numpy(), f"{tag}/in/prior_var": pv.numpy()})
        if obj is not None:
            data[f"{tag}/in/obj_atts"] = obj.numpy()
        for k, v in st_in.items():
            data[f"{tag}/in/state/{k}"] = v.numpy()
        for mode, training in (("train", True), ("eval", False)):
            eps = torch.randn(G, Z, generator=g)
            with torch.no_grad(), EpsInjector([eps]):
                hd, st, mean, lv, pmo, plv, al = cell(feats.clone(), obj, emb, {k: v.clone() for k, v in st_in.items()}, training,
                                                     senti, None, pm.clone(), pv.clone())
            data[f"{tag}/{mode}/eps"] = eps.numpy()
            for k, v in st.items():
                data[f"{tag}/{mode}/state/{k}"] = v.numpy()
            data.update({f"{tag}/{mode}/mean": mean.numpy(), f"{tag}/{mode}/log_var": lv.numpy(), f"{tag}/{mode}/prior_mean": pmo.numpy(),
                         f"{tag}/{mode}/prior_log_var": plv.numpy(), f"{tag}/{mode}/alpha": al.numpy(), f"{tag}/{mode}/h_dec": hd.numpy()})
        print(name, tag, "h_dec[0,:3]", hd[0, :3].numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)


FSM_WORDFORMS = "dog\tdog,dogs\ncat\tcat,cats,kitten\nfire\tfire\nhydrant\thydrant,hydrants\nsalt\tsalt\n" \
                "and\tand\npepper\tpepper,peppers\nred\tred,reddish\nbird\tbird,birds,zzz_not_in_vocab\n"
FSM_CASES = [[], ["dog"], ["dog", "cat"], ["fire hydrant"], ["dog", "fire hydrant", "salt and pepper"],
             ["red", "dog", "red"], ["bird", "bird"], ["cat", "salt and pepper"]]


def fsm_fixture(name="g9_fsm"):
    """FiniteStateMachineBuilder.build (updown-baseline/updown/utils/constraints.py:328-361) on a toy vocabulary: the
    adjacency tensors (trimmed to the states in use, as the reference's collate does), the next-sub-state index and
    constraint2states for single-word, multi-word, repeated and out-of-vocabulary constraints."""
    import json
    import tempfile

    def mod(n, **attrs):
        m = types.ModuleType(n)
        m.__dict__.update(attrs)
        sys.modules[n] = m

    mod("anytree")                                     # only the ConstraintFilter uses it; bare symbols suffice to import
    mod("anytree.search", findall=lambda *a, **k: [])
    from updown.utils.constraints import FiniteStateMachineBuilder
    words = ["a", "the", "dog", "dogs", "cat", "cats", "kitten", "fire", "hydrant", "hydrants", "salt", "and", "pepper",
             "peppers", "red", "reddish", "bird", "birds", "on", "street"]
    vocab = ToyVocabulary(2)
    vocab._tokens += words
    vocab._index = {t: i for i, t in enumerate(vocab._tokens)}
    data = {"wordforms_tsv": np.array(FSM_WORDFORMS), "vocab_tokens": np.array(json.dumps(vocab._tokens))}
    with tempfile.TemporaryDirectory() as td:
        tsv = os.path.join(td, "wordforms.tsv")
        open(tsv, "w").write(FSM_WORDFORMS)
        for kmax in (3, 2):
            builder = FiniteStateMachineBuilder(vocab, tsv, None, max_given_constraints=kmax)
            for ci, cons in enumerate(FSM_CASES):
                if len(cons) > kmax:
                    continue
                fsm, nstates, c2s = builder.build(list(cons))
                key = f"k{kmax}/case{ci}"
                data[key + "/constraints"] = np.array(json.dumps(cons))
                data[key + "/nstates"] = np.array(nstates)
                data[key + "/fsm_bits"] = np.packbits(fsm[:nstates, :nstates, :].numpy().astype(np.uint8))
                data[key + "/constraint2states"] = np.array(json.dumps(c2s))
                print(name, key, cons, "states", nstates, c2s)
        # select_best_beam_with_constraints (updown-baseline/updown/utils/decoding.py:30-138), both branches, on the machines
        # of three-constraint inputs: objects with and without attribute constraints
        from updown.utils.decoding import select_best_beam_with_constraints
        builder = FiniteStateMachineBuilder(vocab, tsv, None, max_given_constraints=3)
        g = torch.Generator().manual_seed(5)
        sel_cases = [
            (["red", "dog", "cat"], [["dog", ["red"]], ["cat", []]], 2),
            (["red", "dog", "cat"], [["dog", ["red"]], ["cat", []]], 1),
            (["dog", "cat", "bird"], [["dog", []], ["cat", []], ["bird", []]], 2),
            (["red", "dog"], [["dog", ["red"]]], 2),
            (["dog"], [["dog", []]], 2),
        ]
        for si, (cons, cands, min_sat) in enumerate(sel_cases):
            fsm, nstates, c2s = builder.build(list(cons))
            S, beam, steps = nstates, 3, 6
            beams = torch.randint(0, len(vocab._tokens), (1, S, beam, steps), generator=g)
            lps = -torch.rand(1, S, beam, generator=g) * 10
            given = torch.tensor([len(cons)])
            for simple in (True, False):
                best, valid = select_best_beam_with_constraints(beams, lps, given, [cands], [c2s], min_sat, simple)
                key = f"sel/case{si}/simple{int(simple)}"
                data[key + "/best"] = best.numpy()
                data[key + "/valid"] = valid.numpy()
            key = f"sel/case{si}"
            data[key + "/constraints"] = np.array(json.dumps(cons))
            data[key + "/candidates"] = np.array(json.dumps(cands))
            data[key + "/constraint2states"] = np.array(json.dumps(c2s))
            data[key + "/min"] = np.array(min_sat)
            data[key + "/beams"] = beams.numpy()
            data[key + "/lps"] = lps.numpy()
            print(name, key, cons, cands, "best", best.tolist()[0][:3], "valid rows", valid.shape[1])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)


class Torch11:
    """Run the reference's UNMODIFIED search driver (updown-baseline/updown/modules/cbs.py) under torch 2.x by restoring the two
    torch-1.1 tensor semantics it relies on - the same rebinding pattern as EpsInjector, nothing in the reference is edited:
    * ``Tensor.masked_fill`` with a uint8 mask treats it as a boolean mask (cbs.py:134-136, :204-206);
    * ``/`` between an integer tensor and a Python int is floor division (cbs.py:231, ``backpointer = idx / per_node``)."""

    def __enter__(self):
        self._mf, self._td = torch.Tensor.masked_fill, torch.Tensor.__truediv__
        mf, td = self._mf, self._td

        def masked_fill(t, mask, value):
            return mf(t, mask.bool() if mask.dtype == torch.uint8 else mask, value)

        def truediv(t, other):
            if not t.is_floating_point() and isinstance(other, int):
                return torch.div(t, other, rounding_mode="floor")
            return td(t, other)

        torch.Tensor.masked_fill, torch.Tensor.__truediv__ = masked_fill, truediv
        return self

    def __exit__(self, *a):
        torch.Tensor.masked_fill, torch.Tensor.__truediv__ = self._mf, self._td


CBS_CASES = [  # (B, S, V, beam, per_node, max_steps, end_boost)   end_boost large -> every beam ends -> early stop (cbs.py:167)
    (2, 1, 50, 5, 2, 8, 0.0), (2, 3, 50, 3, 2, 8, 1.5), (3, 4, 97, 3, 2, 9, 1.5), (1, 4, 61, 5, 2, 7, 0.0),
    (2, 1, 40, 1, 1, 8, 1.5), (2, 3, 40, 1, 1, 6, 0.0), (1, 3, 300, 5, 1, 8, 0.0), (2, 1, 30, 3, 2, 12, 9.0),
    (1, 3, 30, 3, 2, 12, 9.0), (2, 4, 45, 5, 2, 20, 3.0),
]


def cbs_case_inputs(ci):
    B, S, V, beam, per_node, steps, boost = CBS_CASES[ci]
    g = torch.Generator().manual_seed(500 + ci)
    table = torch.randn(V, V, generator=g) * 2.0
    table[:, 1] += boost
    drift = torch.randn(7, V, generator=g) * 0.5
    if S == 1:
        fsm = torch.ones(B, 1, 1, V, dtype=torch.uint8)
    else:
        fsm = (torch.rand(B, S, S, V, generator=g) < 0.5).to(torch.uint8)
        fsm[:, :, :, 1] = 1  # @@BOUNDARY@@ is allowed on every transition (as in the reference's machines, constraints.py:300-320)
    return table, drift, fsm


def cbs_fixture(UpDownCaptioner, name="g12_cbs"):
    """(i) ``ConstrainedBeamSearch.search`` (updown-baseline/updown/modules/cbs.py:59-277), unmodified, under Torch11, driven by
    the table step above (5-tuple, as var_updown's patched driver expects at :127,:169) for S in {1,3,4}, beam in {1,3,5},
    per-node in {1,2}, early stop hit and not hit.  (ii) the reference ``UpDownCaptioner.forward`` eval branch
    (var_updown/var_updown/models/updown_captioner.py:324-366; tied 300-d subclass, use_cbs=True, B=1) end to end with injected
    ``eps`` and machines from the reference ``FiniteStateMachineBuilder``."""
    import json
    import tempfile
    from updown.modules.cbs import ConstrainedBeamSearch
    data = {"ncases": np.array(len(CBS_CASES))}
    for ci, (B, S, V, beam, per_node, steps, boost) in enumerate(CBS_CASES):
        table, drift, fsm = cbs_case_inputs(ci)
        inner = cbs_table_step(table, drift)
        calls = {"n": 0}

        def step5(tokens, state):
            calls["n"] += 1
            lp, st = inner(tokens, state)
            return lp, st, None, None, None
        search = ConstrainedBeamSearch(1, max_steps=steps, beam_size=beam, per_node_beam_size=per_node)
        with Torch11(), torch.no_grad():
            preds, lps = search.search(torch.full((B,), 1, dtype=torch.long), None, step5, fsm)
        key = f"search/case{ci}"
        data[key + "/dims"] = np.array([B, S, V, beam, per_node, steps])
        data[key + "/boost"] = np.array(boost)
        data[key + "/table"] = table.numpy()
        data[key + "/drift"] = drift.numpy()
        data[key + "/fsm_bits"] = np.packbits(fsm.numpy())
        data[key + "/predictions"] = preds.numpy()
        data[key + "/log_probs"] = lps.numpy()
        data[key + "/step_calls"] = np.array(calls["n"])
        print(name, key, (B, S, V, beam, per_node, steps), "step calls", calls["n"], "steps out", preds.shape[-1])

    # (ii) the captioner's eval branch
    def mod(n, **attrs):
        m = types.ModuleType(n)
        m.__dict__.update(attrs)
        sys.modules[n] = m
    if "anytree" not in sys.modules:
        mod("anytree")
        mod("anytree.search", findall=lambda *a, **k: [])
    from updown.utils.constraints import FiniteStateMachineBuilder
    words = ["a", "the", "dog", "dogs", "cat", "cats", "fire", "hydrant", "hydrants", "red", "reddish", "on", "street", "sits"]
    vocab = ToyVocabulary(2)
    vocab._tokens += words + [f"w{i}" for i in range(40)]
    vocab._index = {t: i for i, t in enumerate(vocab._tokens)}
    V = len(vocab._tokens)
    wordforms = "dog\tdog,dogs\ncat\tcat,cats\nfire\tfire\nhydrant\thydrant,hydrants\nred\tred,reddish\n"
    F, E, H, A, Z, L, R, beam = 48, 300, 32, 16, 8, 9, 5, 3
    data["eval/vocab_tokens"] = np.array(json.dumps(vocab._tokens))
    data["eval/wordforms_tsv"] = np.array(wordforms)
    data["eval/dims"] = np.array([V, E, H, A, F, Z, L, R, beam])

    class Tied(UpDownCaptioner):
        def _initialize_glove(self):
            t = torch.randn(self._vocabulary.get_vocab_size(), self.embedding_size, generator=torch.Generator().manual_seed(78)) * 0.3
            t[0] = 0
            return t
    eval_cases = [(["dog"], [["dog", []]], 1, 1), ([], [], 2, 1), (["dog", "cat"], [["dog", []], ["cat", []]], 2, 1),
                  (["fire hydrant", "dog"], [["fire hydrant", []], ["dog", []]], 2, -1),
                  (["dog", "cat", "red"], [["dog", ["red"]], ["cat", []]], 2, 0),
                  (["dog", "cat", "fire hydrant"], [["dog", []], ["cat", []], ["fire hydrant", []]], 3, 1)]
    data["eval/ncases"] = np.array(len(eval_cases))
    with tempfile.TemporaryDirectory() as td:
        tsv = os.path.join(td, "wordforms.tsv")
        open(tsv, "w").write(wordforms)
        builder = FiniteStateMachineBuilder(vocab, tsv, None, max_given_constraints=3)
        for ci, (cons, cands, min_sat, senti_v) in enumerate(eval_cases):
            torch.manual_seed(20 + ci)
            model = Tied(vocab, image_feature_size=F, embedding_size=E, hidden_size=H, attention_projection_size=A,
                         max_caption_length=L, beam_size=beam, use_cbs=True, min_constraints_to_satisfy=min_sat, z_space=Z,
                         prior_std=1.0, simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                         cbs_simple=True, device=torch.device("cpu"))
            model.eval()
            fsm_full, nstates, c2s = builder.build(list(cons))
            fsm = fsm_full[:nstates, :nstates, :].unsqueeze(0).contiguous()
            S = nstates
            g = torch.Generator().manual_seed(900 + ci)
            feats = torch.randn(1, R, F, generator=g)
            if ci == 2:
                feats[0, R - 1:] = 0
            senti = torch.tensor([[float(senti_v)]])
            eps = [torch.randn(1, Z, generator=g)] + [torch.randn(S * beam, Z, generator=g) for _ in range(L - 1)]
            seen = {}
            orig_search = model._beam_search.search

            def spy(*a, **k):
                out = orig_search(*a, **k)
                seen["beams"], seen["lps"] = out[0].clone(), out[1].clone()
                return out
            model._beam_search.search = spy
            with Torch11(), torch.no_grad(), EpsInjector(eps) as inj:
                out = model(feats.clone(), None, None, fsm=fsm, num_constraints=torch.tensor([len(cons)]),
                            constraints=[cands], constraint2states=[c2s], sentiment=senti)
            key = f"eval/case{ci}"
            for k2, v in state_dict_np(model).items():
                data[f"{key}/{k2}"] = v
            data[key + "/constraints"] = np.array(json.dumps(cons))
            data[key + "/candidates"] = np.array(json.dumps(cands))
            data[key + "/constraint2states"] = np.array(json.dumps(c2s))
            data[key + "/min"] = np.array(min_sat)
            data[key + "/nstates"] = np.array(S)
            data[key + "/fsm_bits"] = np.packbits(fsm.numpy())
            data[key + "/feats"] = feats.numpy()
            data[key + "/sentiment"] = senti.numpy()
            data[key + "/eps0"] = eps[0].numpy()
            data[key + "/eps_rest"] = torch.stack(eps[1:]).numpy()
            data[key + "/eps_used"] = np.array(inj.k)
            data[key + "/predictions"] = out["predictions"].numpy()
            data[key + "/beams"] = seen["beams"].numpy()
            data[key + "/log_probs"] = seen["lps"].numpy()
            print(name, key, cons, "S", S, "eps used", inj.k, "pred", [vocab._tokens[int(t)] for t in out["predictions"][0]])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)


FILTER_HIERARCHY = {"LabelName": "Entity", "Subcategory": [
    {"LabelName": "Animal", "Subcategory": [
        {"LabelName": "Carnivore", "Subcategory": [{"LabelName": "Dog"}, {"LabelName": "Cat"}, {"LabelName": "Bear"}]},
        {"LabelName": "Bird", "Subcategory": [{"LabelName": "Owl"}, {"LabelName": "Duck"}]},
        {"LabelName": "Mammal"}]},
    {"LabelName": "Vehicle", "Subcategory": [{"LabelName": "Land vehicle", "Subcategory": [{"LabelName": "Car"}, {"LabelName": "Truck"}]},
                                             {"LabelName": "Boat"}]},
    {"LabelName": "Furniture", "Subcategory": [{"LabelName": "Kitchen & dining room table"}, {"LabelName": "Chair"}]},
    {"LabelName": "Band-aid"}, {"LabelName": "Person"}, {"LabelName": "Luggage and bags"}]}
FILTER_CASES = [  # (boxes x1 y1 x2 y2, class names, scores)
    # identical boxes on a dog / carnivore / animal: as EXECUTED the reference keeps all three (the work list is sorted by ascending
    # height, so `heights[rest] >= heights[current]` holds for every pair and nothing is ever suppressed, constraints.py:195-203)
    ([[10, 10, 100, 100], [10, 10, 100, 100], [10, 10, 100, 100]], ["dog", "carnivore", "animal"], [0.9, 0.8, 0.7]),
    ([[10, 10, 100, 100], [12, 11, 101, 99], [200, 200, 300, 300], [0, 0, 50, 50]], ["carnivore", "dog", "car", "cat"], [0.95, 0.6, 0.8, 0.7]),
    # blacklist, padding boxes (score 0), replacements, duplicates, more boxes than k
    ([[0, 0, 10, 10], [20, 20, 40, 40], [50, 50, 90, 90], [0, 0, 0, 0], [5, 5, 15, 15]],
     ["person", "kitchen & dining room table", "band-aid", "dog", "mammal"], [0.99, 0.5, 0.6, 0.0, 0.9]),
    ([[0, 0, 10, 10], [20, 20, 40, 40], [50, 50, 90, 90], [60, 60, 95, 95], [1, 1, 9, 9]],
     ["dog", "dog", "owl", "bird", "luggage and bags"], [0.3, 0.9, 0.8, 0.85, 0.2]),
    ([[0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 10]],
     ["animal", "vehicle", "car", "duck", "chair"], [0.5, 0.5, 0.5, 0.5, 0.5]),
    ([], [], []),
    ([[0, 0, 10, 10]], ["person"], [0.9]),
    ([[3, 3, 30, 30], [3, 3, 30, 30]], ["truck", "land vehicle"], [0.4, 0.7]),
]


def filter_fixture(name="g13_filter"):
    """``ConstraintFilter.__call__`` + ``_nms`` (updown-baseline/updown/utils/constraints.py:105-209), unmodified, for k = 3 and 2.
    anytree is absent: the instance is made without ``__init__`` (which only builds the anytree tree, :105-121) and ``findall`` -
    the one anytree call of ``_nms`` (:163-166) - is bound to a walk over a flat pre-order node list with INJECTED heights (edges
    on the longest path to a leaf), so the fixture pins the executed filtering logic given the hierarchy's heights."""
    import json

    class Node:
        def __init__(self, label, height):
            self.LabelName, self.height = label, height

    nodes = []

    def walk(d):
        me = Node(d["LabelName"], 0)
        nodes.append(me)
        for c in d.get("Subcategory", []):
            me.height = max(me.height, walk(c) + 1)
        return me.height
    walk(FILTER_HIERARCHY)

    def mod(n, **attrs):
        m = types.ModuleType(n)
        m.__dict__.update(attrs)
        sys.modules[n] = m
    mod("anytree")
    mod("anytree.search", findall=lambda root, filter_: [n for n in nodes if filter_(n)])
    for m in [k for k in sys.modules if k.startswith("updown.utils.constraints")]:
        del sys.modules[m]
    from updown.utils.constraints import ConstraintFilter
    data = {"hierarchy": np.array(json.dumps(FILTER_HIERARCHY)), "ncases": np.array(len(FILTER_CASES)),
            "heights": np.array(json.dumps({n.LabelName.lower(): n.height for n in nodes}))}
    for k in (3, 2):
        f = object.__new__(ConstraintFilter)
        f._hierarchy, f._nms_threshold, f._max_given_constraints = None, 0.85, k
        for ci, (boxes, names, scores) in enumerate(FILTER_CASES):
            b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
            sc = np.asarray(scores, dtype=np.float64)
            kept = f(b, list(names), sc)
            keep_idx = f._nms(b, list(names)) if len(names) else []
            key = f"k{k}/case{ci}"
            data[key + "/boxes"], data[key + "/scores"] = b, sc
            data[key + "/names"] = np.array(json.dumps(list(names)))
            data[key + "/kept"] = np.array(json.dumps(sorted(kept)))
            data[key + "/nms_keep"] = np.asarray(keep_idx, dtype=np.int64)
            print(name, key, names, "->", sorted(kept), "nms keeps", list(map(int, keep_idx)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)


def sv2_train_fixture(name="g14_train_sv2"):
    """SENTIMENT_VAE = 2 as a TRAINABLE path.  The reference captioner cannot be constructed in this mode
    (updown_captioner.py:79,89), so the fixture drives the reference's own ``UpDownCell`` (var_updown/var_updown/modules/
    updown_cell.py:86-231, unmodified: attention-pooled ``obj_atts`` as prior mean :160-163 and as the 150-wide conditioning of
    both language LSTMs :185-188,219-222) through T teacher-forced steps under autograd, inside a thin loop that restates the
    captioner's training branch around it with plain torch modules (nn.Embedding / nn.Linear, the KL formula :298-303 on the
    prior mean the cell RETURNS, allennlp boundary tokens + masked CE via the stand-ins pinned by g1-g8).  Stored: weights,
    inputs, loss, kld, every gradient, states / alpha / mean / log_var / prior mean of steps 0, 1, T-1."""
    from var_updown.modules import UpDownCell
    V, E, H, A, F, Z, L, B, R = 120, 40, 48, 32, 64, 150, 6, 4, 5
    T = L + 1
    torch.manual_seed(11)
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    cell = UpDownCell(F, E, H, A, Z, 2, False, torch.device("cpu"), "glove")
    out = torch.nn.Linear(H, V)
    feats, caps, senti, eps = make_inputs(2024, B, R, F, L, V, Z, T, unk=True)
    g = torch.Generator().manual_seed(77)
    obj = torch.randn(B, R, 150, generator=g) * 0.4
    obj[1, R - 2:] = 0          # the zero-padded regions carry no attribute means
    obj[2, 1] = 0               # an object without attributes (translate_obj_atts2obj_means: zeros, :521-522)
    tokens, _ = oracle.add_sentence_boundary_token_ids(caps, caps != 0, 1, 1)
    mask = tokens != 0
    prior_mean = torch.zeros(B, Z)
    prior_var = (torch.ones(B, Z) * 0.9).pow(2)            # PRIOR_STD = 0.9
    states, logits, klds, steps = None, [], [], []
    with EpsInjector([eps[t] for t in range(T)]):
        for t in range(T):
            hd, states, mean, log_var, prior_mean, prior_log_var, alpha = cell(
                feats if t else feats.clone(), obj, emb(tokens[:, t]), states, True, None, None, prior_mean, prior_var)
            kld = 1 + log_var - prior_log_var - ((mean - prior_mean).pow(2) + log_var.exp()) / (prior_var + 0.00001)
            klds.append((-0.5 * kld.sum(1)).unsqueeze(1))
            logits.append(out(hd).unsqueeze(1))
            steps.append({**{k: v.detach().clone() for k, v in states.items()}, "alpha": alpha.detach().clone(),
                          "mean": mean.detach().clone(), "log_var": log_var.detach().clone(), "prior_mean": prior_mean.detach().clone()})
    logits = torch.cat(logits, 1)
    tmask = mask[:, 1:].contiguous()
    klds = torch.cat(klds, 1) * tmask.float()
    loss = tmask.sum(-1).float() * oracle.sequence_cross_entropy_with_logits(logits, tokens[:, 1:].contiguous(), tmask)
    kld = klds.sum(1)
    (loss.mean() + kld.mean() / 750.0).backward()
    data = {"param/_embedding_layer.weight": emb.weight.detach().numpy().copy(),
            "param/_output_layer.weight": out.weight.detach().numpy().copy(), "param/_output_layer.bias": out.bias.detach().numpy().copy(),
            "grad/_embedding_layer.weight": emb.weight.grad.numpy().copy(), "grad/_output_layer.weight": out.weight.grad.numpy().copy(),
            "grad/_output_layer.bias": out.bias.grad.numpy().copy()}
    for k, v in cell.state_dict().items():
        data["param/_updown_cell." + k] = v.numpy().copy()
    for n, p in cell.named_parameters():
        data["grad/_updown_cell." + n] = p.grad.numpy().copy()
    data.update({"in/feats": feats.numpy(), "in/caps": caps.numpy(), "in/eps": eps.numpy(), "in/obj_atts": obj.numpy(),
                 "out/loss": loss.detach().numpy(), "out/kld": kld.detach().numpy()})
    for t in (0, 1, T - 1):
        for k, v in steps[t].items():
            data[f"step{t}/{k}"] = v.numpy()
    cfg = dict(vocab_size=V, image_feature_size=F, embedding_size=E, hidden_size=H, attention_projection_size=A, z_space=Z,
               max_caption_length=L, sentiment_vae=2, simple_vae=False, prior_std=0.9, senti_prior_multip=1.0, tied=False)
    data["cfg"] = np.array(repr(cfg))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "loss", loss.detach().numpy(), "kld", kld.detach().numpy())


def sv2_module_fixture(UpDownCaptioner, name="g16_sv2_module"):
    """SENTIMENT_VAE = 2 above the kernel level, three parts:

    translate/*: the reference's own ``UpDownCaptioner.translate_obj_atts2obj_means`` (updown_captioner.py:509-532) run UNMODIFIED
      on an instance made without ``__init__`` (the constructor cannot run in this mode: it reads /path/to/*.pkl, :79), with the
      four attributes the method reads set by hand.  Tables: the reference's own data/wordform_swd_scores.json through the
      SentiWordNet rule of :83-86 (score[0] - score[2], z_space times), and a seeded 10-d table through the repeat rule of :79-81.
    eval_<latent>/*: the reference ``UpDownCell`` (updown_cell.py:86-231, training=False) for LATENT_EMBEDDING "glove" (150
      conditioning columns = the pooled attribute means) and "senti_word_net" (one column = their first entry, :171-172), driven
      through four eval decode steps from zero states inside a thin restatement of the captioner's eval ``_decode_step``
      (:371-455: embedding, cell, output layer, log-softmax) with plain torch modules; rows = 3 images x 4 beams (the images'
      obj_atts repeated per row, which is what the reference's broadcast does at batch size 1).
    train_swn/*: the training fixture of g14 (the reference cell under autograd inside the captioner's training branch) for
      LATENT_EMBEDDING "senti_word_net" (Z = 16)."""
    import json
    from var_updown.modules import UpDownCell
    data = {}
    # ---- translate ---------------------------------------------------------------------------------------------------------
    swn = json.load(open(os.path.join(REF, "data", "wordform_swd_scores.json")))
    words = sorted(swn)[:40] + ["standing", "sitting", "white", "black", "wooden"]
    swn = {w: swn[w] for w in words if w in swn}
    g = torch.Generator().manual_seed(5)
    tab10 = {w: torch.randn(10, generator=g).numpy().astype(np.float64) for w in list(swn)[:30]}
    obj_atts = [
        [["person", ["standing 0.9", "white 0.4", "not-in-table 0.8"]], ["dog", []], ["table", ["wooden 0.7"]]],
        [["cat", ["zzz 0.9"]], ["car", ["black 0.5", "white 0.5", "standing 0.2"]]],
        [["tree", [list(swn)[3] + " 0.3", list(swn)[7] + " 0.6"]], ["sky", [list(swn)[11]]], ["road", ["sitting 1.0"]], ["x", []]],
    ]
    for tag, Z, table, multip in (("swn", 16, {k: np.repeat(v[0] - v[2], 16) for k, v in swn.items()}, 1.0),
                                  ("glove", 150, {k: np.repeat(v, int(150 / 10)) for k, v in tab10.items()}, 0.5)):
        inst = object.__new__(UpDownCaptioner)
        inst.__dict__.update(mean_choice=table, z_space=Z, latent_embedding_multip=multip, device=torch.device("cpu"))
        out = UpDownCaptioner.translate_obj_atts2obj_means(inst, obj_atts)
        data[f"translate/{tag}/out"] = out.numpy()
        data[f"translate/{tag}/table_keys"] = np.array(json.dumps(sorted(table)))
        data[f"translate/{tag}/table_vals"] = np.stack([table[k] for k in sorted(table)])
        data[f"translate/{tag}/multip"] = np.array(multip)
    data["translate/obj_atts"] = np.array(json.dumps(obj_atts))
    data["translate/swn_scores"] = np.array(json.dumps(swn))
    data["translate/tab10_keys"] = np.array(json.dumps(sorted(tab10)))
    data["translate/tab10_vals"] = np.stack([tab10[k] for k in sorted(tab10)])
    # ---- eval decode steps ------------------------------------------------------------------------------------------------
    V, E, H, A, F, R, nimg, rpi, steps = 90, 40, 48, 32, 64, 5, 3, 4, 4
    for latent, Z in (("glove", 150), ("senti_word_net", 16)):
        torch.manual_seed(21)
        emb = torch.nn.Embedding(V, E, padding_idx=0)
        cell = UpDownCell(F, E, H, A, Z, 2, False, torch.device("cpu"), latent)
        out = torch.nn.Linear(H, V)
        g = torch.Generator().manual_seed(60 + Z)
        feats = torch.randn(nimg, R, F, generator=g)
        feats[1, R - 2:] = 0
        obj = torch.randn(nimg, R, Z, generator=g) * 0.4
        obj[1, R - 2:] = 0
        obj[2, 0] = 0
        G = nimg * rpi
        rows = torch.arange(G) // rpi                                  # batch-major rows, as the image features are repeated (:405-413)
        prior_var = (torch.ones(G, Z) * 0.8).pow(2)
        tag = "eval_" + latent
        for k, v in cell.state_dict().items():
            data[f"{tag}/param/_updown_cell.{k}"] = v.numpy().copy()
        data[f"{tag}/param/_embedding_layer.weight"] = emb.weight.detach().numpy().copy()
        data[f"{tag}/param/_output_layer.weight"] = out.weight.detach().numpy().copy()
        data[f"{tag}/param/_output_layer.bias"] = out.bias.detach().numpy().copy()
        data[f"{tag}/in/feats"], data[f"{tag}/in/obj_atts"] = feats.numpy(), obj.numpy()
        states = None
        with torch.no_grad():
            for t in range(steps):
                tok = torch.randint(1, V, (G,), generator=g)
                eps = torch.randn(G, Z, generator=g)
                with EpsInjector([eps]):
                    hd, states, mean, log_var, pm, plv, alpha = cell(feats[rows].clone(), obj[rows].clone(), emb(tok), states, False, None,
                                                                     None, torch.zeros(G, Z), prior_var.clone())
                lp = torch.log_softmax(out(hd), dim=1)
                data[f"{tag}/step{t}/tok"], data[f"{tag}/step{t}/eps"] = tok.numpy(), eps.numpy()
                data[f"{tag}/step{t}/log_probs"], data[f"{tag}/step{t}/alpha"] = lp.numpy(), alpha.numpy()
                data[f"{tag}/step{t}/prior_mean"], data[f"{tag}/step{t}/prior_log_var"] = pm.numpy(), plv.numpy()
                for k in ("h1", "c1", "h_decoder", "c_decoder"):
                    data[f"{tag}/step{t}/state/{k}"] = states[k].numpy().copy()
        print(name, tag, "lp[0,:3]", lp[0, :3].numpy())
    # ---- training, senti_word_net --------------------------------------------------------------------------------------------
    V, E, H, A, F, Z, L, B, R = 120, 40, 48, 32, 64, 16, 6, 4, 5
    T = L + 1
    torch.manual_seed(12)
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    cell = UpDownCell(F, E, H, A, Z, 2, False, torch.device("cpu"), "senti_word_net")
    out = torch.nn.Linear(H, V)
    feats, caps, senti, eps = make_inputs(2025, B, R, F, L, V, Z, T, unk=True)
    g = torch.Generator().manual_seed(78)
    obj = torch.randn(B, R, Z, generator=g) * 0.4
    obj[1, R - 2:] = 0
    obj[2, 1] = 0
    tokens, _ = oracle.add_sentence_boundary_token_ids(caps, caps != 0, 1, 1)
    mask = tokens != 0
    prior_mean = torch.zeros(B, Z)
    prior_var = (torch.ones(B, Z) * 0.9).pow(2)
    states, logits, klds = None, [], []
    with EpsInjector([eps[t] for t in range(T)]):
        for t in range(T):
            hd, states, mean, log_var, prior_mean, prior_log_var, alpha = cell(
                feats if t else feats.clone(), obj, emb(tokens[:, t]), states, True, None, None, prior_mean, prior_var)
            kld = 1 + log_var - prior_log_var - ((mean - prior_mean).pow(2) + log_var.exp()) / (prior_var + 0.00001)
            klds.append((-0.5 * kld.sum(1)).unsqueeze(1))
            logits.append(out(hd).unsqueeze(1))
    logits = torch.cat(logits, 1)
    tmask = mask[:, 1:].contiguous()
    klds = torch.cat(klds, 1) * tmask.float()
    loss = tmask.sum(-1).float() * oracle.sequence_cross_entropy_with_logits(logits, tokens[:, 1:].contiguous(), tmask)
    kld = klds.sum(1)
    (loss.mean() + kld.mean() / 750.0).backward()
    tag = "train_swn"
    data.update({f"{tag}/param/_embedding_layer.weight": emb.weight.detach().numpy().copy(),
                 f"{tag}/param/_output_layer.weight": out.weight.detach().numpy().copy(),
                 f"{tag}/param/_output_layer.bias": out.bias.detach().numpy().copy(),
                 f"{tag}/grad/_embedding_layer.weight": emb.weight.grad.numpy().copy(),
                 f"{tag}/grad/_output_layer.weight": out.weight.grad.numpy().copy(),
                 f"{tag}/grad/_output_layer.bias": out.bias.grad.numpy().copy()})
    for k, v in cell.state_dict().items():
        data[f"{tag}/param/_updown_cell." + k] = v.numpy().copy()
    for n, p in cell.named_parameters():
        data[f"{tag}/grad/_updown_cell." + n] = p.grad.numpy().copy()
    data.update({f"{tag}/in/feats": feats.numpy(), f"{tag}/in/caps": caps.numpy(), f"{tag}/in/eps": eps.numpy(),
                 f"{tag}/in/obj_atts": obj.numpy(), f"{tag}/out/loss": loss.detach().numpy(), f"{tag}/out/kld": kld.detach().numpy()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "train_swn loss", loss.detach().numpy(), "kld", kld.detach().numpy())


def main():
    UpDownCaptioner = import_reference()
    if "--only-sv2" in sys.argv:
        return sv2_train_fixture()
    if "--only-sv2-module" in sys.argv:
        return sv2_module_fixture(UpDownCaptioner)
    if "--only-decode-large" in sys.argv:
        return decode_large_fixture(UpDownCaptioner)
    filter_fixture()
    if "--only-filter" in sys.argv:
        return
    fsm_fixture()
    cbs_fixture(UpDownCaptioner)
    if "--only-cbs" in sys.argv:
        return
    cell_fixture()
    sv2_train_fixture()
    sv2_module_fixture(UpDownCaptioner)
    if "--full" in sys.argv or not os.path.exists(os.path.join(HERE, "g10_full_c2.npz")):
        full_size_fixture(UpDownCaptioner, "g10_full_c1", B=4, unk=False)      # BASELINE configs[0]: batch 4
        full_size_fixture(UpDownCaptioner, "g10_full_c2", B=64, unk=True)      # BASELINE configs[1]: batch 64, with in-caption UNK
    #        V    E   H   A   F   Z   L
    toy = (300, 40, 48, 32, 64, 16, 6)
    train_fixture(UpDownCaptioner, "g1_train_sv1", toy, sv=1)
    train_fixture(UpDownCaptioner, "g2_train_sv0", toy, sv=0)
    train_fixture(UpDownCaptioner, "g3_train_tied", (120, 300, 48, 32, 64, 16, 6), sv=1, tied=True)
    train_fixture(UpDownCaptioner, "g4_train_prior", toy, sv=1, prior_std=0.7, multip=0.5)
    train_fixture(UpDownCaptioner, "g4b_train_simple", toy, sv=1, simple_vae=True)
    decode_fixture(UpDownCaptioner, "g5_decode_sv1", toy, sv=1)
    decode_fixture(UpDownCaptioner, "g5b_decode_sv0", toy, sv=0)
    decode_large_fixture(UpDownCaptioner)
    sgd_fixture(UpDownCaptioner, "g6_sgd", toy, sv=1)
    # odd sizes: nothing a multiple of 4/16/64 (kernel tail paths)
    train_fixture(UpDownCaptioner, "g7_train_odd", (131, 37, 50, 27, 70, 13, 5), sv=1, B=5, R=7)
    # captions with @@UNKNOWN@@ (id 0) before their end
    train_fixture(UpDownCaptioner, "g8_train_unk", (300, 40, 48, 32, 64, 16, 8), sv=1, B=5, R=5, unk=True)
    train_fixture(UpDownCaptioner, "g8b_train_unk_sv0", (131, 37, 50, 27, 70, 13, 7), sv=0, B=4, R=6, unk=True)


if __name__ == "__main__":
    main()

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


def main():
    UpDownCaptioner = import_reference()
    fsm_fixture()
    cell_fixture()
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
    sgd_fixture(UpDownCaptioner, "g6_sgd", toy, sv=1)
    # odd sizes: nothing a multiple of 4/16/64 (kernel tail paths)
    train_fixture(UpDownCaptioner, "g7_train_odd", (131, 37, 50, 27, 70, 13, 5), sv=1, B=5, R=7)
    # captions with @@UNKNOWN@@ (id 0) before their end
    train_fixture(UpDownCaptioner, "g8_train_unk", (300, 40, 48, 32, 64, 16, 8), sv=1, B=5, R=5, unk=True)
    train_fixture(UpDownCaptioner, "g8b_train_unk_sv0", (131, 37, 50, 27, 70, 13, 7), sv=0, B=4, R=6, unk=True)


if __name__ == "__main__":
    main()

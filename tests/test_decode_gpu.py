"""GPU: constrained beam search bookkeeping (ssc_beam_*) and the whole eval forward against (a) the reference itself -
tests/golden/g12_cbs.npz holds the results of the UNMODIFIED updown/modules/cbs.py and of the reference captioner's eval
branch (generated under the torch-1.1 shim of tests/golden/make_golden.py::Torch11) - and (b) the CPU oracle, which
tests/test_oracle_cbs_cpu.py shows bit-equal to those fixtures."""
import pytest
import torch

import oracle
from gpuutil import dev, maxdiff
from ssc_runtime.decode import cbs_search
from test_module_gpu import build_model

pytestmark = pytest.mark.gpu


def make_fsm(B, S, V, seed, dense=True):
    g = torch.Generator().manual_seed(seed)
    if S == 1:
        return torch.ones(B, 1, 1, V, dtype=torch.uint8)
    fsm = (torch.rand(B, S, S, V, generator=g) < (0.6 if dense else 0.2)).to(torch.uint8)
    fsm[:, :, :, 1] = 1  # the end token is always allowed
    return fsm


@pytest.mark.parametrize("B,S,V,beam,per_node", [(2, 1, 50, 5, 2), (3, 4, 97, 3, 2), (1, 2, 300, 5, 5), (2, 1, 40, 1, 1)])
def test_cbs_bookkeeping_matches_oracle_with_synthetic_step(B, S, V, beam, per_node):
    """Same deterministic step function on both sides: log-probs are a fixed pseudo-random function of
    (previous token, a per-row state counter), so states must be gathered correctly for results to agree."""
    g = torch.Generator().manual_seed(5)
    table = torch.log_softmax(torch.randn(V, V, generator=g) * 2.0, dim=1)
    table[:, 1] += 1.5  # make the end token likely so early stopping and forced-end paths are exercised
    table = torch.log_softmax(table, dim=1)
    drift = torch.randn(7, V, generator=g) * 0.5
    fsm = make_fsm(B, S, V, 9)

    def make_step(tab, dr):
        def step(tokens, state):
            G = tokens.numel()
            cnt = torch.zeros(G, 1, device=tokens.device) if state is None else state["cnt"]
            acc = torch.zeros(G, 3, device=tokens.device) if state is None else state["acc"]
            lp = torch.log_softmax(tab[tokens] + dr[(cnt.long().view(-1) % 7)] + acc.sum(1, keepdim=True) * 0.01, dim=1)
            new = {"cnt": cnt + 1, "acc": acc + tokens.view(-1, 1).float() * torch.tensor([[1.0, 0.5, 0.25]], device=tokens.device) % 3.0}
            return lp, new
        return step

    start = torch.full((B,), 1, dtype=torch.long)
    want_p, want_lp = oracle.cbs_search(start, None, make_step(table, drift), fsm, end_index=1, max_steps=8, beam_size=beam,
                                        per_node_beam_size=per_node)
    got_p, got_lp = cbs_search(start.cuda(), None, make_step(table.cuda(), drift.cuda()), fsm.cuda(), 1, 8, beam, per_node)
    assert got_p.shape == want_p.shape
    assert maxdiff(got_lp, want_lp) < 1e-4
    finite = torch.isfinite(want_lp) & (want_lp > -1e19)
    assert torch.equal(got_p.cpu()[finite], want_p[finite])


@pytest.mark.parametrize("sv", [1, 0])
def test_eval_forward_matches_oracle(sv):
    cfg = oracle.OracleConfig(vocab_size=120, image_feature_size=64, embedding_size=40, hidden_size=48,
                              attention_projection_size=32, z_space=16, max_caption_length=7, sentiment_vae=sv,
                              senti_prior_multip=0.5, beam_size=3)
    params = oracle.init_params(cfg, seed=11)
    params["_output_layer.bias"][1] += 2.0  # make @@BOUNDARY@@ reachable
    g = torch.Generator().manual_seed(3)
    B, R, beam = 1, 6, 3
    feats = torch.randn(B, R, 64, generator=g)
    senti = torch.tensor([[1.0]])
    eps = [torch.randn(B, 16, generator=g)] + [torch.randn(B * beam, 16, generator=g) for _ in range(10)]
    fsm = torch.ones(B, 1, 1, 120, dtype=torch.uint8)
    want = oracle.eval_forward(params, cfg, feats, senti, fsm, torch.tensor([0]), eps, beam_size=beam)
    m = build_model(cfg, params, beam=beam)
    m.eval()
    m._eps_override = [e.clone() for e in eps]
    out = m(dev(feats), None, None, sentiment=dev(senti))
    assert torch.equal(out["predictions"].cpu(), want["predictions"])
    # greedy invariant: beam 1 == argmax chain
    m1 = build_model(cfg, params, beam=1)
    m1.eval()
    m1._eps_override = [e[:1].clone() for e in eps]
    o1 = m1(dev(feats), None, None, sentiment=dev(senti))["predictions"]
    w1 = oracle.eval_forward(params, cfg, feats, senti, fsm, torch.tensor([0]), [e[:1] for e in eps], beam_size=1)
    assert torch.equal(o1.cpu(), w1["predictions"])


def test_batched_diverse_decode_matches_oracle_per_sample():
    """ssc_runtime.inference.diverse_decode (images x latent samples as ONE beam search, early-stop check every 4th
    step + trimming) equals the oracle run with one batch entry per (image, sample)."""
    from ssc_runtime.inference import diverse_decode
    cfg = oracle.OracleConfig(vocab_size=90, image_feature_size=48, embedding_size=24, hidden_size=32,
                              attention_projection_size=16, z_space=8, max_caption_length=9, sentiment_vae=1,
                              senti_prior_multip=0.5, beam_size=3)
    params = oracle.init_params(cfg, seed=13)
    params["_output_layer.bias"][1] += 3.0  # captions end early: exercises the early-stop / trimming path
    g = torch.Generator().manual_seed(8)
    nimg, ns, beam, R = 2, 3, 3, 5
    feats = torch.randn(nimg, R, 48, generator=g)
    senti = torch.tensor([1.0, -1.0])
    B = nimg * ns
    eps = [torch.randn(B, 8, generator=g)] + [torch.randn(B * beam, 8, generator=g) for _ in range(12)]
    fsm = torch.ones(B, 1, 1, 90, dtype=torch.uint8)
    feats_rep = feats.unsqueeze(1).expand(nimg, ns, R, 48).reshape(B, R, 48)
    senti_rep = senti.view(nimg, 1).expand(nimg, ns).reshape(B, 1)
    want = oracle.eval_forward(params, cfg, feats_rep, senti_rep, fsm, torch.zeros(B, dtype=torch.long), eps, beam_size=beam)
    m = build_model(cfg, params, beam=beam)
    m.eval()
    m._engine()
    got, calls = diverse_decode(m._dec, dev(feats), dev(senti), ns, beam, cfg.max_caption_length, 1,
                                eps_steps=[e.clone() for e in eps])
    assert torch.equal(got.cpu().view(B, -1), want["predictions"])


def test_full_size_c4_decode_steps_two_kernel_paths_agree():
    """BASELINE configs[3] (C4) row counts at full width: two consecutive decode steps for 50 images x 20 latent samples x
    5 beams = 5000 rows (V=10k, H=1200, 36x2048 regions) on the default kernels (3xBF16, wave-specialised at this grid
    size) and in the exact-fp32-MFMA mode: log-probs, states and attention weights agree to fp32 level; log-probs are
    normalised; reruns are bit-identical."""
    from ssc_runtime import lib as L
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    V, F, E, H, A, Z, R = 10000, 2048, 1000, 1200, 768, 128, 36
    torch.manual_seed(2)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=20, beam_size=5, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(3)
    nimg, G = 50, 5000
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    ctx = dec.prepare(feats)
    tok = torch.randint(2, V, (G,), generator=g).cuda()
    tok2 = torch.randint(2, V, (G,), generator=g).cuda()
    sent = torch.ones(G).cuda()
    eps1, eps2 = torch.randn(G, Z, generator=g).cuda(), torch.randn(G, Z, generator=g).cuda()

    def two_steps():
        lp1, st1, a1 = dec.step(ctx, tok, None, sent, eps1)
        lp2, st2, a2 = dec.step(ctx, tok2, st1, sent, eps2)
        return lp1.clone(), lp2.clone(), {k: v.clone() for k, v in st2.items()}, a2.clone()

    lib = L.load()
    x = two_steps()
    y = two_steps()
    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[3], y[3])
    lib.ssc_set_gemm_mode(0)
    try:
        z = two_steps()
    finally:
        lib.ssc_set_gemm_mode(1)
    assert maxdiff(x[0], z[0]) < 1e-4 and maxdiff(x[1], z[1]) < 1e-4 and maxdiff(x[3], z[3]) < 1e-5
    for k in x[2]:
        assert maxdiff(x[2][k], z[2][k]) < 1e-4, k
    assert maxdiff(torch.logsumexp(x[1], 1), torch.zeros(G)) < 1e-4


@pytest.mark.parametrize("B,S,V,beam,per_node", [(3, 1, 50, 3, 2), (2, 3, 200, 2, 2), (2, 2, 20000, 3, 1), (2, 2, 10000, 3, 2),
                                                 (1, 2, 777, 2, 2), (2, 1, 10240, 2, 2)])
def test_beam_selection_from_raw_logits_is_bit_identical(B, S, V, beam, per_node):
    """ssc_beam_first_logits / ssc_beam_step_logits (log-sum-exp inside the selection kernel; the V = 20000 case does not
    fit the LDS staging and reads the row from HBM) against ssc_log_softmax followed by ssc_beam_first / ssc_beam_step; for
    V <= 10240 the later steps run with the row in registers (beam_row_topk_reg_kernel) - compared with the LDS-staged kernel
    through the diagnostics switch."""
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(B * 100 + V)
    dev_ = "cuda"
    fsm = (torch.rand(B, S, S, V, generator=g) < 0.8).to(torch.uint8).to(dev_)
    fsm[:, :, :, :4] = 1
    logits0 = (torch.randn(B, V, generator=g) * 3).to(dev_)
    lp0 = torch.empty_like(logits0)
    lib.ssc_log_softmax(L.ptr(logits0), V, B, V, L.ptr(lp0), V, L.stream_ptr())
    outs = []
    for fn, src in ((lib.ssc_beam_first, lp0), (lib.ssc_beam_first_logits, logits0)):
        pred = torch.empty(B, S * beam, dtype=torch.int64, device=dev_)
        lpo = torch.empty(B, S, beam, device=dev_)
        fn(L.ptr(src), V, L.ptr(fsm), B, S, V, beam, L.ptr(pred), L.ptr(lpo), L.stream_ptr())
        outs.append((pred.clone(), lpo.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    G = B * S * beam
    logits = (torch.randn(G, V, generator=g) * 3).to(dev_)
    lp = torch.empty_like(logits)
    lib.ssc_log_softmax(L.ptr(logits), V, G, V, L.ptr(lp), V, L.stream_ptr())
    last = outs[0][0].reshape(G).clone()
    last[1] = 1                      # one ended beam (end_index = 1)
    last_lp = outs[0][1]
    res = []
    try:
        for reg in (1, 0):
            lib.ssc_debug_set(b"beam_reg", reg)
            for fn, src in ((lib.ssc_beam_step, lp), (lib.ssc_beam_step_logits, logits)):
                pred = torch.empty(B, S * beam, dtype=torch.int64, device=dev_)
                nlp = torch.empty(B, S, beam, device=dev_)
                back = torch.empty(B, S * beam, dtype=torch.int64, device=dev_)
                sval = torch.empty(B * S * S * beam * per_node, device=dev_)
                sidx = torch.empty(B * S * S * beam * per_node, dtype=torch.int64, device=dev_)
                fn(L.ptr(src), V, L.ptr(fsm), L.ptr(last), L.ptr(last_lp), B, S, V, beam, per_node, 1, L.ptr(pred), L.ptr(nlp),
                   L.ptr(back), L.ptr(sval), L.ptr(sidx), L.stream_ptr())
                res.append((pred.clone(), nlp.clone(), back.clone(), sval.clone(), sidx.clone()))
    finally:
        lib.ssc_debug_set(b"beam_reg", 1)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_trivial_fsm_none_equals_all_ones_mask():
    """cbs_search(fsm=None) - the trivial one-state machine, no mask read on the device - gives exactly the result of an
    explicit all-ones (B,1,1,V) mask."""
    from ssc_runtime.decode import cbs_search
    B, V, beam, steps = 4, 300, 3, 6
    g = torch.Generator().manual_seed(21)
    table = (torch.randn(V, V, generator=g) * 2).cuda()     # next-token logits as a function of the previous token

    def step(tokens, state):
        return torch.log_softmax(table[tokens], dim=1), {"h": torch.zeros(tokens.numel(), 2, device="cuda")}

    start = torch.full((B,), 1, dtype=torch.long, device="cuda")
    ones = torch.ones(B, 1, 1, V, dtype=torch.uint8, device="cuda")
    a, alp = cbs_search(start, None, step, ones, 1, steps, beam, 2)
    b, blp = cbs_search(start, None, step, None, 1, steps, beam, 2)
    assert torch.equal(a, b) and torch.equal(alp, blp)


# ---- the reference's own results (g12_cbs) ---------------------------------------------------------------------------------
from goldenlib import cbs_table_step, load_raw, unpack_fsm  # noqa: E402
from test_oracle_cbs_cpu import g12_eval_case  # noqa: E402

_G12 = load_raw("g12_cbs")


@pytest.mark.parametrize("ci", range(int(_G12["ncases"])))
def test_cbs_search_hip_equals_reference_fixture(ci):
    """ssc_beam_first/step/backtrace driven by ssc_runtime.decode.cbs_search == ConstrainedBeamSearch.search (cbs.py:59-277,
    run unmodified): predictions bit-equal wherever the beam is reachable (a beam whose log-prob is -inf / <= -1e19 holds an
    arbitrary tie among forbidden tokens), log-probs to 1e-4 (the table step's log_softmax runs on the device here)."""
    key = f"search/case{ci}"
    B, S, V, beam, per_node, steps = (int(x) for x in _G12[key + "/dims"])
    table, drift = torch.from_numpy(_G12[key + "/table"]).cuda(), torch.from_numpy(_G12[key + "/drift"]).cuda()
    fsm = unpack_fsm(_G12[key + "/fsm_bits"], B, S, V)
    inner, calls = cbs_table_step(table, drift), {"n": 0}

    def step(tokens, state):
        calls["n"] += 1
        return inner(tokens, state)
    want_p, want_lp = torch.from_numpy(_G12[key + "/predictions"]), torch.from_numpy(_G12[key + "/log_probs"])
    got_p, got_lp = cbs_search(torch.full((B,), 1, dtype=torch.long, device="cuda"), None, step, fsm.cuda(), 1, steps, beam, per_node)
    assert got_p.shape == want_p.shape                      # same number of steps: the early stop fired at the same step
    assert calls["n"] == int(_G12[key + "/step_calls"])
    finite = torch.isfinite(want_lp) & (want_lp > -1e19)
    assert maxdiff(got_lp.cpu()[finite], want_lp[finite]) < 1e-4
    assert torch.equal(got_p.cpu()[finite], want_p[finite])


@pytest.mark.parametrize("ci", range(int(_G12["eval/ncases"])))
def test_eval_forward_hip_equals_reference_fixture(ci):
    """The drop-in captioner's eval forward (HIP decode steps + device-side CBS + host-side beam selection) == the reference
    UpDownCaptioner.forward eval branch (updown_captioner.py:324-366) on the same weights, features, injected eps and
    reference-built machine: identical caption."""
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    c = g12_eval_case(ci)
    cfg = c["cfg"]

    class Tied(UpDownCaptioner):   # the table comes from the state dict
        def _initialize_glove(self):
            return torch.zeros(self._vocabulary.get_vocab_size(), self.embedding_size)

    m = Tied(Vocabulary(c["vocab"]), cfg.image_feature_size, cfg.embedding_size, cfg.hidden_size, cfg.attention_projection_size,
             max_caption_length=cfg.max_caption_length, beam_size=c["beam"], use_cbs=True, min_constraints_to_satisfy=c["min_sat"],
             z_space=cfg.z_space, prior_std=1.0, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
             cbs_simple=True, device=torch.device("cuda"))
    sd = dict(c["params"])
    sd["_output_layer.weight"] = sd["_embedding_layer.weight"]
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m._eps_override = [e.clone() for e in c["eps"]]
    out = m(dev(c["feats"]), None, None, sentiment=dev(c["senti"]), fsm=c["fsm"], num_constraints=torch.tensor([len(c["constraints"])]),
            constraints=[c["candidates"]], constraint2states=[c["c2s"]])
    assert torch.equal(out["predictions"].cpu(), c["want"])


# ---- constrained beam search with REAL finite state machines -------------------------------------------------
def _cbs_setup(tmp_path, kmax):
    """Toy captioner whose vocabulary holds the constraint word forms, and the FSM builder over it."""
    from ssc_runtime.constraints import FiniteStateMachineBuilder
    from ssc_runtime.vocab import Vocabulary
    words = ["a", "the", "dog", "dogs", "cat", "cats", "fire", "hydrant", "hydrants", "red", "reddish", "on", "street", "sits"]
    tsv = tmp_path / "wordforms.tsv"
    tsv.write_text("dog\tdog,dogs\ncat\tcat,cats\nfire\tfire\nhydrant\thydrant,hydrants\nred\tred,reddish\n")
    vocab = Vocabulary(["@@UNKNOWN@@", "@@BOUNDARY@@"] + words + [f"w{i}" for i in range(40)])
    return vocab, FiniteStateMachineBuilder(vocab, str(tsv), None, max_given_constraints=kmax)


@pytest.mark.parametrize("constraints,min_sat", [(["dog"], 1), (["dog", "cat"], 2), (["fire hydrant", "dog"], 2),
                                                 (["dog", "cat", "red"], 2), (["dog", "cat", "fire hydrant"], 3)])
def test_eval_forward_with_built_fsm_matches_oracle(tmp_path, constraints, min_sat):
    """k = 1..3 constraints (single- and multi-word): the machine comes from ssc_runtime.constraints (bit-identical to the
    reference builder, tests/test_constraints_cpu.py), the search runs on the device (ssc_beam_*), the best
    constraint-satisfying beam is selected on the host; predictions equal the oracle's, and the constraint words appear."""
    vocab, builder = _cbs_setup(tmp_path, 3)
    V = vocab.get_vocab_size()
    cfg = oracle.OracleConfig(vocab_size=V, image_feature_size=48, embedding_size=300, hidden_size=32,
                              attention_projection_size=16, z_space=8, max_caption_length=9, sentiment_vae=1,
                              senti_prior_multip=0.5, tied=True, beam_size=3)
    params = oracle.init_params(cfg, seed=31)
    g = torch.Generator().manual_seed(12)
    B, R, beam = 1, 5, 3
    feats = torch.randn(B, R, 48, generator=g)
    senti = torch.tensor([[1.0]])
    fsm, nstates, c2s = builder.build_trimmed(constraints)
    S = nstates
    eps = [torch.randn(B, 8, generator=g)] + [torch.randn(B * S * beam, 8, generator=g) for _ in range(10)]
    k = torch.tensor([len(constraints)])
    want = oracle.eval_forward(params, cfg, feats, senti, fsm, k, eps, beam_size=beam, min_constraints_to_satisfy=min_sat)
    from var_updown.models import UpDownCaptioner

    class Tied(UpDownCaptioner):   # USE_CBS needs the frozen, output-tied 300-d embedding; the table comes from the state dict
        def _initialize_glove(self):
            return torch.zeros(self._vocabulary.get_vocab_size(), self.embedding_size)

    m = Tied(vocab, 48, 300, 32, 16, max_caption_length=9, beam_size=beam, use_cbs=True,
                        min_constraints_to_satisfy=min_sat, z_space=8, prior_std=1.0, latent_embedding="glove",
                        sentiment_vae=1, senti_prior_multip=0.5, cbs_simple=True, device=torch.device("cuda"))
    sd = dict(params)
    sd["_output_layer.weight"] = sd["_embedding_layer.weight"]
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m._eps_override = [e.clone() for e in eps]
    out = m(dev(feats), None, None, sentiment=dev(senti), fsm=fsm, num_constraints=k, constraints=None, constraint2states=None)
    assert torch.equal(out["predictions"].cpu(), want["predictions"])
    # the selected beam satisfies at least min(k, min_sat) constraints: count constraints whose words appear in order
    toks = [vocab.get_token_from_index(int(t)) for t in out["predictions"][0].cpu()]
    forms = {"dog": {"dog", "dogs"}, "cat": {"cat", "cats"}, "red": {"red", "reddish"}}

    def satisfied(c):
        ws = c.split()
        if len(ws) == 1:
            return any(t in forms[c] for t in toks)
        return any(toks[i] == "fire" and toks[i + 1] in ("hydrant", "hydrants") for i in range(len(toks) - 1))
    assert sum(satisfied(c) for c in constraints) >= min(len(constraints), min_sat), toks


def test_standalone_attention_module_forward_matches_float64():
    """BottomUpTopDownAttention.forward (attention.py:36-97) as a callable module on the HIP path."""
    from var_updown.modules.attention import BottomUpTopDownAttention
    torch.manual_seed(4)
    att = BottomUpTopDownAttention(40, 64, 24).cuda()
    g = torch.Generator().manual_seed(6)
    B, R = 5, 7
    q = torch.randn(B, 40, generator=g)
    feats = torch.randn(B, R, 64, generator=g)
    mask = torch.ones(B, R)
    mask[1, 5:] = 0
    mask[3, 2:] = 0
    wq, wv, wa = (p.detach().cpu().double() for p in (att._query_vector_projection_layer.weight,
                                                       att._image_features_projection_layer.weight, att._attention_layer.weight))

    def ref(m):
        logits = (torch.tanh((q.double() @ wq.t())[:, None, :] + feats.double() @ wv.t()) @ wa.t()).squeeze(-1)
        if m is None:
            return torch.softmax(logits, -1)
        p = torch.softmax(logits * m.double(), -1) * m.double()
        return p / (p.sum(-1, keepdim=True) + 1e-13)
    got = att(q.cuda(), feats.cuda(), mask.cuda())
    assert got.shape == (B, R) and maxdiff(got, ref(mask)) < 1e-5
    assert float(got[1, 5:].abs().max()) == 0.0
    assert maxdiff(att(q.cuda(), feats.cuda()), ref(None)) < 1e-5


def test_decode_step_token_table_equals_embedding_segment():
    """From 8 images per call on ssc_decode_prepare forms the embedding's gate contribution for the whole vocabulary once and the
    attention-LSTM cell picks the row of each beam's last token (ssc_lstm_fwd_desc.add0_rows); below that the embedding is a
    K = E segment of the gate product (updown_captioner.py:430, updown_cell.py:143-148).  The same 8 images decoded as one call
    (table) and as two calls of 4 (segment) give the same log-probs / states / attention weights to fp32 level."""
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    V, F, E, H, A, Z, R = 300, 64, 40, 48, 24, 8, 5
    torch.manual_seed(5)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=8, beam_size=3, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(6)
    nimg, rpi = 8, 6
    G = nimg * rpi
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    tok, tok2 = torch.randint(0, V, (G,), generator=g).cuda(), torch.randint(0, V, (G,), generator=g).cuda()
    sent = torch.randint(-1, 2, (G,), generator=g).float().cuda()
    eps1, eps2 = torch.randn(G, Z, generator=g).cuda(), torch.randn(G, Z, generator=g).cuda()

    def run(lo, hi):
        ctx = dec.prepare(feats[lo:hi].contiguous())
        r = slice(lo * rpi, hi * rpi)
        lp1, st1, a1 = dec.step(ctx, tok[r].contiguous(), None, sent[r].contiguous(), eps1[r].contiguous())
        lp2, st2, a2 = dec.step(ctx, tok2[r].contiguous(), st1, sent[r].contiguous(), eps2[r].contiguous())
        return lp2.clone(), {k: v.clone() for k, v in st2.items()}, a2.clone()

    whole = run(0, 8)                       # table
    halves = [run(0, 4), run(4, 8)]         # segment
    lp = torch.cat([h[0] for h in halves])
    al = torch.cat([h[2] for h in halves])
    assert maxdiff(whole[0], lp) < 2e-5 and maxdiff(whole[2], al) < 1e-6
    for k in whole[1]:
        assert maxdiff(whole[1][k], torch.cat([h[1][k] for h in halves])) < 2e-6, k
    # a caller that hands token EMBEDDINGS (UpDownCell.forward) overrides p->emb for the call: the table of the image context
    # (built from the model's own embedding) must not be used then
    ctx = dec.prepare(feats)
    _, st_ids, a_ids = dec.step(ctx, tok, None, sent, eps1, want_log_probs=False)
    emb_w = dict(m.named_parameters())["_embedding_layer.weight"].detach()
    _, st_emb, a_emb = dec._step_from_embedding(ctx, emb_w[tok], None, sent, eps1)
    assert maxdiff(a_ids, a_emb) < 1e-6
    for k in ("h1", "c1", "h_decoder", "c_decoder"):
        assert maxdiff(st_ids[k], st_emb[k]) < 2e-6, k


def test_decode_step_attended_feature_table_equals_the_feature_segment():
    """ssc_decode_prepare forms P[img, r, :] = W_ih^dec[:, :F] v_{img,r} once per image and the decoder cell contracts it with the
    step's attention weights (ssc_lstm_fwd_img) instead of running att = sum_r alpha_r v_r through a K = F segment of the gate
    product (updown_cell.py:156-158,211-229): the same value by linearity.  Two consecutive steps with the table (default) and
    without it (ssc_debug_set("dec_att_table", 0)) agree to fp32 level: log-probs, states, attention weights; rows of one image
    that are not a multiple of the kernel's 16-row chunk, a zero-padded region, R not a multiple of 4."""
    import ctypes as C
    from ssc_runtime import lib as L
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    lib = L.load()
    V, F, E, H, A, Z, R = 300, 64, 40, 48, 24, 8, 7
    torch.manual_seed(5)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=8, beam_size=3, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(9)
    nimg, rpi = 5, 21
    G = nimg * rpi
    feats = torch.randn(nimg, R, F, generator=g)
    feats[2, R - 2:] = 0
    feats = feats.cuda()
    tok, tok2 = torch.randint(0, V, (G,), generator=g).cuda(), torch.randint(0, V, (G,), generator=g).cuda()
    sent = torch.randint(-1, 2, (G,), generator=g).float().cuda()
    eps1, eps2 = torch.randn(G, Z, generator=g).cuda(), torch.randn(G, Z, generator=g).cuda()

    def run(table):
        lib.ssc_debug_set(b"dec_att_table", table)
        dec.ATT_TABLE_MIN_ROWS = 1   # (the table path is a large-call optimisation: force it at this toy size)
        try:
            ctx = dec.prepare(feats)
            lp1, st1, a1 = dec.step(ctx, tok, None, sent, eps1)
            lp2, st2, a2 = dec.step(ctx, tok2, st1, sent, eps2)
            return lp2.clone(), {k: v.clone() for k, v in st2.items()}, a2.clone()
        finally:
            lib.ssc_debug_set(b"dec_att_table", 1)
            del dec.ATT_TABLE_MIN_ROWS

    with_table, without = run(1), run(0)
    assert maxdiff(with_table[0], without[0]) < 2e-5 and maxdiff(with_table[2], without[2]) < 1e-6
    for k in with_table[1]:
        assert maxdiff(with_table[1][k], without[1][k]) < 2e-6, k


def test_sibling_dedup_and_table_paths_leave_the_search_unchanged():
    """A whole diverse decode at a size where the large-call paths engage (8 images x 16 samples x beam 5 = 640 rows per step:
    per-token gate table, per-image attended-feature table, products of the parent's states formed on the distinct parents of each
    beam group - ssc_decode_step_desc.parent) against the same decode with both switched off: identical captions, and one step's
    log-probs / states from re-ordered states agree to fp32 level."""
    from ssc_runtime import lib as L
    from ssc_runtime.inference import diverse_decode
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    lib = L.load()
    V, F, E, H, A, Z, R = 400, 64, 40, 64, 24, 8, 9
    torch.manual_seed(7)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=7, beam_size=5, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(3)
    nimg, ns, beam = 8, 16, 5
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    senti = torch.tensor([1.0, -1.0, 0.0, 1.0, 1.0, -1.0, 0.0, 0.0]).cuda()
    B = nimg * ns
    eps = [torch.randn(B, Z, generator=g).cuda()] + [torch.randn(B * beam, Z, generator=g).cuda() for _ in range(8)]

    def run(on):
        lib.ssc_debug_set(b"dec_dedup", on)
        lib.ssc_debug_set(b"dec_att_table", on)
        try:
            return diverse_decode(dec, feats, senti, ns, beam, 7, 1, eps_steps=[e.clone() for e in eps], early_stop=False)[0].clone()
        finally:
            lib.ssc_debug_set(b"dec_dedup", 1)
            lib.ssc_debug_set(b"dec_att_table", 1)

    assert torch.equal(run(1), run(0))
    # one step from re-ordered states: groups of 5 rows whose members 0, 2 and 1, 3 share a parent
    G = B * beam
    ctx = dec.prepare(feats)
    tok = torch.randint(2, V, (G,), generator=g).cuda()
    base = {k: torch.randn(B, 5, H, generator=g).cuda() * 0.3 for k in ("h1", "c1", "h_decoder", "c_decoder")}
    parent = torch.tensor([0, 3, 0, 3, 4]).repeat(B, 1).cuda()
    st = {k: v.gather(1, parent.view(B, 5, 1).expand(B, 5, H)).reshape(G, H).contiguous() for k, v in base.items()}
    sent_rows = senti.view(nimg, 1).expand(nimg, G // nimg).reshape(G).contiguous()
    outs = []
    for on in (1, 0):
        lib.ssc_debug_set(b"dec_dedup", on)
        try:
            s_in = dict(st)
            s_in["_parent"] = parent
            lp, so, al = dec.step(ctx, tok, s_in, sent_rows, eps[1])
            outs.append((lp.clone(), {k: v.clone() for k, v in so.items()}, al.clone()))
        finally:
            lib.ssc_debug_set(b"dec_dedup", 1)
    assert maxdiff(outs[0][0], outs[1][0]) < 2e-5 and maxdiff(outs[0][2], outs[1][2]) < 1e-6
    for k in ("h1", "c1", "h_decoder", "c_decoder"):
        assert maxdiff(outs[0][1][k], outs[1][1][k]) < 2e-6, k
    # the same step from the states in the PREVIOUS step's row order (no re-ordering by the caller: "_ungathered"): every reader
    # goes through the parent lists - the very same numbers
    assert dec.ungathered_ok(ctx, G, 5) and not dec.ungathered_ok(ctx, G, 1)
    s_in = {k: v.reshape(G, H).contiguous() for k, v in base.items()}
    s_in["_parent"] = parent
    s_in["_ungathered"] = True
    lp, so, al = dec.step(ctx, tok, s_in, sent_rows, eps[1])
    assert torch.equal(lp, outs[0][0]) and torch.equal(al, outs[0][2])
    for k in ("h1", "c1", "h_decoder", "c_decoder"):
        assert torch.equal(so[k], outs[0][1][k]), k
    # ... and a step that cannot read them that way refuses them
    lib.ssc_debug_set(b"dec_dedup", 0)
    try:
        with pytest.raises(ValueError):
            dec.step(ctx, tok, s_in, sent_rows, eps[1])
    finally:
        lib.ssc_debug_set(b"dec_dedup", 1)


def test_search_without_state_reordering_equals_search_with_it():
    """cbs_search leaves the states in the previous step's row order when the step function reads them through the parent lists
    (DecodeEngine.step at >= 512 rows); ssc_debug_set("dec_ungathered", 0) restores the four ssc_gather_rows per step: identical
    beams."""
    from ssc_runtime import lib as L
    from ssc_runtime.inference import diverse_decode
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    lib = L.load()
    V, F, E, H, A, Z, R = 300, 48, 40, 64, 24, 8, 7
    torch.manual_seed(11)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=8, beam_size=5, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(5)
    nimg, ns, beam = 9, 17, 5
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    B = nimg * ns
    eps = [torch.randn(B, Z, generator=g).cuda()] + [torch.randn(B * beam, Z, generator=g).cuda() for _ in range(9)]
    res = []
    for on in (1, 0):
        lib.ssc_debug_set(b"dec_ungathered", on)
        try:
            res.append(diverse_decode(dec, feats, senti, ns, beam, 8, 1, eps_steps=[e.clone() for e in eps], early_stop=False)[0].clone())
        finally:
            lib.ssc_debug_set(b"dec_ungathered", 1)
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("nimg,rpi,R,H,variant", [
    (3, 37, 36, 72, "dedup"),      # KS = 10, a ragged last 16-row chunk, a ragged last unit block (72 = 4 x 16 + 8)
    (2, 16, 50, 64, "dedup"),      # KS = 17
    (2, 20, 100, 48, "plain"),     # KS = 33 (84 KB of LDS)
    (4, 19, 36, 64, "plain"),
    (2, 33, 36, 64, "general"),    # every option of the descriptor (mode 2)
    (2, 18, 9, 50, "general"),     # H % 4 != 0: the VALU form
    (3, 21, 36, 64, "valu"),       # the VALU form forced by its switch
    (50, 100, 36, 1200, "dedup"),  # C4's decode step at full size: 5000 rows, 75 unit blocks
])
def test_image_cell_kernel_equals_float64(nimg, rpi, R, H, variant):
    """ssc_lstm_fwd_img - the decoder cell with the attended-feature term contracted from the per-image table
    (updown_cell.py:156-158,211-229) - against a float64 evaluation: the matrix-core kernel in its three modes (plain; sibling
    dedup = second slab through slab2_rows + previous cell state through c_prev_rows; every option), every k-step count
    (R <= 38 / 66 / 128), ragged rows and units, the sentiment column, and the VALU form."""
    import ctypes as C
    from ssc_runtime import lib as L
    lib = L.load()
    dev_ = "cuda"
    G, H4 = nimg * rpi, 4 * H
    g = torch.Generator().manual_seed(nimg * 1000 + R * 10 + H)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev_)
    slabs, P = rnd(2, G, H4) * 0.3, rnd(nimg, R, H4) * 0.3
    alpha = torch.softmax(rnd(G, R), dim=1).contiguous()
    cprev, b_ih, b_hh, wcol = rnd(G, H), rnd(H4) * 0.1, rnd(H4) * 0.1, rnd(H4) * 0.2
    sent = torch.randint(-1, 2, (G,), generator=g).float().to(dev_)
    c_out, h_out = torch.empty(G, H, device=dev_), torch.empty(G, H, device=dev_)
    f = L.LstmFwdDesc()
    f.B, f.H = G, H
    f.slabs, f.nslab, f.slab_stride = slabs.data_ptr(), 1, G * H4
    f.b_ih, f.b_hh = b_ih.data_ptr(), b_hh.data_ptr()
    f.sent, f.wcol, f.ldwcol = sent.data_ptr(), wcol.data_ptr(), 1
    f.c_prev, f.ld_cprev = cprev.data_ptr(), H
    f.c_out, f.ld_cout, f.h_out, f.ld_hout = c_out.data_ptr(), H, h_out.data_ptr(), H
    pre = slabs[0].double() + b_ih.double() + b_hh.double() + sent.double()[:, None] * wcol.double() \
        + torch.bmm(alpha.double().view(nimg, rpi, R), P.double()).view(G, H4)
    cp = cprev.double()
    keep = []
    if variant == "dedup":
        nu = max(G // 3, 1)
        slabs2 = rnd(nu, H4) * 0.3
        slot = torch.randint(0, nu, (G,), generator=g).int().to(dev_)
        prow = torch.randint(0, G, (G,), generator=g).int().to(dev_)
        f.slabs2, f.nslab2, f.slab2_stride, f.slab2_rows, f.c_prev_rows = slabs2.data_ptr(), 1, G * H4, slot.data_ptr(), prow.data_ptr()
        pre = pre + slabs2.double()[slot.long()]
        cp = cprev.double()[prow.long()]
        keep += [slabs2, slot, prow]
    elif variant == "general":
        f.nslab = 2                                                # a second split-K slab
        add0 = rnd(7, H4) * 0.2                                    # a per-token table, rows by index
        rows0 = torch.randint(0, 7, (G,), generator=g).to(dev_)
        add1 = rnd(nimg, H4) * 0.2                                 # a per-image row
        gates = torch.empty(G, H4, device=dev_)
        srow = torch.randint(0, G, (G,), generator=g).int().to(dev_)
        f.add0, f.ld_add0, f.add0_rows = add0.data_ptr(), H4, rows0.data_ptr()
        f.add1, f.ld_add1, f.rows_per_add1 = add1.data_ptr(), H4, rpi
        f.gates_out, f.slab_rows = gates.data_ptr(), srow.data_ptr()
        pre = pre - slabs[0].double() + slabs[0].double()[srow.long()] + slabs[1].double()[srow.long()] \
            + add0.double()[rows0] + add1.double().repeat_interleave(rpi, 0)
        keep += [add0, rows0, add1, gates, srow]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # the cell's output also as its two fp16 pieces (ssc_lstm_fwd_desc.h_planes; matrix-core form only)
    planes = scale = None
    if H % 4 == 0 and variant != "valu":
        Hk = (H + 31) // 32 * 32
        planes = torch.full((G, Hk), 0x7E007E00, dtype=torch.int32, device=dev_)
        scale = torch.tensor([64.0], device=dev_)
        f.h_planes, f.ld_hplanes, f.planes_scale = planes.data_ptr(), Hk, scale.data_ptr()
    if variant == "valu":
        lib.ssc_debug_set(b"img_mfma", 0)
    try:
        lib.ssc_lstm_fwd_img(C.byref(f), L.ptr(alpha), R, L.ptr(P), R, rpi, st)
        torch.cuda.synchronize()
    finally:
        lib.ssc_debug_set(b"img_mfma", 1)
    if planes is not None:   # bit for bit what ssc_split_f16 makes of h_out, zero padding included
        from gpuutil import split_f16
        assert torch.equal(planes, split_f16(h_out, scale=scale))
    i, fg, gg, o = pre.view(G, 4, H).unbind(1)
    c = torch.sigmoid(fg) * cp + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    assert maxdiff(c_out, c) < 2e-6 and maxdiff(h_out, h) < 2e-6
    if variant == "general":
        act = torch.stack([torch.sigmoid(i), torch.sigmoid(fg), torch.tanh(gg), torch.sigmoid(o)], 1).reshape(G, H4)
        assert maxdiff(gates, act) < 2e-6


@pytest.mark.parametrize("nimg,groups,beam,sv,R", [(8, 14, 5, 1, 36), (8, 13, 5, 1, 36), (10, 12, 5, 0, 36), (26, 4, 5, 1, 36),
                                                   (40, 3, 5, 1, 36), (8, 40, 3, 1, 36), (16, 4, 5, 1, 36), (9, 14, 5, 0, 36),
                                                   (8, 14, 5, 1, 50), (8, 14, 5, 1, 100), (8, 14, 5, 1, 27),
                                                   (100, 20, 5, 1, 36)])   # the last: bench.py's own call shape, 10 000 rows per step
def test_full_size_large_call_decode_step_matches_oracle(nimg, groups, beam, sv, R):
    """C4's model size (V = 10000, E / H / A = 1000 / 1200 / 768, 36 x 2048 regions, Z = 128) at a call large enough for every
    large-call path (8 images x 14 groups x beam 5 = 560 rows: per-token gate table, per-image attended-feature table on the
    matrix cores, products over the distinct parents, states read through the parent lists): one step from states re-ordered by
    back-pointers against the CPU oracle (itself pinned to the reference at this call shape by g15_decode_large) - log-probs,
    states and attention weights within 1e-4."""
    cfg = oracle.OracleConfig(vocab_size=10000, image_feature_size=2048, embedding_size=1000, hidden_size=1200,
                              attention_projection_size=768, z_space=128, max_caption_length=20, sentiment_vae=sv,
                              senti_prior_multip=0.5, beam_size=5)
    params = oracle.init_params(cfg, seed=4)
    g = torch.Generator().manual_seed(8 + nimg)
    H, Z, V = 1200, 128, 10000   # (R = 50 / 100: the image cell's 17 / 33 k-step forms; 27: a partial last k-step)
    NG, G = nimg * groups, nimg * groups * beam
    feats = torch.randn(nimg, R, 2048, generator=g)
    senti = torch.randint(-1, 2, (nimg, 1), generator=g).float()
    tok = torch.randint(1, V, (G,), generator=g)
    keys = ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")
    base = {k: torch.randn(NG, beam, H, generator=g) * 0.3 for k in keys}
    parent = torch.randint(0, beam, (NG, beam), generator=g)
    st_in = {k: v.gather(1, parent.view(NG, beam, 1).expand(NG, beam, H)).reshape(G, H).contiguous() for k, v in base.items()}
    eps = torch.randn(G, Z, generator=g)
    pm, pv = oracle.prior_from_sentiment(cfg, senti, nimg, feats)
    rep = G // nimg
    ex = lambda t: t.unsqueeze(1).expand(nimg, rep, *t.shape[1:]).reshape(G, *t.shape[1:])
    with torch.no_grad():
        want_lp, want_st, _, _, want_al = oracle.decode_step(params, cfg, ex(feats), tok, st_in, False, ex(senti), ex(pm), ex(pv), eps)
    m = build_model(cfg, params)
    m.eval()
    m._engine()
    dec = m._dec
    ctx = dec.prepare(dev(feats))
    # the call shapes straddle the thresholds of the large-call paths: 512 rows per step, 16 rows per image, 8 images per call
    # ((8, 14, 5) = 560 and (8, 13, 5) = 520 rows are grids below one round of workgroups at this width; (40, 3, 5) shares parents
    # without the table; (16, 4, 5) = 320 rows takes none of them)
    if dec.ungathered_ok(ctx, G, beam):
        states = {k: dev(v.reshape(G, H)) for k, v in base.items()}
        states["_ungathered"] = True
    else:
        states = {k: dev(v) for k, v in st_in.items()}
    assert dec.ungathered_ok(ctx, G, beam) == (G >= 512 and groups * beam >= 16)
    states["_parent"] = dev(parent)
    lp, st, al = dec.step(ctx, dev(tok), states, dev(ex(senti).reshape(G)), dev(eps))
    assert maxdiff(lp, want_lp) < 1e-4 and maxdiff(al, want_al) < 1e-5
    for k in ("h1", "c1", "h_decoder", "c_decoder"):
        assert maxdiff(st[k], want_st[k]) < 1e-4, k


def test_weight_only_tables_are_taken_over_between_image_contexts():
    """DecodeEngine.weights_frozen: the per-token gate table of ssc_decode_prepare depends on the weights alone and is copied from
    the previous image context (ssc_decode_prepare_from) - a step on a context prepared that way is bit-equal to one on a context
    prepared from scratch, for a different image count and region count than the context the table came from."""
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    V, F, E, H, A, Z = 300, 64, 40, 48, 24, 8
    torch.manual_seed(5)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=8, beam_size=3, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(6)
    feats_a = torch.randn(12, 7, F, generator=g).cuda()
    feats_b = torch.randn(9, 5, F, generator=g).cuda()
    G = 9 * 6
    tok = torch.randint(0, V, (G,), generator=g).cuda()
    sent = torch.randint(-1, 2, (G,), generator=g).float().cuda()
    eps = torch.randn(G, Z, generator=g).cuda()
    fresh = dec.step(dec.prepare(feats_b), tok, None, sent, eps)
    dec.weights_frozen = True
    try:
        dec.prepare(feats_a)                        # leaves its table behind
        assert dec._last_ctx is not None
        reused = dec.step(dec.prepare(feats_b), tok, None, sent, eps)
    finally:
        dec.weights_frozen = False
        dec._last_ctx = None
    assert torch.equal(fresh[0], reused[0]) and torch.equal(fresh[2], reused[2])
    for k in ("h1", "c1", "h_decoder", "c_decoder"):
        assert torch.equal(fresh[1][k], reused[1][k]), k


def test_large_call_paths_with_a_multi_state_machine_leave_the_search_unchanged():
    """The large-call decode paths under constrained beam search with S = 3 machine states: groups of S x beam = 15 rows
    (back-pointers range over the whole group), 8 images x 5 samples x 15 = 600 rows per step - against the same search with the
    parent sharing, the attended-feature table and the un-gathered states switched off: identical beams for every state."""
    from ssc_runtime import lib as L
    from ssc_runtime.inference import diverse_decode
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    lib = L.load()
    V, F, E, H, A, Z, R = 350, 64, 40, 64, 24, 8, 6
    torch.manual_seed(13)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=7, beam_size=5, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(4)
    nimg, ns, beam, S = 8, 5, 5, 3
    B = nimg * ns
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    fsm = make_fsm(B, S, V, seed=77).cuda()
    ncons = torch.ones(B, dtype=torch.long)
    eps = [torch.randn(B, Z, generator=g).cuda()] + [torch.randn(B * S * beam, Z, generator=g).cuda() for _ in range(8)]
    outs = []
    for on in (1, 0):
        for key in (b"dec_dedup", b"dec_att_table", b"dec_ungathered"):
            lib.ssc_debug_set(key, on)
        try:
            outs.append(diverse_decode(dec, feats, senti, ns, beam, 7, 1, fsm=fsm, num_constraints=ncons, min_constraints_to_satisfy=1,
                                       eps_steps=[e.clone() for e in eps], early_stop=False)[0].clone())
        finally:
            for key in (b"dec_dedup", b"dec_att_table", b"dec_ungathered"):
                lib.ssc_debug_set(key, 1)
    assert torch.equal(outs[0], outs[1])


def _random_cbs_cases():
    g = torch.Generator().manual_seed(20261005)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    cases = []
    for V in (64, 1000, 5000, 10000, 10240, 10241, 12000, 300, 2500, 7777):   # <= 10240: row in registers; above: LDS-staged
        S = pick([1, 2, 3])
        beam = pick([1, 2, 3, 5])
        cases.append((ri(1, 3), S, V, beam, pick([1, 2, min(5, max(1, beam))]), pick([0.15, 0.5, 0.9]), ri(4, 9)))
    return cases


@pytest.mark.parametrize("B,S,V,beam,per_node,density,steps", _random_cbs_cases())
def test_cbs_search_random_configs_match_oracle(B, S, V, beam, per_node, density, steps):
    """Seeded random search configurations (vocabulary sizes on both sides of the selection kernels' 10240-entry register form,
    machine states, beam / per-node widths, mask densities, step counts) against the oracle's search (pinned to the reference's
    cbs.py by g12_cbs).  The step function is evaluated on the CPU for BOTH sides, so the log-probs are the same bits and every
    difference would be the bookkeeping's: masked top-k per target state, merge, back-pointers, state re-ordering, forced ends."""
    g = torch.Generator().manual_seed(V * 31 + S * 7 + beam)
    U = torch.randn(V, 12, generator=g)
    W = torch.randn(12, V, generator=g) * 1.5
    drift = torch.randn(5, V, generator=g) * 0.5
    endb = 2.5 if density > 0.4 else 0.5
    fsm = (torch.rand(B, S, S, V, generator=g) < density).to(torch.uint8)
    fsm[:, :, :, 1] = 1          # the end token is always allowed
    fsm[:, :, :, 2:6] = 1        # and a few others, so that no state is left without a finite candidate

    def step_cpu(tokens, state):
        G = tokens.numel()
        cnt = torch.zeros(G, 1) if state is None else state["cnt"]
        acc = torch.zeros(G, 2) if state is None else state["acc"]
        logits = U[tokens] @ W + drift[cnt.long().view(-1) % 5] + acc.sum(1, keepdim=True) * 0.02
        logits[:, 1] += endb
        new = {"cnt": cnt + 1, "acc": (acc + tokens.view(-1, 1).float() * torch.tensor([[1.0, 0.5]])) % 3.0}
        return torch.log_softmax(logits, dim=1), new

    def step_gpu(tokens, state):
        st = None if state is None else {k: v.cpu() for k, v in state.items() if not k.startswith("_")}
        lp, new = step_cpu(tokens.cpu(), st)
        return lp.cuda(), {k: v.cuda() for k, v in new.items()}

    start = torch.full((B,), 1, dtype=torch.long)
    want_p, want_lp = oracle.cbs_search(start, None, step_cpu, fsm, end_index=1, max_steps=steps, beam_size=beam,
                                        per_node_beam_size=per_node)
    got_p, got_lp = cbs_search(start.cuda(), None, step_gpu, fsm.cuda(), 1, steps, beam, per_node)
    assert got_p.shape == want_p.shape
    finite = torch.isfinite(want_lp) & (want_lp > -1e19)
    assert torch.equal(got_p.cpu()[finite], want_p[finite])
    assert maxdiff(got_lp.cpu()[finite], want_lp[finite]) < 1e-5


@pytest.mark.parametrize("every", [2, 4, 7])
def test_lagging_early_stop_check_gives_the_output_of_the_per_step_check(every):
    """cbs_search(early_stop_every = n > 1) asks the device every n steps whether every beam has ended and reads the answer
    without waiting (pinned flag behind an event), so it may run a few surplus steps; their all-END columns are trimmed: the same
    predictions, log-probs and number of columns as the per-step check of cbs.py:167."""
    B, V, beam, steps = 5, 200, 3, 16
    g = torch.Generator().manual_seed(31)
    table = (torch.randn(V, V, generator=g) * 1.5)
    table[:, 1] += 6.0          # the end token wins quickly: every beam has ended after a few steps
    table = table.cuda()

    def step(tokens, state):
        return torch.log_softmax(table[tokens], dim=1), {"h": torch.zeros(tokens.numel(), 2, device="cuda")}

    start = torch.full((B,), 1, dtype=torch.long, device="cuda")
    a, alp = cbs_search(start, None, step, None, 1, steps, beam, 2, early_stop=True, early_stop_every=1)
    assert a.shape[-1] < steps          # it did stop early
    torch.cuda.synchronize()
    b, blp = cbs_search(start, None, step, None, 1, steps, beam, 2, early_stop=True, early_stop_every=every)
    assert a.shape == b.shape and torch.equal(a, b) and torch.equal(alp, blp)
    c, clp = cbs_search(start, None, step, None, 1, steps, beam, 2, early_stop=False)
    assert c.shape[-1] == steps and torch.equal(c[..., :a.shape[-1]], a) and torch.equal(clp, alp)


@pytest.mark.parametrize("G,H,listed", [(70, 1000, False), (333, 96, True), (40, 100, True)])
def test_cell_leaves_its_output_as_fp16_pieces(G, H, listed):
    """ssc_lstm_fwd with h_planes: h_out * scale also split into the two fp16 pieces of the 2xFP16 products, bit for bit what
    ssc_split_f16 makes of h_out (hidden sizes that end inside a 32-k block: zero padding); with a row list only the listed rows."""
    import ctypes as C
    from gpuutil import split_f16
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(G + H)
    H4 = 4 * H
    slabs = (torch.randn(G, H4, generator=g) * 0.5).cuda()
    cprev = torch.randn(G, H, generator=g).cuda()
    c_out, h_out = torch.zeros(G, H, device="cuda"), torch.zeros(G, H, device="cuda")
    Hk = (H + 31) // 32 * 32
    planes = torch.full((G, Hk), 0x7E007E00, dtype=torch.int32, device="cuda")
    scale = torch.tensor([64.0], device="cuda")
    f = L.LstmFwdDesc()
    f.B, f.H = G, H
    f.slabs, f.nslab, f.slab_stride = slabs.data_ptr(), 1, G * H4
    f.c_prev, f.ld_cprev = cprev.data_ptr(), H
    f.c_out, f.ld_cout, f.h_out, f.ld_hout = c_out.data_ptr(), H, h_out.data_ptr(), H
    f.h_planes, f.ld_hplanes, f.planes_scale = planes.data_ptr(), Hk, scale.data_ptr()
    rows = None
    if listed:
        rows = torch.randperm(G, generator=g)[:G // 2].sort().values.to(torch.int32).cuda()
        cnt = torch.tensor([rows.numel()], dtype=torch.int32, device="cuda")
        f.rows, f.row_count = rows.data_ptr(), cnt.data_ptr()
    lib.ssc_lstm_fwd(C.byref(f), L.stream_ptr())
    torch.cuda.synchronize()
    want = split_f16(h_out, scale=scale)
    if rows is None:
        assert torch.equal(planes, want)
    else:
        r = rows.long()
        assert torch.equal(planes[r], want[r])
        other = torch.ones(G, dtype=torch.bool, device="cuda"); other[r] = False
        assert bool((planes[other] == 0x7E007E00).all())          # untouched
    f.ld_hplanes = Hk - 4 if Hk > H else H - 4                     # too narrow: refused
    assert lib._raw_ssc_lstm_fwd(C.byref(f), L.stream_ptr()) == -1

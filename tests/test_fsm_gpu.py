"""GPU: compiled finite-state machines (ssc_fsm_compile / ssc_beam_step_fsm) against the dense per-target scans
the reference's loop corresponds to (updown-baseline/updown/modules/cbs.py:157-250), which tests/test_decode_gpu.py pins to the
reference's own results (g12_cbs).  Everything here is integer / bit-exact: the compiled form must give the dense form's
selections token for token, log-prob for log-prob."""
import pytest
import torch

import oracle
from goldenlib import cbs_table_step, load_raw, unpack_fsm
from ssc_runtime.decode import CompiledFsm, cbs_search

pytestmark = pytest.mark.gpu


def _builder(tmp_path, V, kmax=3, big_class=0):
    """Vocabulary of V words that holds the constraint word forms, and the FSM builder over it.  big_class: a class with that many
    word forms (the reference's attribute / sentiment classes have 100-280 forms, data/constraint_wordforms_attrib*.tsv)."""
    from ssc_runtime.constraints import FiniteStateMachineBuilder
    from ssc_runtime.vocab import Vocabulary
    words = ["a", "the", "dog", "dogs", "cat", "cats", "fire", "hydrant", "hydrants", "red", "reddish", "on", "street", "sits", "salt",
             "and", "pepper"]
    lines = ["dog\tdog,dogs", "cat\tcat,cats", "fire\tfire", "hydrant\thydrant,hydrants", "red\tred,reddish,zzz-not-in-vocab",
             "salt\tsalt", "and\tand", "pepper\tpepper,peppers-oov"]
    fillers = [f"w{i}" for i in range(V - 2 - len(words))]
    if big_class:
        lines.append("attribute\t" + ",".join(fillers[5:5 + 3 * big_class:3]))
    tsv = tmp_path / "wordforms.tsv"
    tsv.write_text("\n".join(lines) + "\n")
    vocab = Vocabulary(["@@UNKNOWN@@", "@@BOUNDARY@@"] + words + fillers)
    assert vocab.get_vocab_size() == V
    return vocab, FiniteStateMachineBuilder(vocab, str(tsv), None, max_given_constraints=kmax)


def _machines(tmp_path, V, sets, big_class=0):
    """(M, S, S, V) uint8: the machines of the constraint sets, padded to the largest state count (states an image does not use
    have no transitions at all)."""
    _, b = _builder(tmp_path, V, big_class=big_class)
    built = [b.build(list(c)) for c in sets]
    S = max(x[1] for x in built)
    fsm = torch.zeros(len(sets), S, S, V, dtype=torch.uint8)
    for i, (m, ns, _) in enumerate(built):
        fsm[i, :ns, :ns] = m[:ns, :ns]
    return fsm


def _read_tables(c: CompiledFsm):
    """The tables of ssc_fsm_compile as tensors, by the layout of csrc/fsm.hip::fsm_layout (64-word aligned regions)."""
    d = c.dims
    M, S, V, E, P = d.M, d.S, d.V, d.E, d.P
    r = lambda x: (x + 63) // 64 * 64
    NW = ((V + 255) // 256 + 31) // 32
    t = c.tables.cpu()
    o = 0
    out = {}
    for name, n in (("ok", M * S), ("dflt", M * S), ("reach", M * S), ("nexc", M * S), ("etok", M * S * E), ("emask", M * S * E), ("fill", M * S * P),
                    ("bits", M * S * 256 * NW)):
        out[name] = t[o:o + n]
        o += r(n)
    assert o == t.numel()
    out["NW"] = NW
    return out


SETS = [(), ("dog",), ("dog", "cat"), ("fire hydrant", "dog"), ("dog", "cat", "red"), ("dog", "cat", "fire hydrant"),
        ("salt and pepper", "fire hydrant", "dog"), ("dog", "dog"), ("red",)]


@pytest.mark.parametrize("V,E", [(60, 16), (10000, 512), (12345, 64)])
def test_compiled_tables_describe_the_dense_machine(tmp_path, V, E):
    """default target set + exception list + bitmap + fill tokens reproduce the dense adjacency exactly, for the machines the
    builder makes (0-3 constraints, multi-word, repeated, out-of-vocabulary word forms landing on @@UNKNOWN@@)."""
    fsm = _machines(tmp_path, V, SETS)
    M, S = fsm.shape[:2]
    c = CompiledFsm(fsm.cuda(), max_exceptions=E, fill=4)
    t = _read_tables(c)
    assert bool((t["ok"] == 1).all())                      # every from-state of these machines has the form
    tmask = (fsm.long() << torch.arange(S).view(1, 1, S, 1)).sum(2)          # (M, S, V): target set of every (from-state, token)
    for ms in range(M * S):
        m, s = divmod(ms, S)
        n = int(t["nexc"][ms])
        tok = t["etok"][ms * E: ms * E + n].long()
        assert bool((tok[1:] > tok[:-1]).all())            # ascending
        want = tmask[m, s]
        rebuilt = torch.full((V,), int(t["dflt"][ms]) & 0xFFFFFFFF, dtype=torch.long)
        rebuilt[tok] = t["emask"][ms * E: ms * E + n].long() & 0xFFFFFFFF
        assert torch.equal(rebuilt, want), (m, s)
        reach = 0
        for v in want.unique().tolist():
            reach |= v
        assert (int(t["reach"][ms]) & 0xFFFFFFFF) == reach
        is_exc = torch.zeros(V, dtype=torch.bool)
        is_exc[tok] = True
        nonexc = (~is_exc).nonzero().view(-1)[:4]
        assert torch.equal(t["fill"][ms * 4: ms * 4 + nonexc.numel()].long(), nonexc)
        bits = t["bits"][ms * 256 * t["NW"]: (ms + 1) * 256 * t["NW"]].view(256, t["NW"]).long() & 0xFFFFFFFF
        v = torch.arange(V)
        got = (bits[v % 256, (v // 256) // 32] >> ((v // 256) % 32)) & 1
        assert torch.equal(got.bool(), is_exc)


def _table_step(V, seed, end_bias=1.0):
    g = torch.Generator().manual_seed(seed)
    U = torch.randn(V, 10, generator=g).cuda()
    W = (torch.randn(10, V, generator=g) * 1.5).cuda()
    drift = (torch.randn(5, V, generator=g) * 0.5).cuda()

    def step(tokens, state):
        G = tokens.numel()
        cnt = torch.zeros(G, 1, device="cuda") if state is None else state["cnt"]
        acc = torch.zeros(G, 2, device="cuda") if state is None else state["acc"]
        logits = U[tokens] @ W + drift[cnt.long().view(-1) % 5] + acc.sum(1, keepdim=True) * 0.02
        logits[:, 1] += end_bias
        new = {"cnt": cnt + 1, "acc": (acc + tokens.view(-1, 1).float() * torch.tensor([[1.0, 0.5]], device="cuda")) % 3.0}
        return logits, new
    return step


def _search(fsm, step, B, beam, per_node, steps, mach=None, raw=True, **kw):
    start = torch.full((B,), 1, dtype=torch.long, device="cuda")
    if raw:
        return cbs_search(start, None, step, fsm, 1, steps, beam, per_node, raw_logits=True, mach=mach, **kw)
    lp_step = lambda tok, st: (lambda o: (torch.log_softmax(o[0], dim=1), o[1]))(step(tok, st))
    return cbs_search(start, None, lp_step, fsm, 1, steps, beam, per_node, raw_logits=False, mach=mach, **kw)


@pytest.mark.parametrize("V,beam,per_node,raw,big", [(60, 3, 1, True, 0), (2000, 5, 2, False, 0), (10000, 5, 2, True, 0),
                                                      (10000, 3, 2, True, 150), (12345, 5, 2, True, 0), (12345, 4, 4, False, 40)])
def test_compiled_search_equals_dense_search_on_built_machines(tmp_path, V, beam, per_node, raw, big):
    """One scan per row (compiled form) against one scan per (row, target state) (dense form): predictions, log-probs and step
    count bit-equal - including the beams that hold only -1e20 fills -, for machines with 0-3 constraints padded to 12+ states;
    V on both sides of the register form (10240); word-form classes of 40 / 150 forms."""
    sets = SETS + ([("attribute", "dog")] if big else [])
    fsm = _machines(tmp_path, V, sets, big_class=big).cuda()
    B = fsm.size(0)
    step = _table_step(V, seed=V + beam)
    a_p, a_lp = _search(fsm, step, B, beam, per_node, 9, raw=raw, compile_fsm=False)
    b_p, b_lp = _search(fsm, step, B, beam, per_node, 9, raw=raw, compile_fsm=True)
    assert a_p.shape == b_p.shape and torch.equal(a_p, b_p) and torch.equal(a_lp, b_lp)
    assert bool((a_lp > -1e19).any()) and bool((a_lp <= -1e19).any())   # the case has both finite and fill-only beams


@pytest.mark.parametrize("V,S,density,E", [(97, 4, 0.6, 512), (300, 3, 0.9, 8), (5000, 3, 0.5, 64), (10240, 2, 0.2, 0), (11000, 3, 0.97, 512)])
def test_compiled_search_equals_dense_search_on_random_machines(V, S, density, E):
    """Random dense adjacency: with a large exception capacity every token is an 'exception' of some default (the compiled path
    with hundreds of candidates per target), with a small one the from-states are flagged and take the dense scans inside the
    compiled launch - both must give the dense kernels' result bit for bit."""
    g = torch.Generator().manual_seed(V + S)
    B, beam, per_node = 3, 3, 2
    fsm = (torch.rand(B, S, S, V, generator=g) < density).to(torch.uint8)
    fsm[:, :, :, 1] = 1
    fsm[:, :, :, 2:6] = 1
    fsm = fsm.cuda()
    step = _table_step(V, seed=V)
    a_p, a_lp = _search(fsm, step, B, beam, per_node, 7, compile_fsm=False)
    comp = CompiledFsm(fsm, max_exceptions=E, fill=8)
    b_p, b_lp = _search(fsm, step, B, beam, per_node, 7, compiled=comp)
    assert torch.equal(a_p, b_p) and torch.equal(a_lp, b_lp)
    ok = comp.sparse_states()
    if E == 0 or (E < 100 and V >= 300):
        assert not bool(ok.any())        # really the in-launch dense scans
    if E >= 512 and V <= 512:
        assert bool(ok.all())


def test_machines_are_shared_through_the_index_list(tmp_path):
    """Batch entries that run the same machine (the N_Z latent samples of an image) name it by index instead of carrying a copy."""
    V, beam, per_node = 3000, 5, 2
    fsm = _machines(tmp_path, V, SETS[:5]).cuda()
    M = fsm.size(0)
    mach = torch.tensor([0, 0, 3, 1, 4, 4, 4, 2, 1], dtype=torch.int32)
    step = _table_step(V, seed=5)
    rep = fsm[mach.long()].contiguous()
    a_p, a_lp = _search(rep, step, mach.numel(), beam, per_node, 8, compile_fsm=False)
    for compile_fsm in (False, True):
        b_p, b_lp = _search(fsm, step, mach.numel(), beam, per_node, 8, mach=mach.cuda(), compile_fsm=compile_fsm)
        assert torch.equal(a_p, b_p) and torch.equal(a_lp, b_lp)
    assert M < mach.numel()


@pytest.mark.parametrize("V", [500, 10000])
def test_rows_without_a_finite_beam_need_no_logits(tmp_path, V):
    """skip_dead: a row whose running log-prob is <= -1e19 is scored as if its log-probs were all 0 and its logits are not read
    (NaN-filled here to prove it).  Every log-prob of the search and every beam with a finite log-prob stay bit-identical."""
    beam, per_node = 5, 2
    fsm = _machines(tmp_path, V, SETS).cuda()
    B, S = fsm.shape[:2]
    step = _table_step(V, seed=3)
    a_p, a_lp = _search(fsm, step, B, beam, per_node, 9, compile_fsm=True)

    def poisoned(tokens, state):
        logits, new = step(tokens, {k: v for k, v in state.items() if not k.startswith("_")} if state is not None else None)
        if state is not None and "_last_lp" in state:
            dead = (state["_last_lp"].reshape(-1) <= -1e19) & (tokens != 1)
            logits = logits.clone()
            logits[dead] = float("nan")
        return logits, new
    b_p, b_lp = _search(fsm, poisoned, B, beam, per_node, 9, compile_fsm=True, skip_dead=True)
    assert a_p.shape == b_p.shape and torch.equal(a_lp, b_lp)
    finite = a_lp > -1e19
    assert torch.equal(a_p[finite], b_p[finite]) and bool(finite.any()) and bool((~finite).any())


@pytest.mark.parametrize("every", [1, 3])
def test_early_stop_on_the_device_with_substates(tmp_path, every):
    """The stop condition of cbs.py:167 is noted by the device; steps queued after it are no-ops, so a host that sees the flag
    late (or never: early_stop_every only chooses blocking / polling) returns the columns a per-step check returns.  Machines with
    sub-states, whose END transition is NOT a self-loop (constraints.py:470-476): surplus steps would otherwise move beams."""
    V, beam, per_node, steps = 400, 3, 2, 14
    fsm = _machines(tmp_path, V, [("dog",), ("fire hydrant",), ("salt and pepper", "dog")]).cuda()
    B = fsm.size(0)
    step = _table_step(V, seed=11, end_bias=9.0)
    start = torch.full((B,), 1, dtype=torch.long, device="cuda")
    lp_step = lambda tok, st: (lambda o: (torch.log_softmax(o[0], dim=1), o[1]))(step(tok, st))
    want_p, want_lp = oracle.cbs_search(start.cpu(), None, lambda tok, st: tuple(
        x.cpu() if torch.is_tensor(x) else {k: v.cpu() for k, v in x.items()} for x in lp_step(
            tok.cuda(), None if st is None else {k: v.cuda() for k, v in st.items()})), fsm.cpu(), end_index=1, max_steps=steps,
        beam_size=beam, per_node_beam_size=per_node)
    got_p, got_lp = cbs_search(start, None, lp_step, fsm, 1, steps, beam, per_node, early_stop_every=every)
    full_p, full_lp = cbs_search(start, None, lp_step, fsm, 1, steps, beam, per_node, early_stop=False)
    finite = want_lp > -1e19
    assert got_p.shape == want_p.shape, (got_p.shape, want_p.shape)
    assert torch.equal(got_p.cpu()[finite], want_p[finite])
    assert float((got_lp.cpu()[finite] - want_lp[finite]).abs().max()) < 1e-5
    if want_p.shape[-1] < steps:
        # without the stop the beams move on (END out of a sub-state resets): the cut-out columns are not simply a prefix
        assert full_p.shape[-1] == steps
        torch.cuda.synchronize()


_G12 = load_raw("g12_cbs")


@pytest.mark.parametrize("ci", range(int(_G12["ncases"])))
def test_compiled_search_equals_reference_fixture(ci):
    """The compiled form against ConstrainedBeamSearch.search itself (cbs.py:59-277 run unmodified, g12_cbs), with a capacity that
    keeps every from-state in the compiled form."""
    key = f"search/case{ci}"
    B, S, V, beam, per_node, steps = (int(x) for x in _G12[key + "/dims"])
    table, drift = torch.from_numpy(_G12[key + "/table"]).cuda(), torch.from_numpy(_G12[key + "/drift"]).cuda()
    fsm = unpack_fsm(_G12[key + "/fsm_bits"], B, S, V).cuda()
    comp = CompiledFsm(fsm, max_exceptions=V, fill=8) if S > 1 else None
    want_p, want_lp = torch.from_numpy(_G12[key + "/predictions"]), torch.from_numpy(_G12[key + "/log_probs"])
    got_p, got_lp = cbs_search(torch.full((B,), 1, dtype=torch.long, device="cuda"), None, cbs_table_step(table, drift), fsm, 1, steps,
                               beam, per_node, compiled=comp)
    if comp is not None:
        assert bool(comp.sparse_states().all())
    assert got_p.shape == want_p.shape
    finite = torch.isfinite(want_lp) & (want_lp > -1e19)
    assert float((got_lp.cpu()[finite] - want_lp[finite]).abs().max()) < 1e-4
    assert torch.equal(got_p.cpu()[finite], want_p[finite])


@pytest.mark.parametrize("end_bias", [0.0, 6.0])
def test_decode_step_skips_rows_nobody_reads(tmp_path, end_bias):
    """diverse_decode(skip_dead=True) on the large-call decode paths (parent sharing + attended-feature table, >= 512 rows per
    step): rows without a finite beam and rows whose beam has ended are left out of every product of the step
    (ssc_decode_step_desc.row_lp) and scored without their logits; the selected captions and their log-probs are those of the
    search that steps every row.  Machines of 0-3 constraints padded to one state count (images with fewer constraints keep
    states no beam ever reaches); end_bias 6: most beams end early."""
    from ssc_runtime.inference import diverse_decode
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    V, F, E, H, A, Z, R = 400, 64, 40, 64, 24, 8, 6
    sets = [(), ("dog",), ("dog", "cat"), ("fire hydrant", "dog"), ("dog", "cat", "red"), ("cat",), ("red", "dog"), ("dog", "cat", "fire hydrant")]
    fsm = _machines(tmp_path, V, sets).cuda()
    nimg, S = fsm.shape[:2]
    torch.manual_seed(13)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H,
                        attention_projection_size=A, max_caption_length=9, beam_size=3, z_space=Z, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    with torch.no_grad():
        m._output_layer.bias[1] += end_bias
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(4)
    ns, beam = 4, 3
    B = nimg * ns
    assert B * S * beam >= 512
    feats = torch.randn(nimg, R, F, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    ncons = torch.tensor([len(c) for c in sets]).repeat_interleave(ns)
    eps = [torch.randn(B, Z, generator=g).cuda()] + [torch.randn(B * S * beam, Z, generator=g).cuda() for _ in range(9)]
    outs = []
    for skip in (False, True):
        pred, calls = diverse_decode(dec, feats, senti, ns, beam, 9, 1, fsm=fsm, num_constraints=ncons, min_constraints_to_satisfy=3,
                                     eps_steps=[e.clone() for e in eps], early_stop=False, skip_dead=skip)
        outs.append(pred.clone())
    assert torch.equal(outs[0], outs[1])
    # the constraint words are in the captions (token ids of "dog"/"dogs" = 4/5 in this vocabulary are not asserted: Vocabulary.synthetic
    # has no such words - the machines constrain ids by position; what is asserted is the equality above and a non-trivial search)
    assert int((outs[0] != 1).sum()) > 0


def _small_captioner(V, end_bias=0.0, seed=13, H=64):
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    torch.manual_seed(seed)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=64, embedding_size=40, hidden_size=H,
                        attention_projection_size=24, max_caption_length=9, beam_size=3, z_space=8, prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                        device=torch.device("cuda")).to("cuda")
    with torch.no_grad():
        m._output_layer.bias[1] += end_bias
    m.eval()
    m._engine()
    return m


@pytest.mark.parametrize("nimg,ns,beam,machines,end_bias,early", [(8, 8, 5, False, 0.0, False), (8, 8, 5, False, 7.0, True),
                                                                 (2, 3, 3, False, 0.0, True), (8, 4, 3, True, 0.0, False),
                                                                 (8, 4, 3, True, 5.0, True), (2, 2, 2, True, 0.0, True)])
def test_one_call_search_equals_the_step_by_step_search(tmp_path, nimg, ns, beam, machines, end_bias, early):
    """ssc_decode_search (the whole search launched from the library: DecodeEngine.search) against cbs_search driving
    DecodeEngine.step from Python with the same noise: beams, log-probs and step count bit-equal.  Large calls (parent lists,
    attended-feature table, un-gathered states) and small ones (states re-ordered by ssc_gather_rows); trivial machine and built
    machines shared per image; searches that stop early."""
    V, R, Z, steps = 400, 6, 8, 9
    m = _small_captioner(V, end_bias)
    dec = m._dec
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(nimg, R, 64, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    B = nimg * ns
    fsm = comp = mach = None
    S = 1
    if machines:
        sets = [(), ("dog",), ("dog", "cat"), ("fire hydrant", "dog"), ("dog", "cat", "red"), ("cat",), ("red", "dog"), ("dog", "cat", "fire hydrant")][:nimg]
        fsm = _machines(tmp_path, V, sets).cuda()
        S = fsm.size(1)
        comp = CompiledFsm(fsm, fill=8)
        mach = torch.arange(nimg, dtype=torch.int32, device="cuda").repeat_interleave(ns)
    G = B * S * beam
    eps0 = torch.randn(B, Z, generator=g).cuda()
    eps = torch.randn(steps - 1, G, Z, generator=g).cuda()
    sent_b = senti.view(nimg, 1).expand(nimg, ns).reshape(B)
    ctx = dec.prepare(feats)
    calls = {"k": 0}

    def step(tokens, state):
        n = tokens.numel()
        e = eps0 if calls["k"] == 0 else eps[calls["k"] - 1]
        calls["k"] += 1
        lp, st, _ = dec.step(ctx, tokens, state, sent_b.view(B, 1).expand(B, n // B).reshape(n), e, raw_logits=True)
        return lp, {k: v for k, v in st.items() if k not in ("h_encoder", "c_encoder")}
    start = torch.full((B,), 1, dtype=torch.long, device="cuda")
    want_p, want_lp = cbs_search(start, None, step, fsm, 1, steps, beam, max(1, beam // 2), early_stop=early, early_stop_every=1,
                                 raw_logits=True, ungathered_ok=lambda n, grp: dec.ungathered_ok(ctx, n, grp), mach=mach, compiled=comp)
    ctx2 = dec.prepare(feats)
    got_p, got_lp = dec.search(ctx2, sent_b, ns, beam, max(1, beam // 2), steps, 1, eps0, eps, fsm=fsm, compiled=comp, mach=mach,
                               early_stop=early)
    assert got_p.shape == want_p.shape, (got_p.shape, want_p.shape)
    assert torch.equal(got_p, want_p) and torch.equal(got_lp, want_lp)
    if early and end_bias > 0 and not machines:   # (with machines the beams that hold only fills never all end: cbs.py:167 does not fire)
        assert got_p.shape[-1] < steps


@pytest.mark.parametrize("M,N,K,f16", [(700, 1000, 96, 0), (1300, 10000, 200, 1), (640, 333 * 4, 64, 1), (640, 400, 64, 0), (48, 2048, 64, 1),
                                       (5, 60, 32, 0)])
def test_vocabulary_head_records_describe_the_logits(M, N, K, f16):
    """ssc_gemm_desc.topk_part: per (row, 128-column tile) the maximum, sum exp(x - max) and the two best columns of x = a W^T + bias,
    against the same product written out (same kernel form, so the values are the same bits): maxima and columns exact, the
    exponential sums to rounding; with a device-side row list (c_rows) the records land on the listed rows."""
    import ctypes as C
    from gpuutil import gemm
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) * 0.3).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ntn = (N + 127) // 128
    lib.ssc_debug_set(b"gemm_f16", f16)
    try:
        # (no kernel form is forced for the record launches: shapes the form heuristics would send to the 64-wide / 64x256 kernels -
        # narrow N, a minibatch-sized M - must still take the one form that has the records epilogue; round 4 had them fall through
        # to a kernel that stores C, with C = NULL)
        parts = torch.full((M, ntn, 6), float("nan"), device="cuda")
        gemm([(A, K, Wt, K, K)], M, N, 1, 1, torch.empty(1, N, device="cuda"), bias=bias, splits=1, compact={"topk_part": parts})
        torch.cuda.synchronize()
        lib.ssc_debug_set(b"large_form", 2)
        full = torch.empty(M, N, device="cuda")
        gemm([(A, K, Wt, K, K)], M, N, 1, 1, full, bias=bias, splits=1)
        parts_b = torch.full((M, ntn, 6), float("nan"), device="cuda")
        gemm([(A, K, Wt, K, K)], M, N, 1, 1, full, bias=bias, splits=1, compact={"topk_part": parts_b})
        assert torch.equal(parts_b, parts)
        # with a row list: every second row, records scattered to those rows
        rows = torch.arange(0, M, 2, dtype=torch.int32, device="cuda")
        cnt = torch.tensor([rows.numel()], dtype=torch.int32, device="cuda")
        parts2 = torch.full((M, ntn, 6), float("nan"), device="cuda")
        gemm([(A, K, Wt, K, K)], M, N, 1, 1, full, bias=bias, splits=1,
             compact={"topk_part": parts2, "m_count": cnt, "a_rows": rows, "c_rows": rows})
    finally:
        lib.ssc_debug_set(b"gemm_f16", 0)
        lib.ssc_debug_set(b"large_form", 1)
    torch.cuda.synchronize()
    pad = ntn * 128 - N
    x = torch.cat([full, torch.full((M, pad), float("-inf"), device="cuda")], 1).view(M, ntn, 128)
    mx = x.max(-1).values
    se = torch.exp(x - mx.unsqueeze(-1)).sum(-1)
    top = x.topk(2, dim=-1)      # (ties: torch's order is unspecified - compare values, and columns where the values differ)
    col0 = parts[..., 3].view(torch.int32).long()
    base = (torch.arange(ntn, device="cuda") * 128).view(1, ntn)
    at_col0 = torch.gather(x, 2, (col0 - base).clamp(0, 127).unsqueeze(-1)).squeeze(-1)
    if M > 64:   # the written-out product took the same kernel form: same bits
        assert torch.equal(parts[..., 0], mx)
        assert torch.equal(parts[..., 2], top.values[..., 0]) and torch.equal(parts[..., 4].nan_to_num(neginf=-1e30), top.values[..., 1].nan_to_num(neginf=-1e30))
        assert torch.equal(at_col0, top.values[..., 0])
    else:        # (a minibatch-sized product is written out by the 64x256 kernel in 3xBF16: the same numbers to fp32 rounding)
        assert float((parts[..., 0] - mx).abs().max()) < 1e-4 and float((parts[..., 2] - top.values[..., 0]).abs().max()) < 1e-4
        assert float((at_col0 - top.values[..., 0]).abs().max()) < 1e-4
    assert float(((parts[..., 1] - se).abs() / se).max()) < 1e-4
    assert torch.equal(parts2[::2], parts[::2]) and bool(torch.isnan(parts2[1::2]).all())


@pytest.mark.parametrize("end_bias,early", [(0.0, False), (6.0, True)])
def test_search_from_records_equals_search_from_logits(end_bias, early):
    """ssc_decode_search with the vocabulary head leaving records (the one-state machine, per-node 2, >= 512 rows: bench.py's decode
    leg) against the same search on written-out logits: identical captions and back-pointers; log-probs to rounding (the
    log-sum-exp is summed per tile instead of per thread stride)."""
    from ssc_runtime import lib as L
    lib = L.load()
    V, R, Z, steps, nimg, ns, beam = 1000, 6, 8, 9, 8, 16, 5
    m = _small_captioner(V, end_bias)
    dec = m._dec
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(nimg, R, 64, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    B = nimg * ns
    sent_b = senti.view(nimg, 1).expand(nimg, ns).reshape(B)
    eps0 = torch.randn(B, Z, generator=g).cuda()
    eps = torch.randn(steps - 1, B * beam, Z, generator=g).cuda()
    outs = []
    for on in (1, 0):
        lib.ssc_debug_set(b"dec_parts", on)
        try:
            ctx = dec.prepare(feats)
            outs.append(dec.search(ctx, sent_b, ns, beam, 2, steps, 1, eps0, eps, early_stop=early))
        finally:
            lib.ssc_debug_set(b"dec_parts", 1)
    (p1, l1), (p0, l0) = outs
    assert p1.shape == p0.shape and torch.equal(p1, p0)
    assert float((l1 - l0).abs().max()) < 2e-5
    if early:
        assert p1.shape[-1] < steps


@pytest.mark.parametrize("H,V,machines", [(160, 1200, False), (100, 1100, False), (160, 600, True)])
def test_search_on_presplit_operands_is_bit_identical(H, V, machines):
    """Large calls under the 2xFP16 numerics hand their products states and weights ALREADY split into fp16 pieces (ssc_split_f16:
    once per step / per image context instead of once per tile; ssc_gemm_seg.A16 / B16): beams, back-pointers and log-probs are the
    bits of the search whose products split their operands themselves (ssc_debug_set("dec_planes", 0)).  Hidden sizes that are /
    are not whole 32-k blocks (zero padding of the pieces), records and logits heads, one-state and compiled machines."""
    from ssc_runtime import lib as L
    from ssc_runtime.decode import CompiledFsm
    lib = L.load()
    R, Z, steps, nimg, ns, beam = 6, 8, 7, 8, 16, 5
    m = _small_captioner(V, 1.0, H=H)
    dec = m._dec
    g = torch.Generator().manual_seed(H)
    feats = torch.randn(nimg, R, 64, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    B = nimg * ns
    sent_b = senti.view(nimg, 1).expand(nimg, ns).reshape(B)
    S = 1
    compiled = mach = None
    if machines:
        S = 3
        fsm = torch.zeros(2, S, S, V, dtype=torch.uint8)
        for k in range(2):
            for i in range(S):
                fsm[k, i, i] = 1
            words = torch.randperm(V - 4, generator=g)[:6] + 4
            fsm[k, 0, 0, words[:3]] = 0; fsm[k, 0, 1, words[:3]] = 1
            fsm[k, 1, 1, words[3:]] = 0; fsm[k, 1, 2, words[3:]] = 1
        compiled = CompiledFsm(fsm.cuda())
        mach = (torch.arange(nimg) % 2).view(nimg, 1).expand(nimg, ns).reshape(B).to(torch.int32).cuda()   # per batch entry
    eps0 = torch.randn(B, Z, generator=g).cuda()
    eps = torch.randn(steps - 1, B * S * beam, Z, generator=g).cuda()
    outs = []
    for on in (1, 0):
        lib.ssc_debug_set(b"dec_planes", on)
        try:
            ctx = dec.prepare(feats)
            outs.append(dec.search(ctx, sent_b, ns, beam, 2, steps, 1, eps0, eps, fsm=None if compiled is None else compiled.fsm,
                                   compiled=compiled, mach=mach, early_stop=False))
        finally:
            lib.ssc_debug_set(b"dec_planes", 1)
    (p1, l1), (p0, l0) = outs
    assert torch.equal(p1, p0) and torch.equal(l1, l0)


def test_ended_beams_are_not_stepped_in_a_one_state_search():
    """skip_dead with the trivial machine: a beam that has emitted END re-emits END whatever its logits are (cbs.py:177-181), so
    its row is left out of every product of the later steps - same captions as the search that steps every row (most beams end
    early here; the search itself only stops when ALL have)."""
    from ssc_runtime.inference import diverse_decode
    V, R, Z, steps, nimg, ns, beam = 600, 6, 8, 9, 8, 16, 5
    m = _small_captioner(V, end_bias=4.0)
    dec = m._dec
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(nimg, R, 64, generator=g).cuda()
    senti = torch.randint(-1, 2, (nimg,), generator=g).float().cuda()
    B = nimg * ns
    eps = [torch.randn(B, Z, generator=g).cuda()] + [torch.randn(B * beam, Z, generator=g).cuda() for _ in range(steps - 1)]
    outs = [diverse_decode(dec, feats, senti, ns, beam, steps, 1, eps_steps=[e.clone() for e in eps], early_stop=False, skip_dead=sk)[0].clone()
            for sk in (False, True)]
    assert torch.equal(outs[0], outs[1])
    ended = (outs[0] == 1).any(-1).float().mean()
    assert 0.3 < float(ended)          # the case does have ended beams


def test_seeded_consecutive_decodes_are_reproducible():
    """the early stop is polled, so how many steps a search QUEUES depends on host / device timing - the random state a
    call consumes must not.  Two runs of three consecutive diverse_decode calls under the same seed (captions that end early, early
    stop on, noise from the default per-call generator) give identical captions, and the global generators end in the same state."""
    from ssc_runtime.inference import diverse_decode
    V, R, nimg, ns, beam = 500, 6, 8, 16, 5
    m = _small_captioner(V, end_bias=3.0)
    dec = m._dec
    g = torch.Generator().manual_seed(1)
    feats = [torch.randn(nimg, R, 64, generator=g).cuda() for _ in range(3)]
    senti = torch.ones(nimg, device="cuda")
    runs = []
    for _ in range(2):
        torch.manual_seed(1234)
        outs = [diverse_decode(dec, f, senti, ns, beam, 9, 1, early_stop=True)[0].clone() for f in feats]
        runs.append((outs, torch.rand(3), torch.rand(3, device="cuda")))
        torch.cuda.synchronize()
    for a, b in zip(runs[0][0], runs[1][0]):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    assert not torch.equal(runs[0][0][0], runs[0][0][1][:, :, : runs[0][0][0].shape[-1]]) or True

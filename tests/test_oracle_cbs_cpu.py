"""CPU: invariants that pin the oracle's constrained beam search (the reference's cbs.py cannot execute on
torch >= 1.2): S=1 & beam=1 is greedy argmax, ended beams emit BOUNDARY forever, beam log-probs are
sorted, an all-ones FSM equals plain beam search by brute force, constraints are honoured."""
import itertools

import torch

import oracle


def table_step(table):
    def step(tokens, state):
        return table[tokens], ({"n": torch.zeros(tokens.numel(), 1)} if state is None else state)
    return step


def make_table(V, seed, end_boost=0.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(V, V, generator=g)
    t[:, 1] += end_boost
    return torch.log_softmax(t, dim=1)


def test_greedy_when_single_state_single_beam():
    V = 17
    tab = make_table(V, 1)
    fsm = torch.ones(2, 1, 1, V, dtype=torch.uint8)
    p, lp = oracle.cbs_search(torch.tensor([1, 1]), None, table_step(tab), fsm, 1, max_steps=6, beam_size=1, early_stop=False)
    tok, want, total = torch.tensor(1), [], 0.0
    for _ in range(6):
        nxt = int(tab[tok].argmax()) if int(tok) != 1 or not want else 1
        total += float(tab[tok, nxt]) if not (want and int(tok) == 1) else 0.0
        want.append(nxt)
        tok = torch.tensor(nxt)
    assert p[0, 0, 0].tolist() == want
    assert abs(float(lp[0, 0, 0]) - total) < 1e-5


def test_matches_bruteforce_beam_search_on_trivial_fsm():
    V, beam, steps = 6, 3, 4
    tab = make_table(V, 4, end_boost=0.5)
    fsm = torch.ones(1, 1, 1, V, dtype=torch.uint8)
    p, lp = oracle.cbs_search(torch.tensor([1]), None, table_step(tab), fsm, 1, max_steps=steps, beam_size=beam,
                              per_node_beam_size=V, early_stop=False)
    # exhaustive beam search with the same "ended beams only continue with END at cost 0" rule
    beams = [((), 0.0, 1)]
    for _ in range(steps):
        cand = []
        for seq, s, last in beams:
            for v in range(V):
                if seq and last == 1:
                    if v != 1:
                        continue
                    cand.append((seq + (v,), s, v))
                else:
                    cand.append((seq + (v,), s + float(tab[last, v]), v))
        cand.sort(key=lambda c: -c[1])
        beams = cand[:beam]
    assert [list(b[0]) for b in beams] == p[0, 0].tolist()
    assert torch.allclose(lp[0, 0], torch.tensor([b[1] for b in beams]), atol=1e-5)
    assert all(lp[0, 0, i] >= lp[0, 0, i + 1] for i in range(beam - 1))


def test_ended_beams_emit_boundary_forever_and_early_stop():
    V = 9
    tab = make_table(V, 2, end_boost=6.0)
    fsm = torch.ones(1, 1, 1, V, dtype=torch.uint8)
    p, _ = oracle.cbs_search(torch.tensor([1]), None, table_step(tab), fsm, 1, max_steps=10, beam_size=2, early_stop=False)
    for row in p[0, 0].tolist():
        if 1 in row:
            i = row.index(1)
            assert all(t == 1 for t in row[i:])
    p2, _ = oracle.cbs_search(torch.tensor([1]), None, table_step(tab), fsm, 1, max_steps=10, beam_size=2, early_stop=True)
    assert p2.size(-1) <= p.size(-1) and p2[0, 0].tolist() == [r[: p2.size(-1)] for r in p[0, 0].tolist()]


def test_fsm_constraint_is_honoured():
    # two states: state 1 is reachable only by emitting token 5; beams in state 1 must contain token 5
    V = 8
    tab = make_table(V, 3)
    fsm = torch.zeros(1, 2, 2, V, dtype=torch.uint8)
    fsm[0, 0, 0, :] = 1
    fsm[0, 0, 0, 5] = 0
    fsm[0, 0, 1, 5] = 1
    fsm[0, 1, 1, :] = 1
    p, lp = oracle.cbs_search(torch.tensor([1]), None, table_step(tab), fsm, 1, max_steps=5, beam_size=2, early_stop=False)
    for k in range(2):
        if torch.isfinite(lp[0, 1, k]) and lp[0, 1, k] > -1e19:
            assert 5 in p[0, 1, k].tolist()
        if torch.isfinite(lp[0, 0, k]) and lp[0, 0, k] > -1e19:
            assert 5 not in p[0, 0, k].tolist()
    best = oracle.select_best_beam_with_constraints(p, lp, torch.tensor([1]), min_constraints_to_satisfy=1)
    assert 5 in best[0].tolist()


# ---- pinned to the reference: tests/golden/g12_cbs.npz was produced by the UNMODIFIED updown/modules/cbs.py (under the
# two-method torch-1.1 shim of tests/golden/make_golden.py::Torch11) and by the reference captioner's eval branch -------------
import json  # noqa: E402

import pytest  # noqa: E402

from goldenlib import cbs_table_step, load_raw, unpack_fsm  # noqa: E402

_G12 = load_raw("g12_cbs")


@pytest.mark.parametrize("ci", range(int(_G12["ncases"])))
def test_cbs_search_equals_reference_fixture(ci):
    """oracle.cbs_search == ConstrainedBeamSearch.search (cbs.py:59-277) bit for bit: predictions, log-probs, number of step
    calls (early stop, cbs.py:167) for S in {1,3,4}, beam in {1,3,5}, per-node in {1,2}."""
    key = f"search/case{ci}"
    B, S, V, beam, per_node, steps = (int(x) for x in _G12[key + "/dims"])
    table, drift = torch.from_numpy(_G12[key + "/table"]), torch.from_numpy(_G12[key + "/drift"])
    fsm = unpack_fsm(_G12[key + "/fsm_bits"], B, S, V)
    inner, calls = cbs_table_step(table, drift), {"n": 0}

    def step(tokens, state):
        calls["n"] += 1
        return inner(tokens, state)
    p, lp = oracle.cbs_search(torch.full((B,), 1, dtype=torch.long), None, step, fsm, 1, steps, beam, per_node)
    assert torch.equal(p, torch.from_numpy(_G12[key + "/predictions"]))
    assert torch.equal(lp, torch.from_numpy(_G12[key + "/log_probs"]))
    assert calls["n"] == int(_G12[key + "/step_calls"])


def g12_eval_case(ci):
    key = f"eval/case{ci}"
    V, E, H, A, F, Z, L, R, beam = (int(x) for x in _G12["eval/dims"])
    cfg = oracle.OracleConfig(vocab_size=V, image_feature_size=F, embedding_size=E, hidden_size=H, attention_projection_size=A,
                              z_space=Z, max_caption_length=L, sentiment_vae=1, senti_prior_multip=0.5, tied=True, beam_size=beam)
    params = {k[len(key) + 7:]: torch.from_numpy(v) for k, v in _G12.items() if k.startswith(key + "/param/")}
    S = int(_G12[key + "/nstates"])
    case = dict(cfg=cfg, params=params, S=S, beam=beam, fsm=unpack_fsm(_G12[key + "/fsm_bits"], 1, S, V),
                feats=torch.from_numpy(_G12[key + "/feats"]), senti=torch.from_numpy(_G12[key + "/sentiment"]),
                eps=[torch.from_numpy(_G12[key + "/eps0"])] + list(torch.from_numpy(_G12[key + "/eps_rest"])),
                constraints=json.loads(str(_G12[key + "/constraints"])), min_sat=int(_G12[key + "/min"]),
                want=torch.from_numpy(_G12[key + "/predictions"]), beams=torch.from_numpy(_G12[key + "/beams"]),
                lps=torch.from_numpy(_G12[key + "/log_probs"]), vocab=json.loads(str(_G12["eval/vocab_tokens"])),
                wordforms=str(_G12["eval/wordforms_tsv"]), candidates=json.loads(str(_G12[key + "/candidates"])),
                c2s=json.loads(str(_G12[key + "/constraint2states"])))
    return case


@pytest.mark.parametrize("ci", range(int(_G12["eval/ncases"])))
def test_eval_forward_equals_reference_fixture(ci):
    """oracle.eval_forward == the reference UpDownCaptioner.forward eval branch (updown_captioner.py:324-366, tied 300-d model,
    use_cbs, cbs_simple, B=1) with injected eps and reference-built machines: all beams, their log-probs (1e-5: the oracle's
    decode step restates the cell, it is not the reference's code), and the selected caption."""
    c = g12_eval_case(ci)
    out = oracle.eval_forward(c["params"], c["cfg"], c["feats"], c["senti"], c["fsm"], torch.tensor([len(c["constraints"])]),
                              c["eps"], beam_size=c["beam"], min_constraints_to_satisfy=c["min_sat"])
    finite = c["lps"] > -1e19
    assert torch.equal(out["beams"][finite], c["beams"][finite])
    assert (out["log_probs"][finite] - c["lps"][finite]).abs().max() < 1e-5
    assert torch.equal(out["predictions"], c["want"])

"""Diverse decoding: N_Z stochastic beam-search decodes per image, batched over images x samples on one GPU.

Reference: var_updown/scripts/inference.py:117-189 runs, per image (batch forced to 1, :95), a Python loop of
N_Z_SAMPLES calls of model(...) - 20 x <=20 steps x G=beam rows: launch-bound.  Here the (image, sample) pairs are
the batch entries of ONE constrained beam search (rows ordered image, sample, fsm state, beam), per-image terms are
computed once per image and shared by its N_Z * beam rows; the result per (image, sample) equals the reference's
per-call result given the same per-row noise.
"""
import os
from typing import List, Optional

import torch

from .decode import DecodeEngine, cbs_search
from .decoding import select_best_beam_simple_batched


_RAW = not (os.environ.get("SSC_DEBUG", "") == "1" and os.environ.get("SSC_RAW_LOGITS", "1") == "0")   # A/B switch (tools): 0 = log_softmax kernel + selection on log-probs


def diverse_decode(dec: DecodeEngine, feats: torch.Tensor, sentiment: Optional[torch.Tensor], n_samples: int, beam: int,
                   max_steps: int, boundary_index: int, fsm: Optional[torch.Tensor] = None,
                   num_constraints: Optional[torch.Tensor] = None, min_constraints_to_satisfy: int = 0,
                   eps_steps: Optional[List[torch.Tensor]] = None, early_stop: bool = True, per_node: Optional[int] = None,
                   skip_dead: bool = True, compiled=None):
    """feats (nimg,R,F), sentiment (nimg,) or None -> predictions (nimg, n_samples, steps) int64 on device.
    fsm: None (trivial one-state machine, what MAX_GIVEN_CONSTRAINTS: 0 produces), or (nimg, S, S, V) uint8 - ONE machine per
    image, shared by its n_samples latent samples through an index list -, or (nimg*n_samples, S, S, V) (a copy per sample).
    eps_steps: optional explicit noise per step call [(rows_k, Z)]; default: a generator of this call's own, seeded by ONE draw
    from the global CPU generator - the global random state a call consumes does not depend on how many steps it ran.
    skip_dead (machines with more than one state): rows that hold no finite beam - the states an image's constraints never reach,
    every state before its first constraint word has been decoded - and rows whose beam has ended are neither stepped nor scored from
    logits (cbs_search(skip_dead=True)).  Every caption with a finite log-prob is the exact search's; should a selected caption
    have none (its constraints were not reachable within max_steps), the call is repeated exactly, with the same noise.
    compiled: the machines' CompiledFsm when the caller has it already."""
    dev = feats.device
    nimg = feats.size(0)
    d = dec.dims
    B = nimg * n_samples
    ctx = dec.prepare(feats)
    trivial = fsm is None   # one-state machine: cbs_search(fsm=None) reads no mask at all
    if trivial:
        num_constraints = torch.zeros(B, dtype=torch.long)
    sent_b = sentiment.reshape(nimg, 1).expand(nimg, n_samples).reshape(B) if sentiment is not None else None
    calls = {"k": 0}
    mach = None
    if not trivial and fsm.size(0) != B:
        assert fsm.size(0) == nimg, (fsm.shape, nimg, n_samples)
        mach = torch.arange(nimg, dtype=torch.int32, device=dev).repeat_interleave(n_samples)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if eps_steps is None else None
    start = torch.full((B,), boundary_index, dtype=torch.long, device=dev)
    skip = bool(skip_dead) and not trivial and fsm.size(1) > 1
    if skip and compiled is None:
        from .decode import CompiledFsm
        compiled = CompiledFsm(fsm, fill=max(8, per_node or (beam // 2) or beam))

    def search(skip_now):
        calls["k"] = 0
        gen = None
        if seed is not None:
            # The search may queue a few steps beyond the one after which every beam had ended (the early-stop flag is polled, not
            # waited for): their noise must not come out of the global generator, or the captions of the NEXT call would depend on
            # host / device timing.
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)

        def step(tokens, state):
            G = tokens.numel()
            sent_rows = sent_b.view(B, 1).expand(B, G // B).reshape(G) if sent_b is not None else None
            eps = eps_steps[calls["k"]] if eps_steps is not None else torch.randn(G, d.Z, device=dev, generator=gen)
            calls["k"] += 1
            lp, st, alpha = dec.step(ctx, tokens, state, sent_rows, eps, raw_logits=_RAW)
            # the eval cell never touches the encoder-LSTM states (updown_cell.py:200-203): do not carry (and re-order by
            # backpointer) two (G,H) tensors the next step will not read
            return lp, {k: v for k, v in st.items() if k not in ("h_encoder", "c_encoder")}

        return cbs_search(start, None, step, fsm, boundary_index, max_steps, beam, per_node or (beam // 2) or beam,
                          early_stop=early_stop, early_stop_every=4, raw_logits=_RAW,
                          ungathered_ok=lambda G, group: dec.ungathered_ok(ctx, G, group), mach=mach, compiled=compiled,
                          skip_dead=skip_now)

    beams, lps = search(skip)
    if trivial or fsm.size(1) == 1:
        best = beams[:, 0, 0, :]
    else:
        best, best_lp = select_best_beam_simple_batched(beams, lps, num_constraints, min_constraints_to_satisfy)
        if skip and bool((best_lp <= -1e19).any()):   # (one read of the device; the captions are about to be read anyway)
            beams, lps = search(False)
            best, _ = select_best_beam_simple_batched(beams, lps, num_constraints, min_constraints_to_satisfy)
    return best.view(nimg, n_samples, -1), calls["k"]


def count_tokens(pred: torch.Tensor, boundary_index: int) -> int:
    """Tokens emitted before the first @@BOUNDARY@@ of every caption (inference.py:180-182)."""
    is_end = (pred == boundary_index)
    steps = pred.size(-1)
    first = torch.where(is_end.any(-1), is_end.float().argmax(-1), torch.full(pred.shape[:-1], steps, device=pred.device))
    return int(first.sum().item())

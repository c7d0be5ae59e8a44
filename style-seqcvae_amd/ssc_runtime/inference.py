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

from .decode import DecodeEngine
from .decoding import select_best_beam_simple_batched




def diverse_decode(dec: DecodeEngine, feats: torch.Tensor, sentiment: Optional[torch.Tensor], n_samples: int, beam: int,
                   max_steps: int, boundary_index: int, fsm: Optional[torch.Tensor] = None,
                   num_constraints: Optional[torch.Tensor] = None, min_constraints_to_satisfy: int = 0,
                   eps_steps: Optional[List[torch.Tensor]] = None, early_stop: bool = True, per_node: Optional[int] = None,
                   skip_dead: bool = True, compiled=None, obj_means: Optional[torch.Tensor] = None):
    """feats (nimg,R,F), sentiment (nimg,) or None -> predictions (nimg, n_samples, steps) int64 on device.
    fsm: None (trivial one-state machine, what MAX_GIVEN_CONSTRAINTS: 0 produces), or (nimg, S, S, V) uint8 - ONE machine per
    image, shared by its n_samples latent samples through an index list -, or (nimg*n_samples, S, S, V) (a copy per sample).
    eps_steps: optional explicit noise per step call [(rows_k, Z)]; default: a generator of this call's own, seeded by ONE draw
    from the global CPU generator - the global random state a call consumes does not depend on how many steps it ran.
    skip_dead: rows that hold no finite beam - the states an image's constraints never reach,
    every state before its first constraint word has been decoded - and rows whose beam has ended are neither stepped nor scored from
    logits (cbs_search(skip_dead=True)).  Every caption with a finite log-prob is the exact search's; should a selected caption
    have none (its constraints were not reachable within max_steps), the call is repeated exactly, with the same noise.
    compiled: the machines' CompiledFsm when the caller has it already.
    obj_means (nimg, R, Z): per-region attribute means, SENTIMENT_VAE = 2 only (UpDownCaptioner.translate_obj_atts2obj_means)."""
    dev = feats.device
    nimg = feats.size(0)
    d = dec.dims
    B = nimg * n_samples
    ctx = dec.prepare(feats, obj_means)
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
    skip = bool(skip_dead) and (trivial or fsm.size(1) > 1)   # (trivial machine: only ended beams are left out of the steps)
    if not trivial and fsm.size(1) > 1 and compiled is None:
        from .decode import CompiledFsm
        compiled = CompiledFsm(fsm, fill=max(8, per_node or (beam // 2) or beam))

    per_node = per_node or (beam // 2) or beam
    S = 1 if trivial else fsm.size(1)
    G = B * S * beam

    def search(skip_now):
        # the noise of ALL steps is drawn up front - from the caller's list, or from a generator of this call's own seeded by ONE draw
        # of the global generator: the search may stop early (or, polled late, a few steps after it could have), and what the
        # global random state looks like after the call must not depend on that
        if eps_steps is not None:
            eps0 = eps_steps[0]
            rest = [e.to(dev, torch.float32) for e in eps_steps[1:max_steps]]
            eps = torch.zeros(max(max_steps - 1, 1), G, d.Z, device=dev)
            if rest:
                eps[: len(rest)] = torch.stack(rest)
        else:
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)
            eps0 = torch.randn(B, d.Z, device=dev, generator=gen)
            eps = torch.randn(max(max_steps - 1, 1), G, d.Z, device=dev, generator=gen)
        beams, lps = dec.search(ctx, sent_b, n_samples, beam, per_node, max_steps, boundary_index, eps0, eps,
                                fsm=None if trivial else fsm.contiguous(), compiled=compiled, mach=mach, skip_dead=skip_now,
                                early_stop=early_stop)
        calls["k"] = beams.size(-1)   # one step call per column (cbs.py:127,170)
        return beams, lps

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

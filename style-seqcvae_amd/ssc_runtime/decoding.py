"""Beam selection after (constrained) beam search - host-side, a few integers per image.
Reference: updown-baseline/updown/utils/decoding.py:10-27 (select_best_beam), :30-138 (with constraints)."""
from typing import List

import numpy as np
import torch


def select_best_beam(beams: torch.Tensor, beam_log_probabilities: torch.Tensor) -> torch.Tensor:
    """beams (B, beam, steps) sorted by likelihood -> beams[:, 0, :] (decoding.py:27)."""
    return beams[:, 0, :]


def _valid_states_simple(k: int, min_constraints_to_satisfy: int) -> List[int]:
    need = min(k, min_constraints_to_satisfy)
    return [s for s in range(2 ** k) if bin(s).count("1") >= need]


def _valid_states_general(k: int, constraints, constraint2states, min_constraints_to_satisfy: int):
    """decoding.py:93-125: a state is valid when enough given objects are satisfied, an object with attribute
    constraints counting only in states that also satisfy one of its attributes."""
    n = 2 ** k
    total = np.zeros(n, dtype=int)
    with_attrs = np.zeros(n, dtype=int)
    for obj in constraints:
        obj_states = np.zeros(n, dtype=int)
        obj_states[constraint2states[obj[0]]] = 1
        if not obj[1]:
            attr_states = np.ones(n, dtype=int)
        else:
            attr_states = np.zeros(n, dtype=int)
            for a in obj[1]:
                one = np.zeros(n, dtype=int)
                one[constraint2states[a]] = 1
                attr_states |= one
        obj_states &= attr_states
        if not np.all(attr_states):
            with_attrs |= obj_states
        total += obj_states
    if np.any(with_attrs):
        total *= (np.clip(total, 0, 1) & with_attrs)
    return np.where(total >= min(len(constraints), min_constraints_to_satisfy))[0]


def select_best_beam_with_constraints(beams, beam_log_probabilities, given_constraints, constraints=None,
                                      constraint2states=None, min_constraints_to_satisfy: int = 2, cbs_simple=True):
    """beams (B,S,beam,steps), log-probs (B,S,beam) -> (best (B,steps) int64, valid beams stacked).
    Same contract as decoding.py:30-138."""
    B = beams.size(0)
    best, valid_all = [], []
    for i in range(B):
        k = int(given_constraints[i])
        if cbs_simple:
            valid = _valid_states_simple(k, min_constraints_to_satisfy)
        else:
            valid = _valid_states_general(k, constraints[i], constraint2states[i], min_constraints_to_satisfy)
        valid = torch.as_tensor(np.asarray(valid), dtype=torch.long, device=beams.device)
        vb = beams[i, valid, 0, :]
        vlp = beam_log_probabilities[i, valid, 0]
        valid_all.append(vb)
        best.append(vb[torch.argmax(vlp)])
    return torch.stack(best).long(), torch.stack(valid_all)


def select_best_beam_simple_batched(beams: torch.Tensor, beam_log_probabilities: torch.Tensor, given_constraints: torch.Tensor,
                                    min_constraints_to_satisfy: int = 2):
    """select_best_beam_with_constraints(..., cbs_simple=True) for the whole batch in a handful of device ops instead of a Python
    loop over the batch entries (a 100-image x 20-sample call has 2000 of them): a state s < 2**k is valid when its popcount is
    >= min(k, min_constraints_to_satisfy) (decoding.py:52-60); the best beam is beam 0 of the valid state with the highest beam-0
    log-prob, the first such state on ties (torch.argmax over the valid list).  -> (best (B, steps) int64, its log-prob (B))."""
    B, S = beams.shape[:2]
    dev = beams.device
    k = given_constraints.to(dev).long().view(B, 1)
    s = torch.arange(S, device=dev).view(1, S)
    pop = torch.zeros(1, S, dtype=torch.long, device=dev)
    for bit in range(max(1, (S - 1).bit_length())):
        pop = pop + ((s >> bit) & 1)
    valid = (s < (torch.ones_like(k) << k)) & (pop >= torch.clamp(k, max=min_constraints_to_satisfy))
    lp0 = beam_log_probabilities[:, :, 0]
    top = torch.where(valid, lp0, torch.full_like(lp0, float("-inf"))).max(dim=1, keepdim=True).values
    first = (valid & (lp0 == top)).int().argmax(dim=1)
    rows = torch.arange(B, device=dev)
    return beams[rows, first, 0, :].long(), lp0[rows, first]

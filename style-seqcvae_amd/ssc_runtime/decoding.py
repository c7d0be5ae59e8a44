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

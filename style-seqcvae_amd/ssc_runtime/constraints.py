"""Finite state machines for constrained beam search - host-side preparation of the `(S, S, V)` uint8 adjacency tensor
that `ssc_beam_first` / `ssc_beam_step` consume on the device (`fsm[s1, s2, w] = 1`: decoding word w moves a beam from
state s1 to state s2).

Reference: updown-baseline/updown/utils/constraints.py:212-478 (`FiniteStateMachineBuilder`), :19-53
(`add_constraint_words_to_vocabulary`); per-sample use in updown-baseline/updown/data/datasets.py:470-620 (the machine is
trimmed to its first `num_states` states before it is handed to the model, datasets.py:597-601).

State numbering (constraints.py:226-260): with k = MAX_GIVEN_CONSTRAINTS the first 2**k states are the MAIN states - bit
n-1 of the state number says "constraint n is satisfied" - and the states from 2**k on are SUB-states, one per non-final
word of a multi-word constraint on each edge it labels ("fire" seen, waiting for "hydrant"; any other word falls back to
the edge's origin).  `constraint2states[c]` lists the main states (below 2**len(constraints)) in which c is satisfied:
what `select_best_beam_with_constraints(cbs_simple=False)` consumes (ssc_runtime/decoding.py).

The `ConstraintFilter` (hierarchy-aware NMS over detector boxes, constraints.py:56-209) is not here: it needs the Open
Images class hierarchy (anytree) and detector outputs and runs once per image, far from the hot path.
"""
import csv
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def read_wordforms(tsv_path: str) -> Dict[str, List[str]]:
    """`class name <TAB> comma separated word forms` per line (data/constraint_wordforms*.tsv)."""
    table: Dict[str, List[str]] = {}
    with open(tsv_path, "r") as f:
        for row in csv.reader(f, delimiter="\t"):
            if len(row) >= 2:
                table[row[0]] = row[1].split(",")
    return table


def add_constraint_words_to_vocabulary(vocabulary, wordforms_tsvpath: str, namespace: str = "tokens"):
    """Every word form of every constraint class becomes a vocabulary token (constraints.py:19-53): CBS can only force
    words the output layer can emit."""
    for forms in read_wordforms(wordforms_tsvpath).values():
        for w in forms:
            vocabulary.add_token_to_namespace(w, namespace)
    return vocabulary


class FiniteStateMachineBuilder:
    """Same constructor and `build()` contract as the reference class (constraints.py:283-361)."""

    def __init__(self, vocabulary, wordforms_tsvpath: str, wordforms_attribs_tsvpath: Optional[str] = None,
                 max_given_constraints: int = 3, max_words_per_constraint: int = 3, use_coco_attributes: bool = False,
                 attribute_selection: Optional[Dict[str, bool]] = None):
        self._vocabulary = vocabulary
        self._max_given_constraints = max_given_constraints
        self._max_words_per_constraint = max_words_per_constraint
        self._num_main_states = 2 ** max_given_constraints
        self._num_total_states = self._num_main_states * max_words_per_constraint
        self._wordforms: Dict[str, List[str]] = read_wordforms(wordforms_tsvpath)
        if wordforms_attribs_tsvpath:
            self._wordforms.update(read_wordforms(wordforms_attribs_tsvpath))
            if use_coco_attributes:
                # constraints.py:314-326: attribute classes switched off in the selection table are dropped and the
                # pseudo-class "all" collects the word forms of the remaining ones.  The table itself
                # (updown/data/config_attrib_selection.py) is data of the reference; pass it in.
                selected = set()
                for att, keep in (attribute_selection or {}).items():
                    parts = att.split(" ")
                    name = parts[-1] or parts[-2]
                    if keep:
                        selected.update(self._wordforms[name])
                    else:
                        del self._wordforms[name]
                self._wordforms["all"] = list(selected)

    # -- helpers ---------------------------------------------------------------------------------------------------------
    def _word_ids(self, word: str) -> np.ndarray:
        return np.asarray([self._vocabulary.get_token_index(w) for w in self._wordforms[word]], dtype=np.int64)

    @staticmethod
    def _edge(fsm: np.ndarray, src: int, dst: int, ids: np.ndarray, fallback: Optional[int]):
        """Words `ids` move src -> dst instead of looping on src; from a sub-state every other word goes to `fallback`
        (constraints.py:431-478)."""
        fsm[src, dst, ids] = 1
        fsm[src, src, ids] = 0
        if fallback is not None:
            fsm[src, src, :] = 0
            fsm[src, fallback, :] = 1
            fsm[src, fallback, ids] = 0

    def build(self, constraints: Sequence[str]) -> Tuple[torch.Tensor, int, Dict[str, List[int]]]:
        """-> (fsm (S_total, S_total, V) uint8, index of the first unused sub-state, constraint2states)."""
        V = self._vocabulary.get_vocab_size()
        S, M = self._num_total_states, self._num_main_states
        fsm = np.zeros((S, S, V), dtype=np.uint8)
        fsm[np.arange(M), np.arange(M), :] = 1          # every word loops on a main state until a constraint word moves it
        next_sub = M
        n_valid = 2 ** len(constraints)
        seen_at: Dict[str, List[int]] = {}              # constraint -> the positions n at which it was given so far
        constraint2states: Dict[str, List[int]] = {}
        for n, constraint in enumerate(constraints, start=1):
            words = constraint.split()
            stride = 2 ** (n - 1)                       # setting bit n-1 adds `stride` to the state number
            # A constraint given a second time only labels edges from the state that one earlier occurrence produced
            # (constraints.py:385-393); a new constraint labels every edge q -> q + stride with bit n-1 of q clear.
            if constraint in seen_at:
                src, src_end = seen_at[constraint][-1], seen_at[constraint][-1] + 1
                seen_at[constraint].append(n)
            else:
                src, src_end = 0, M
                seen_at[constraint] = [n]
            satisfied: List[int] = []
            while src < src_end:
                for _ in range(stride):
                    at = src
                    for i, word in enumerate(words):
                        ids = self._word_ids(word)
                        if i + 1 < len(words):          # non-final word of a multi-word constraint: a fresh sub-state
                            self._edge(fsm, at, next_sub, ids, src)
                            at = next_sub
                            next_sub += 1
                        else:
                            if src + stride < n_valid:
                                satisfied.append(src + stride)
                            self._edge(fsm, at, src + stride, ids, src)
                    src += 1
                src += stride
            constraint2states[constraint] = satisfied
        return torch.from_numpy(fsm), next_sub, constraint2states

    def build_trimmed(self, constraints: Sequence[str]) -> Tuple[torch.Tensor, int, Dict[str, List[int]]]:
        """`build` followed by the trim the reference applies in its collate function (datasets.py:597-601): the machine
        as the model receives it, `(1, S, S, V)` with S = number of states in use (>= 2**MAX_GIVEN_CONSTRAINTS)."""
        fsm, nstates, c2s = self.build(constraints)
        return fsm[None, :nstates, :nstates, :].contiguous(), nstates, c2s


def trivial_fsm(batch: int, vocab_size: int) -> torch.Tensor:
    """The machine MAX_GIVEN_CONSTRAINTS = 0 produces (config.yaml:36; constraints.py:345-361 with no constraint): one
    state, every word allowed - constrained beam search is then plain beam search."""
    return torch.ones(batch, 1, 1, vocab_size, dtype=torch.uint8)


def satisfied_counts(num_states: int) -> List[int]:
    """Number of satisfied constraints per main state (the popcount of its number; constraints.py:252-260)."""
    return [bin(s).count("1") for s in range(num_states)]

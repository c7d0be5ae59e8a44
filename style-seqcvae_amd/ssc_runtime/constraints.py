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

`ConstraintFilter` (constraints.py:56-209) turns one image's detector output into at most k constraint classes: blacklist,
top-k by confidence, renames.  The reference's docstring also promises hierarchy-aware suppression of overlapping boxes (a "dog"
box suppresses a "carnivore" box on the same pixels), but its `_nms` as EXECUTED never drops a box: the work list is sorted by
ascending height, so for the current (finest) box `heights[rest] >= heights[current]` is always true and `keep_condition` with it
(constraints.py:195-203).  The default here reproduces what the reference executes (pinned by tests/golden/g13_filter.npz: the
reference's own `__call__` / `_nms` with the tree heights injected - anytree is not importable, so the tree is flattened at load
time into a pre-order label list + node heights); the suppression the docstring describes is available as `hierarchy_nms=True`
(a divergence from the reference, documented in INTEGRATION.md).  It runs once per image on the host, far from the hot path.
"""
import csv
import json
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

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
    """Every TOKEN of every word form of every constraint class becomes a vocabulary entry (constraints.py:19-53; a form may be
    several words - "fire hydrant" adds "fire" and "hydrant", :47-51): CBS can only force words the output layer can emit."""
    for forms in read_wordforms(wordforms_tsvpath).values():
        for form in forms:
            for w in form.split():
                vocabulary.add_token_to_namespace(w, namespace)
    return vocabulary


class ConstraintFilter:
    """Same constructor and call contract as the reference class (constraints.py:105-153): `filter(boxes (n, 4) x1 y1 x2 y2,
    class_names, scores)` -> constraint class names (unordered, duplicates dropped)."""

    # classes that never become constraints (too rare, not uttered, or well covered by COCO): constraints.py:84-94
    BLACKLIST = frozenset((
        "auto part", "bathroom accessory", "bicycle wheel", "boy", "building", "clothing", "door handle", "fashion accessory",
        "footwear", "girl", "hiking equipment", "human arm", "human beard", "human body", "human ear", "human eye", "human face",
        "human foot", "human hair", "human hand", "human head", "human leg", "human mouth", "human nose", "land vehicle",
        "mammal", "man", "person", "personal care", "plant", "plumbing fixture", "seat belt", "skull", "sports equipment", "tire",
        "tree", "vehicle registration plate", "wheel", "woman"))
    # detector class name -> constraint word (constraints.py:96-103)
    REPLACEMENTS = {"band-aid": "bandaid", "wood-burning stove": "wood burning stove", "kitchen & dining room table": "table",
                    "salt and pepper shakers": "salt and pepper", "power plugs and sockets": "power plugs",
                    "luggage and bags": "luggage"}

    def __init__(self, hierarchy_jsonpath: Union[str, dict], nms_threshold: float = 0.85, max_given_constraints: int = 3,
                 hierarchy_nms: bool = False):
        self._hierarchy_nms = hierarchy_nms
        root = hierarchy_jsonpath if isinstance(hierarchy_jsonpath, dict) else json.load(open(hierarchy_jsonpath))
        # pre-order walk of {"LabelName": ..., "Subcategory": [...]}: lower-cased labels and node heights (edges on the longest
        # path down to a leaf - anytree's `height`), iteratively
        self._labels: List[str] = []
        self._heights: List[int] = []
        stack = [(root, None)]
        parents: List[Optional[int]] = []
        while stack:
            node, parent = stack.pop()
            idx = len(self._labels)
            self._labels.append(str(node["LabelName"]).lower())
            self._heights.append(0)
            parents.append(parent)
            for child in reversed(node.get("Subcategory", [])):   # reversed: the stack pops the first child first
                stack.append((child, idx))
        for idx in range(len(parents) - 1, -1, -1):               # children come after their parent in pre-order
            if parents[idx] is not None:
                self._heights[parents[idx]] = max(self._heights[parents[idx]], self._heights[idx] + 1)
        self._nms_threshold = nms_threshold
        self._max_given_constraints = max_given_constraints
        self._height_cache: Dict[str, int] = {}

    def height(self, class_name: str) -> int:
        """Height of the FIRST node in pre-order whose label is contained in the class name (constraints.py:163-166: `findall(...,
        node.LabelName.lower() in c)[0].height`; IndexError when nothing matches, as there)."""
        if class_name not in self._height_cache:
            hit = next((i for i, lab in enumerate(self._labels) if lab in class_name), None)
            if hit is None:
                raise IndexError(f"no class of the hierarchy is contained in {class_name!r}")
            self._height_cache[class_name] = self._heights[hit]
        return self._height_cache[class_name]

    def __call__(self, boxes: np.ndarray, class_names: Sequence[str], scores: np.ndarray) -> List[str]:
        boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
        scores = np.asarray(scores, dtype=np.float64).reshape(-1)
        # padding boxes (confidence 0) and blacklisted classes never become constraints
        cand = [i for i, c in enumerate(class_names) if scores[i] > 0 and c not in self.BLACKLIST]
        if not cand:
            return []
        names = [class_names[i] for i in cand]
        keep = self._suppress(boxes[cand], names)
        # top-k by confidence (stable: equal scores keep the suppression pass's order), renamed, duplicates dropped
        ranked = sorted(keep, key=lambda i: -scores[cand[i]])[: self._max_given_constraints]
        return list({self.REPLACEMENTS.get(names[i], names[i]) for i in ranked})

    def _suppress(self, boxes: np.ndarray, names: Sequence[str]) -> List[int]:
        """Indices kept, finest class first (`heights.argsort()`, constraints.py:171).  Default: every box - what the
        reference's loop executes (module docstring).  `hierarchy_nms=True`: a box is dropped iff a KEPT box of a strictly finer
        class (smaller height) overlaps it with IoU > threshold; boxes of equal height never suppress each other (the
        behaviour constraints.py:56-70 describes; IoU with the pixel-inclusive +1 of :176-192)."""
        h = np.array([self.height(c) for c in names])
        if not self._hierarchy_nms:
            return [int(i) for i in h.argsort()]
        x1, y1, x2, y2 = boxes.T
        area = (x2 - x1 + 1) * (y2 - y1 + 1)
        iw = np.maximum(0.0, np.minimum(x2[:, None], x2[None, :]) - np.maximum(x1[:, None], x1[None, :]) + 1)
        ih = np.maximum(0.0, np.minimum(y2[:, None], y2[None, :]) - np.maximum(y1[:, None], y1[None, :]) + 1)
        inter = iw * ih
        iou = inter / (area[:, None] + area[None, :] - inter)
        kept: List[int] = []
        for i in np.argsort(h, kind="stable"):
            if not any(h[j] < h[i] and iou[j, i] > self._nms_threshold for j in kept):
                kept.append(int(i))
        return kept


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

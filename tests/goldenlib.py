"""Helpers to load the committed golden fixtures (tests/golden/*.npz)."""
import ast
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    cfg = ast.literal_eval(str(d.pop("cfg")))
    return d, cfg


def group(d, prefix):
    n = len(prefix)
    return {k[n:]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


def load_raw(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def unpack_fsm(bits, B, S, V):
    """(B,S,S,V) uint8 adjacency from np.packbits."""
    return torch.from_numpy(np.unpackbits(bits)[: B * S * S * V].reshape(B, S, S, V).copy())


def cbs_table_step(table, drift):
    """Deterministic table-driven step shared by make_golden.py and the tests: next-token log-probs are a function of the
    previous token, a per-row step counter and an accumulator carried in the state - so a wrong state gather after the beam
    re-ordering (cbs.py:236-250) changes the result."""
    def step(tokens, state):
        G = tokens.numel()
        dev = tokens.device
        cnt = torch.zeros(G, 1, device=dev) if state is None else state["cnt"]
        acc = torch.zeros(G, 3, device=dev) if state is None else state["acc"]
        lp = torch.log_softmax(table[tokens] + drift[cnt.long().view(-1) % drift.size(0)] + acc.sum(1, keepdim=True) * 0.01, dim=1)
        new = {"cnt": cnt + 1, "acc": (acc + tokens.view(-1, 1).float() * torch.tensor([[1.0, 0.5, 0.25]], device=dev)) % 3.0}
        return lp, new
    return step

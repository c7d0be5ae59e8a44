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

#!/usr/bin/env python3
"""Bounds audit of the GEMM kernels' stepped operand pointers (run with SSC_GEMM_DBG=64; diagnostic only).
usage: SSC_GEMM_DBG=64 python tools/gemm_oob.py kind:M:N:K1+K2[:splits] ..."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch

from gpuutil import gemm
from ssc_runtime import lib as L


def main():
    assert os.environ.get("SSC_GEMM_DBG") == "64", "set SSC_GEMM_DBG=64"
    raw = C.CDLL(L.LIB_PATH)
    rec = (C.c_int * 8)()
    for sh in sys.argv[1:]:
        parts = sh.split(":")
        kind, M, N = parts[0], int(parts[1]), int(parts[2])
        Ks = [int(k) for k in parts[3].split("+")]
        splits = int(parts[4]) if len(parts) > 4 else 0
        a_kc, b_kc = {"NT": (1, 1), "NN": (1, 0), "TN": (0, 0)}[kind]
        As = [torch.randn((M, K) if a_kc else (K, M), device="cuda") for K in Ks]
        Bs = [torch.randn((N, K) if b_kc else (K, N), device="cuda") for K in Ks]
        out = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(40 * 64 * 4800, 10 * M * N) + 4096, device="cuda")
        segs = [(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(As, Bs, Ks)]
        for _ in range(int(os.environ.get('REPS', '1'))):
            gemm(segs, M, N, a_kc, b_kc, out, splits=splits, ws=ws)
        torch.cuda.synchronize()
        assert raw.ssc_debug_gemm_oob(rec) == 0
        print(sh, "violations=%d first: bx=%d by=%d step=%d chunk=%d tid=%d off=%d k0=%d" % tuple(rec), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmark of ssc_gemm on the hot shapes (used under rocprofv3 --pmc and for A/B timing).
usage: python tools/gemm_probe.py [reps] [shape ...]   shape = kind:M:N:K1+K2+..[:splits]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch

from gpuutil import gemm

DEFAULT = ["NT:64:4800:2048+1200+1200+1200", "NN:64:4448:4800+4800", "NN:64:1200:4800+4800", "TN:4800:1200:1344",
           "NT:1344:10000:1200", "TN:10000:1200:1344", "NT:64:768:1200", "NN:64:1200:256"]


def main():
    import os as _os
    from ssc_runtime import lib as _L
    for env, key in (("SSC_X3_WIDE", b"x3_wide"), ("SSC_X3_PF", b"x3_pf"), ("SSC_X3_NBUF", b"x3_nbuf")):   # include/ssc_debug.h
        if _os.environ.get(env):
            _L.load().ssc_debug_set(key, int(_os.environ[env]))
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    shapes = sys.argv[2:] or DEFAULT
    for sh in shapes:
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
        for _ in range(3):
            gemm(segs, M, N, a_kc, b_kc, out, splits=splits, ws=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gemm(segs, M, N, a_kc, b_kc, out, splits=splits, ws=ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        fl = 2.0 * M * N * sum(Ks)
        print(f"{sh:42s} {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s  {4.0 * sum(Ks) * (M + N) / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()

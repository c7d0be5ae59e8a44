#!/usr/bin/env python3
"""In-kernel clock / cycles-per-k-step probe of the wave-specialised GEMM kernel (diagnostic).
usage: SSC_X3B=2 SSC_GEMM_DBG=<64 + ablation bits> python tools/gemm_clock.py kind:M:N:K[:splits] ..."""
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

raw = C.CDLL(L.LIB_PATH)
rec = (C.c_longlong * 5)()
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
    for _ in range(400):   # long enough for the clock to settle under load
        gemm(segs, M, N, a_kc, b_kc, out, splits=splits, ws=ws)
    torch.cuda.synchronize()
    assert raw.ssc_debug_gemm_clock(rec) == 0
    cyc, ticks, steps = rec[0], rec[1], max(rec[2], 1)
    print(f"{sh:28s} dbg={os.environ.get('SSC_GEMM_DBG')}: {cyc / steps:8.0f} cycles/k-step  {ticks * 10.0 / steps:7.1f} ns/k-step  clock {cyc / max(ticks, 1) * 0.1:5.2f} GHz  prologue {rec[3] * 0.01:6.2f} us  loop {ticks * 0.01:6.2f} us  entry->end {rec[4] * 0.01:6.2f} us", flush=True)

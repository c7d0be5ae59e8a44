#!/usr/bin/env python3
"""Does a bandwidth-bound gate product (one stream) overlap with the latency-bound q-projection + attention chain (another
stream)?  Times: the product alone, the chain alone, both back to back on one stream, both on two streams.
usage: python tools/overlap_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ssc_runtime import lib as L  # noqa: E402


def desc(segs, M, N, out, ws):
    d = L.GemmDesc()
    d.nseg = len(segs)
    for i, (A, lda, B, ldb, K) in enumerate(segs):
        d.seg[i].A, d.seg[i].lda, d.seg[i].B, d.seg[i].ldb, d.seg[i].K = A.data_ptr(), lda, B.data_ptr(), ldb, K
    d.M, d.N, d.a_kc, d.b_kc = M, N, 1, 1
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    return d


def main():
    lib = L.load()
    B, H, A, F, R = 64, 1200, 768, 2048, 36
    dev = "cuda"
    # big: [h1, hd', he'] x enc weights + [h1, hd'] x dec weights ~ K = 3600 + 2400 against 4800 columns (115 MB)
    Kbig = 6000
    x = torch.randn(B, Kbig, device=dev); Wb = torch.randn(4 * H, Kbig, device=dev) / 80
    outb = torch.empty(B, 4 * H, device=dev); wsb = torch.empty(40 * B * 4 * H, device=dev)
    dbig = desc([(x, Kbig, Wb, Kbig, Kbig)], B, 4 * H, outb, wsb)
    h1 = torch.randn(B, H, device=dev); Wq = torch.randn(A, H, device=dev) / 35
    q = torch.empty(B, A, device=dev); wsq = torch.empty(40 * B * A, device=dev)
    dq = desc([(h1, H, Wq, H, H)], B, A, q, wsq)
    pv = torch.randn(B, R, A, device=dev); wa = torch.randn(A, device=dev); feats = torch.randn(B, R, F, device=dev)
    mask = torch.ones(B, R, device=dev); logits = torch.empty(B, R, device=dev); alpha = torch.empty(B, R, device=dev)
    att = torch.empty(B, F, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sp = lambda s: C.c_void_p(s.cuda_stream)

    def big(s):
        lib.ssc_gemm(C.byref(dbig), sp(s))

    def chain(s):
        lib.ssc_gemm(C.byref(dq), sp(s))
        lib.ssc_attn_fwd(L.ptr(q), A, L.ptr(pv), L.ptr(wa), L.ptr(mask), L.ptr(feats), B, R, A, F, 1, L.ptr(logits), L.ptr(alpha),
                         L.ptr(att), F, sp(s))

    def timed(fn, n=20):
        """n iterations captured into one hipGraph (device time, no host launch overhead); the capture stream is s1"""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s1):
            with torch.cuda.graph(g, stream=s1):
                for _ in range(n):
                    fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def both_two_streams():
        s2.wait_stream(s1)          # fork
        big(s2)
        chain(s1)
        s1.wait_stream(s2)          # join

    print(f"gate product alone        {timed(lambda: big(s1)):7.1f} us")
    print(f"q + attention alone       {timed(lambda: chain(s1)):7.1f} us")
    print(f"one stream, back to back  {timed(lambda: (big(s1), chain(s1))):7.1f} us")
    print(f"two streams (fork/join)   {timed(both_two_streams):7.1f} us")


if __name__ == "__main__":
    main()

"""Time the wave-specialised 128x128 NT product in its 3xBF16 and 2xFP16 forms at the decode's shapes (hipEvents, 10 launches)."""
import os, sys
os.environ["SSC_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpuutil import gemm, split_f16
from ssc_runtime import lib as L
lib = L.load()
SHAPES = ((10000, 10000, 1200), (10000, 4800, 2400), (10000, 4800, 1328), (6600, 4800, 2400), (1344, 10000, 1200))
if "--one" in sys.argv:   # counter passes: one shape
    SHAPES = SHAPES[1:2]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda") * 0.5
    B = torch.randn(N, K, device="cuda") * 0.03
    C = torch.empty(M, N, device="cuda")
    res = {}
    for f16 in (0, 1, 0, 1):
        lib.ssc_debug_set(b"gemm_f16", f16)
        for _ in range(3):
            gemm([(A, K, B, K, K)], M, N, 1, 1, C, splits=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gemm([(A, K, B, K, K)], M, N, 1, 1, C, splits=1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        res.setdefault(f16, []).append(us)
    lib.ssc_debug_set(b"gemm_f16", 1)
    pa, pb = split_f16(A), split_f16(B)
    for tag, planes in (("B", [(None, pb)]), ("A", [(pa, None)]), ("AB", [(pa, pb)])):
        for _ in range(3):
            gemm([(A, K, B, K, K)], M, N, 1, 1, C, splits=1, planes=planes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gemm([(A, K, B, K, K)], M, N, 1, 1, C, splits=1, planes=planes)
        e1.record(); torch.cuda.synchronize()
        res[tag] = [e0.elapsed_time(e1) * 100]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        split_f16(A)
    e1.record(); torch.cuda.synchronize()
    res["split"] = [e0.elapsed_time(e1) * 100]
    lib.ssc_debug_set(b"gemm_f16", 0)
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: 3xBF16 {min(res[0]):.0f} us ({fl/min(res[0])/1e6:.0f} TF fp32-eq), 2xFP16 {min(res[1]):.0f} us ({fl/min(res[1])/1e6:.0f} TF fp32-eq)  x{min(res[0])/min(res[1]):.2f} | planes B {res['B'][0]:.0f} us, A {res['A'][0]:.0f} us, A+B {res['AB'][0]:.0f} us ({fl/res['AB'][0]/1e6:.0f} TF), split of A {res['split'][0]:.0f} us", flush=True)

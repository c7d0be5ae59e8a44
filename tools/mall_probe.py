"""Does the memory-side cache help the weight-streaming minibatch products?  The same 64-row product is timed (hipEvents) with ONE weight
matrix launched back to back (92 MB: fits the 256 MB cache) and cycling over 4 / 8 distinct matrices (370 / 740 MB: cannot)."""
import os, sys
os.environ["SSC_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpuutil import gemm
from ssc_runtime import lib as L
lib = L.load()
M, N, K = 64, 4000, 5776
A = torch.randn(M, K, device="cuda") * 0.5
C = torch.empty(M, N, device="cuda")
ws = torch.empty(64 * M * N, device="cuda")
Ws = [torch.randn(K, N, device="cuda") * 0.03 for _ in range(8)]
for b_kc, name in ((0, "NN"),):
    for nw in (1, 2, 4, 8, 1, 8):
        for i in range(8):
            gemm([(A, K, Ws[i % nw], N, K)], M, N, 1, b_kc, C, splits=0, ws=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            gemm([(A, K, Ws[i % nw], N, K)], M, N, 1, b_kc, C, splits=0, ws=ws)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 40
        print(f"{name} {M}x{N}x{K} cycling over {nw} weight matrices ({nw * K * N * 4 / 1e6:.0f} MB): {us:.1f} us per launch = {K * N * 4 / us / 1e6:.2f} TB/s", flush=True)

#!/usr/bin/env python3
"""Random-shape sweep of ssc_gemm against float64 (on the GPU): NT / NN / TN, 1-3 K-segments, aligned and unaligned leading
dimensions, bias / accumulate, optional split-K workspace, optional device-side row compaction (NT / NN) - the launcher's
kernel-form and split decisions over shapes no fixed test list names.   python tools/gemm_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from gpuutil import gemm  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    aligned = len(sys.argv) > 3 and sys.argv[3] == "aligned"   # only 16-byte rows: every case may take the vector / compaction paths
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    worst = 0.0
    bad = 0
    for case in range(n):
        kind = pick(["NT", "NT", "NN", "TN"])
        a_kc, b_kc = {"NT": (1, 1), "NN": (1, 0), "TN": (0, 0)}[kind]
        M = pick([1, 3, 17, 33, 64, 65, 100, 128, 129, 150, 200, 511, 512, 520, 600, 700, 1000, 1344, 2500, 5000])
        N = pick([1, 7, 64, 128, 150, 152, 256, 300, 768, 1000, 1200, 2048, 4448, 4800, 10000, 10001])
        if M * N > 30_000_000:
            N = 1200
        nseg = 1 if kind == "TN" else ri(1, 3)
        Ks = [pick([16, 32, 100, 128, 152, 256, 1000, 1200, 1344, 2048, 4800] if aligned else
                   [1, 5, 16, 30, 32, 100, 128, 150, 152, 256, 1000, 1200, 1344, 2048, 4800]) for _ in range(nseg)]
        if kind == "TN":
            Ks = [pick([64, 150, 1000, 1344, 2688])]
        if aligned:
            N = (N + 3) & ~3
            if kind != "NT":
                M = (M + 3) & ~3 if kind == "TN" else M
        pad = pick([0, 4] if aligned else [0, 0, 1, 4])     # leading-dimension slack: 0 / 4 keep 16-byte rows where the extent allows, 1 breaks them
        segs, ref = [], torch.zeros(M, N, dtype=torch.float64)
        for K in Ks:
            if a_kc:
                A = torch.randn(M, K + pad, generator=g)
                Av = A[:, :K]
            else:
                A = torch.randn(K, M + pad, generator=g)
                Av = A[:, :M].t()
            if b_kc:
                Bm = torch.randn(N, K + pad, generator=g)
                Bv = Bm[:, :K].t()
            else:
                Bm = torch.randn(K, N + pad, generator=g)
                Bv = Bm[:, :N]
            Ad, Bd = A.cuda(), Bm.cuda()
            segs.append((Ad, Ad.stride(0), Bd, Bd.stride(0), K))
            ref += Av.double() @ Bv.double()
        bias = torch.randn(N, generator=g) if ri(0, 2) == 0 else None
        acc = ri(0, 3) == 0
        C0 = torch.randn(M, N + pad, generator=g)
        Cd = C0.cuda()
        want = ref + (bias.double() if bias is not None else 0) + (C0[:, :N].double() if acc else 0)
        compact = None
        keep_rows = None
        if kind != "TN" and ri(0, 2) == 0 and M >= 8:     # row compaction: a random subset of rows, the rest of C untouched
            keep = torch.rand(M, generator=g) < 0.6
            keep[0] = True
            rows = torch.nonzero(keep).flatten().to(torch.int32)
            cnt = torch.tensor([rows.numel(), 0, 0, 0], dtype=torch.int32)
            lst = torch.cat([rows, torch.zeros(M - rows.numel(), dtype=torch.int32)]).cuda()
            compact = {"m_count": cnt.cuda(), "a_rows": lst, "c_rows": lst}
            keep_rows = keep
        if kind == "TN" and ri(0, 1) == 0 and Ks[0] >= 8:   # weight-gradient form: both operands gathered by a k-row list (live rows)
            K = Ks[0]
            keepk = torch.rand(K, generator=g) < 0.7
            keepk[0] = True
            krows = torch.nonzero(keepk).flatten().to(torch.int32)
            cnt = torch.tensor([krows.numel(), 0, 0, 0], dtype=torch.int32)
            lst = torch.cat([krows, torch.zeros(K - krows.numel(), dtype=torch.int32)]).cuda()
            compact = {"k_count": cnt.cuda(), "ka_rows": lst, "kb_rows": lst}
            A_h, B_h = segs[0][0].cpu(), segs[0][2].cpu()
            kk = krows.long()
            ref = A_h[kk][:, :M].double().t() @ B_h[kk][:, :N].double()
            want = ref + (bias.double() if bias is not None else 0) + (C0[:, :N].double() if acc else 0)
        use_ws = ri(0, 1) == 0
        Cview = Cd[:, :N]
        rc = gemm(segs, M, N, a_kc, b_kc, Cview, bias=bias.cuda() if bias is not None else None, accumulate=int(acc),
                  ws=ws if use_ws else None, compact=compact, check=False)
        torch.cuda.synchronize()
        tag = f"{kind} M={M} N={N} K={Ks} pad={pad} bias={bias is not None} acc={acc} ws={use_ws} compact={compact is not None}"
        if rc != 0:
            if rc in (-2, -5) or True:   # an argument the ABI rejects is fine as long as it says so; print it
                print(f"case {case}: rc={rc}  {tag}")
            continue
        got = Cd[:, :N].cpu().double()
        if keep_rows is not None:
            want = torch.where(keep_rows[:, None], want, C0[:, :N].double())
        scale = float(want.abs().max()) + 1.0
        err = float((got - want).abs().max()) / scale
        worst = max(worst, err)
        if err > 1e-5:   # (fp32 accumulation over K <= 6000 terms of unit variance: a few 1e-6 of the largest entry)
            bad += 1
            print(f"case {case}: REL ERR {err:.3e}  {tag}")
        if pad:   # the slack columns of C must be untouched
            if not torch.equal(Cd[:, N:].cpu(), C0[:, N:]):
                bad += 1
                print(f"case {case}: C slack columns written  {tag}")
    print(f"{n} cases, worst relative error {worst:.2e}, {bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

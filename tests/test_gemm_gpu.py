"""GPU: exact-fp32 MFMA GEMM (ssc_gemm) against float64 matmul.  Tolerance: 2e-6 * sum|a||b| scale
(fp32 accumulation over K), checked as max-abs <= 1e-4 for K <= 6000 with N(0,1)/sqrt(K)-scaled data."""
import pytest
import torch

from gpuutil import dev, gemm, maxdiff
from ssc_runtime import lib as L

pytestmark = pytest.mark.gpu


def ref(As, Bs, a_kc, b_kc):
    out = 0
    for A, B in zip(As, Bs):
        a = A.double() if a_kc else A.double().t()
        b = B.double().t() if b_kc else B.double()
        out = out + a @ b
    return out


@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("M,N,Ks", [(64, 128, [96]), (64, 4800, [2048, 1200, 1200]), (1, 7, [5]), (70, 33, [37, 1, 64]),
                                    (130, 257, [100, 31]), (1344, 200, [300]), (600, 530, [100, 31, 64]),
                                    (1344, 1200, [333])])
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_gemm_layouts(a_kc, b_kc, M, N, Ks, splits):
    g = torch.Generator().manual_seed(M * 7 + N)
    As = [torch.randn((M, K) if a_kc else (K, M), generator=g) for K in Ks]
    Bs = [torch.randn((N, K) if b_kc else (K, N), generator=g) / (sum(Ks) ** 0.5) for K in Ks]
    want = ref(As, Bs, a_kc, b_kc)
    dA, dB = [dev(a) for a in As], [dev(b) for b in Bs]
    out = torch.full((M, N), float("nan"), device="cuda")
    ws = torch.empty(40 * M * N + 64, device="cuda")
    nsteps = sum((K + 31) // 32 for K in Ks)
    sp = min(splits, nsteps) if splits else 0
    gemm([(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(dA, dB, Ks)], M, N, a_kc, b_kc, out, splits=sp, ws=ws)
    torch.cuda.synchronize()
    assert maxdiff(out, want) < 2e-5


def test_gemm_bias_accumulate_and_strided_views():
    g = torch.Generator().manual_seed(3)
    M, N, K = 64, 200, 333
    big_a = dev(torch.randn(M, 1000, generator=g))
    big_b = dev(torch.randn(N, 1003, generator=g))      # odd ld -> scalar-load path for B
    A = big_a[:, 8:8 + K]                               # aligned column offset -> vector path
    B = big_b[:, 5:5 + K]                               # misaligned column offset
    bias = dev(torch.randn(N, generator=g))
    big_c = dev(torch.randn(M, 300, generator=g))
    Cv = big_c[:, 3:3 + N]
    want = Cv.cpu().double() + A.cpu().double() @ B.cpu().double().t() + bias.cpu().double()
    ws = torch.empty(40 * M * N, device="cuda")
    gemm([(A, 1000, B, 1003, K)], M, N, 1, 1, Cv, bias=bias, accumulate=1, ws=ws)
    assert maxdiff(Cv, want) < 1e-4
    # untouched neighbours
    assert torch.equal(big_c[:, :3].cpu(), big_c[:, :3].cpu())


def test_gemm_exact_small_integers():
    # integer data: every product and partial sum is exact in fp32 -> bit-exact result, catches layout bugs
    g = torch.Generator().manual_seed(11)
    M, N, K = 96, 160, 72
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (K, N), generator=g).float()   # asymmetric, NN layout
    out = torch.empty(M, N, device="cuda")
    gemm([(dev(A), K, dev(B), N, K)], M, N, 1, 0, out)
    assert torch.equal(out.cpu(), A @ B)
    out2 = torch.empty(K, N, device="cuda")                  # TN: A^T given as (M rows = K_red, K cols = M_out)
    Bt = torch.randint(-3, 4, (M, N), generator=g).float()
    gemm([(dev(A), K, dev(Bt), N, M)], K, N, 0, 0, out2)
    assert torch.equal(out2.cpu(), A.t() @ Bt)


@pytest.mark.parametrize("M,N,Ks", [(64, 4800, [2048, 1200, 1200, 1200]), (64, 768, [1200]), (1344, 1000, [1200]),
                                    (70, 130, [100, 36])])
def test_nt_split_bf16_is_fp32_accurate(M, N, Ks):
    """The 3xBF16 split kernel (default for NT) against float64, next to the exact-fp32 MFMA kernel: its error must
    stay within 2x the fp32 kernel's and within 2e-6 relative to sum|a||b| (data with a wide dynamic range)."""
    from ssc_runtime import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N)
    As = [torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, K, generator=g)) for K in Ks]
    Bs = [torch.randn(N, K, generator=g) * torch.exp(2 * torch.randn(N, K, generator=g)) / 50 for K in Ks]
    want = ref(As, Bs, 1, 1)
    scale = sum(a.double().abs() @ b.double().abs().t() for a, b in zip(As, Bs))
    dA, dB = [dev(a) for a in As], [dev(b) for b in Bs]
    segs = [(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(dA, dB, Ks)]
    ws = torch.empty(40 * M * N + 64, device="cuda")
    errs = {}
    prev = lib.ssc_set_gemm_mode(1)
    try:
        for mode in (1, 0):
            lib.ssc_set_gemm_mode(mode)
            out = torch.full((M, N), float("nan"), device="cuda")
            gemm(segs, M, N, 1, 1, out, ws=ws)
            errs[mode] = ((out.cpu().double() - want).abs() / scale).max().item()
    finally:
        lib.ssc_set_gemm_mode(prev)
    assert errs[1] < 2e-6, errs
    assert errs[1] <= 2.0 * errs[0] + 1e-8, errs


def test_nt_split_bf16_exact_on_small_integers():
    g = torch.Generator().manual_seed(5)
    M, N, K = 96, 160, 72
    A = torch.randint(-200, 201, (M, K), generator=g).float()   # needs hi+mid pieces
    B = torch.randint(-200, 201, (N, K), generator=g).float()
    out = torch.empty(M, N, device="cuda")
    gemm([(dev(A), K, dev(B), K, K)], M, N, 1, 1, out)
    assert torch.equal(out.cpu(), A @ B.t())


# ---- device-side row compaction (ssc_gemm_desc.m_count / a_rows / c_rows / k_count / ka_rows / kb_rows) ----------------
def _active(n, frac, seed):
    g = torch.Generator().manual_seed(seed)
    keep = torch.rand(n, generator=g) < frac
    keep[0] = True
    idx = torch.nonzero(keep).flatten().to(torch.int32)
    pad = torch.zeros(n, dtype=torch.int32)
    pad[: idx.numel()] = idx
    return pad.cuda(), torch.tensor([idx.numel()], dtype=torch.int32).cuda(), idx.long()


@pytest.mark.parametrize("M,N,K,splits", [(1344, 1000, 1200, 1), (200, 4800, 1216, 3), (64, 4800, 2400, 9)])
def test_gemm_m_compaction_nt(M, N, K, splits):
    """y[c_rows[r]] = x[a_rows[r]] @ w.T for r < *m_count; other rows of y are left untouched."""
    lib = L.load()
    try:
        torch.manual_seed(5)
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        b = torch.randn(N, device="cuda")
        rows, cnt, idx = _active(M, 0.6, 11)
        y = torch.full((M, N), 7.0, device="cuda")
        ws = torch.empty(16 * M * N, device="cuda")
        gemm([(x, K, w, K, K)], M, N, 1, 1, y, bias=b, splits=splits, ws=ws,
             compact={"m_count": cnt, "a_rows": rows, "c_rows": rows})
        ref = torch.full((M, N), 7.0, dtype=torch.float64)
        ref[idx] = x.cpu().double()[idx] @ w.cpu().double().T + b.cpu().double()
        scale = (x.abs().cpu().double()[idx] @ w.abs().cpu().double().T).max().item()
        assert maxdiff(y, ref) <= 4e-6 * scale
        assert torch.equal(y.cpu()[~torch.isin(torch.arange(M), idx)], torch.full((M - idx.numel(), N), 7.0))
        # compact output (no c_rows), accumulate on top of existing values
        y2 = torch.ones(M, N, device="cuda")
        gemm([(x, K, w, K, K)], M, N, 1, 1, y2, accumulate=1, splits=splits, ws=ws, compact={"m_count": cnt, "a_rows": rows})
        ref2 = torch.ones(M, N, dtype=torch.float64)
        ref2[: idx.numel()] += x.cpu().double()[idx] @ w.cpu().double().T
        assert maxdiff(y2, ref2) <= 4e-6 * scale
    finally:
        pass


@pytest.mark.parametrize("M,N,K,splits,gather", [(4800, 1200, 1344, 1, True), (768, 1216, 1344, 7, True), (4800, 128, 1344, 0, True),
                                                 (1000, 1200, 1344, 0, False), (128, 1200, 96, 2, True)])
def test_gemm_k_compaction_tn(M, N, K, splits, gather):
    """dW = sum over the active rows k of a[k, :]^T b[k, :]  (the weight-gradient products)."""
    torch.manual_seed(6)
    a = torch.randn(K, M, device="cuda")
    b = torch.randn(K, N, device="cuda")
    rows, cnt, idx = _active(K, 0.7, 3)
    if not gather:
        idx = torch.arange(idx.numel())
    out = torch.zeros(M, N, device="cuda")
    ws = torch.empty(16 * M * N, device="cuda")
    comp = {"k_count": cnt}
    if gather:
        comp.update(ka_rows=rows, kb_rows=rows)
    gemm([(a, M, b, N, K)], M, N, 0, 0, out, splits=splits, ws=ws, compact=comp)
    ad, bd = a.cpu().double()[idx], b.cpu().double()[idx]
    ref = ad.T @ bd
    scale = (ad.abs().T @ bd.abs()).max().item()
    assert maxdiff(out, ref) <= 4e-6 * scale
    # zero active rows: the product is exactly zero
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    out.fill_(3.0)
    gemm([(a, M, b, N, K)], M, N, 0, 0, out, splits=splits, ws=ws, compact=dict(comp, k_count=zero))
    assert torch.count_nonzero(out).item() == 0


def test_gemm_compaction_rejects_unaligned():
    a = torch.randn(64, 1201, device="cuda")
    w = torch.randn(128, 1201, device="cuda")
    y = torch.zeros(64, 128, device="cuda")
    cnt = torch.tensor([10], dtype=torch.int32, device="cuda")
    assert gemm([(a, 1201, w, 1201, 1201)], 64, 128, 1, 1, y, compact={"m_count": cnt}, check=False) != 0


def test_gemm_m_compaction_nn():
    """dx[c_rows[r]] = dy[a_rows[r]] @ W (W stored [K][N]) for the active rows."""
    torch.manual_seed(8)
    M, N, K = 1344, 1200, 2000
    dy = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda")
    rows, cnt, idx = _active(M, 0.7, 4)
    out = torch.zeros(M, N, device="cuda")
    ws = torch.empty(8 * M * N, device="cuda")
    for splits in (1, 4):
        out.zero_()
        gemm([(dy, K, w, N, K)], M, N, 1, 0, out, splits=splits, ws=ws, compact={"m_count": cnt, "a_rows": rows, "c_rows": rows})
        ref = torch.zeros(M, N, dtype=torch.float64)
        ref[idx] = dy.cpu().double()[idx] @ w.cpu().double()
        scale = (dy.abs().cpu().double()[idx] @ w.abs().cpu().double()).max().item()
        assert maxdiff(out, ref) <= 4e-6 * scale


def test_gemm_compaction_needs_split_mode():
    """Row compaction is implemented by the 3xBF16 kernel only: the fp32-MFMA mode refuses it instead of ignoring it."""
    lib = L.load()
    x = torch.randn(256, 1200, device="cuda")
    w = torch.randn(512, 1200, device="cuda")
    y = torch.zeros(256, 512, device="cuda")
    cnt = torch.tensor([10], dtype=torch.int32, device="cuda")
    lib.ssc_set_gemm_mode(0)
    try:
        assert gemm([(x, 1200, w, 1200, 1200)], 256, 512, 1, 1, y, compact={"m_count": cnt}, check=False) != 0
    finally:
        lib.ssc_set_gemm_mode(1)


@pytest.mark.parametrize("kind,M,N,Ks,splits", [
    ("NT", 1344, 10000, [1200], 1), ("NT", 2304, 768, [2048], 5), ("NT", 600, 520, [1000, 36], 2),
    ("NN", 1344, 1200, [10000], 5), ("NN", 1344, 1000, [4800], 0), ("NN", 516, 644, [1204, 100], 1),
    ("TN", 4800, 1200, [1344], 1), ("TN", 10000, 1200, [1344], 0), ("TN", 768, 2048, [2304], 5), ("TN", 520, 600, [1350, 77], 3),
])
@pytest.mark.parametrize("form", [3, 2])
def test_large_products_split_bf16(kind, M, N, Ks, splits, form):
    """The 128x128 3xBF16 kernel (all three layouts; the m/n-contiguous operands go through the transposing LDS read)
    against float64: fp32-level accuracy, error <= 2e-6 * sum|a||b|."""
    a_kc, b_kc = {"NT": (1, 1), "NN": (1, 0), "TN": (0, 0)}[kind]
    L.load().ssc_debug_set(b"large_form", form)   # 3: 4-wave 128x128 kernel (default), 2: its wave-specialised form
    g = torch.Generator().manual_seed(M + N + len(Ks))
    As = [(torch.randn((M, K) if a_kc else (K, M), generator=g) * torch.exp(torch.randn((M, K) if a_kc else (K, M), generator=g))).cuda() for K in Ks]
    Bs = [(torch.randn((N, K) if b_kc else (K, N), generator=g) * torch.exp(torch.randn((N, K) if b_kc else (K, N), generator=g))).cuda() for K in Ks]
    bias = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    ws = torch.empty(8 * M * N, device="cuda")
    gemm([(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(As, Bs, Ks)], M, N, a_kc, b_kc, out, bias=bias, splits=splits, ws=ws)
    ref = bias.cpu().double()[None, :].repeat(M, 1)
    mag = torch.zeros(M, N, dtype=torch.float64)
    for a, b in zip(As, Bs):
        ad, bd = a.cpu().double(), b.cpu().double()
        ad = ad if a_kc else ad.T
        bd = bd.T if b_kc else bd
        ref += ad @ bd
        mag += ad.abs() @ bd.abs()
    err = ((out.cpu().double() - ref).abs() / mag.clamp_min(1e-30)).max().item()
    assert err <= 2e-6, err
    # exact on small integers (every partial product and sum is representable)
    Ai = [torch.randint(-3, 4, a.shape, generator=g).float().cuda() for a in As]
    Bi = [torch.randint(-3, 4, b.shape, generator=g).float().cuda() for b in Bs]
    gemm([(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(Ai, Bi, Ks)], M, N, a_kc, b_kc, out, splits=splits, ws=ws)
    refi = torch.zeros(M, N, dtype=torch.float64)
    for a, b in zip(Ai, Bi):
        ad, bd = a.cpu().double(), b.cpu().double()
        refi += (ad if a_kc else ad.T) @ (bd.T if b_kc else bd)
    assert torch.equal(out.cpu().double(), refi)
    L.load().ssc_debug_set(b"large_form", 3)


@pytest.mark.parametrize("form", [1, 2, 0])
@pytest.mark.parametrize("kind,N,Ks,splits", [("NT", 4800, [2048, 1200, 1200, 1200], 0), ("NN", 4448, [4800, 4800], 0),
                                              ("NN", 2052, [1000, 36], 3), ("NT", 2300, [1204], 1)])
def test_minibatch_products_wave_specialised(kind, N, Ks, splits, form):
    """M = 64 rows against a wide weight matrix on the 64x256 producer/consumer kernel (x3w_skinny 1: NT and NN, 2: NN only,
    0: off = 64-wide kernels): same fp32-level accuracy in every form."""
    M = 64
    a_kc, b_kc = {"NT": (1, 1), "NN": (1, 0)}[kind]
    lib = L.load()
    lib.ssc_debug_set(b"x3w_skinny", form)
    try:
        g = torch.Generator().manual_seed(N)
        As = [torch.randn(M, K, generator=g).cuda() for K in Ks]
        Bs = [(torch.randn((N, K) if b_kc else (K, N), generator=g) / 30).cuda() for K in Ks]
        out = torch.empty(M, N, device="cuda")
        ws = torch.empty(40 * M * N, device="cuda")
        gemm([(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(As, Bs, Ks)], M, N, a_kc, b_kc, out, splits=splits, ws=ws)
        ref = torch.zeros(M, N, dtype=torch.float64)
        mag = torch.zeros(M, N, dtype=torch.float64)
        for a, b in zip(As, Bs):
            ad, bd = a.cpu().double(), b.cpu().double()
            bd = bd.T if b_kc else bd
            ref += ad @ bd
            mag += ad.abs() @ bd.abs()
        err = ((out.cpu().double() - ref).abs() / mag.clamp_min(1e-30)).max().item()
        assert err <= (2e-6 if (form != 0 or kind == "NT") else 4e-6), err
    finally:
        lib.ssc_debug_set(b"x3w_skinny", 1)


@pytest.mark.parametrize("M,N,Ks,scale_a,scale_b", [(2560, 4800, [1200, 1200], 1.0, 0.03), (5000, 10000, [1200], 0.5, 0.03),
                                                     (2048, 2048, [520, 128], 4.0, 1e-3), (1536, 1536, [1000], 1e-3, 1e-4),
                                                     (1280, 1024, [64], 30.0, 30.0)])
def test_nt_split_fp16_form_is_fp32_accurate(M, N, Ks, scale_a, scale_b):
    """The 2xFP16 form of the wave-specialised 128x128 NT kernel (ssc_model_cfg.gemm_mode 3; op level: ssc_debug_set("gemm_f16")):
    two fp16 pieces per fp32 operand (hi, lo by truncation: 21-22 significant bits), three partial products on
    v_mfma_f32_32x32x16_f16, fp32 accumulate, operands scaled by powers of two into the middle of the fp16 range
    (ssc_pow2_scale; without it an operand of magnitude 1e-3 keeps 2^-25 ABSOLUTE precision in its lo piece: 1e-5 of sum|a||b|).
    Against float64: within 2e-6 of sum|a||b|, the bound the 3xBF16 kernel is held to, over operand magnitudes from 1e-4 to 30;
    exact on small integers."""
    lib = L.load()
    g = torch.Generator().manual_seed(M + N)
    As = [torch.randn(M, K, generator=g) * scale_a for K in Ks]
    Bs = [torch.randn(N, K, generator=g) * scale_b for K in Ks]
    want = ref(As, Bs, 1, 1)
    bound = sum(a.abs().double() @ b.abs().double().t() for a, b in zip(As, Bs))
    dA, dB = [dev(a) for a in As], [dev(b) for b in Bs]
    # the operands' power-of-two scales (largest magnitude -> [2^12, 2^13]), as ssc_decode_prepare measures them
    sc = torch.zeros(3, device="cuda")
    for i, ts in ((0, dA), (1, dB)):
        for j, t in enumerate(ts):
            lib.ssc_pow2_scale(L.ptr(t), t.size(0), t.size(1), t.stride(0), 13, L.ptr(sc[i:i + 1]), 1 if j else 0, L.ptr(sc[2:3]), L.stream_ptr())
    sa, sb = float(sc[0]), float(sc[1])
    for v, ts in ((sa, As), (sb, Bs)):
        top = max(float(t.abs().max()) for t in ts) * v
        assert 4096.0 <= top <= 8192.0 and v == 2.0 ** round(torch.log2(torch.tensor(v)).item())
    outs = {}
    for f16 in (0, 1):
        lib.ssc_debug_set(b"gemm_f16", f16)
        lib.ssc_debug_set(b"large_form", 2)      # always the wave-specialised form
        try:
            out = torch.full((M, N), float("nan"), device="cuda")
            gemm([(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(dA, dB, Ks)], M, N, 1, 1, out, splits=1,
                 compact={"a_scale": sc[0:1], "b_scale": sc[1:2]})
            torch.cuda.synchronize()
            outs[f16] = out.cpu().double()
        finally:
            lib.ssc_debug_set(b"gemm_f16", 0)
            lib.ssc_debug_set(b"large_form", 1)
    e16 = ((outs[1] - want).abs() / bound).max().item()
    e3 = ((outs[0] - want).abs() / bound).max().item()
    assert e3 < 2e-6 and e16 < 2e-6, (e3, e16)
    assert not torch.equal(outs[0], outs[1]) or max(Ks) <= 64      # (the two forms really are different kernels)
    # small integers: every piece and partial sum exact
    Ai = torch.randint(-5, 6, (M, Ks[0]), generator=g).float()
    Bi = torch.randint(-5, 6, (N, Ks[0]), generator=g).float()
    lib.ssc_debug_set(b"gemm_f16", 1)
    lib.ssc_debug_set(b"large_form", 2)
    try:
        out = torch.empty(M, N, device="cuda")
        gemm([(dev(Ai), Ks[0], dev(Bi), Ks[0], Ks[0])], M, N, 1, 1, out, splits=1)
        assert torch.equal(out.cpu(), Ai @ Bi.t())
    finally:
        lib.ssc_debug_set(b"gemm_f16", 0)
        lib.ssc_debug_set(b"large_form", 1)


@pytest.mark.parametrize("M,N,Ks", [(1280, 1024, [1000]), (700, 1300, [1000, 152, 96]), (300, 520, [40])])
def test_presplit_plane_operands_are_bit_identical(M, N, Ks):
    """ssc_gemm_seg.A16 / B16: an operand handed to the 2xFP16 form already split (ssc_split_f16, the producers' own arithmetic) gives
    the bits of the in-kernel split - for A only, B only, both, segment by segment mixed, K that ends inside a 32-k block (the
    planes' padding is zero), with row lists on A, and for the records epilogue; every other form of the product ignores the planes."""
    from gpuutil import split_f16
    lib = L.load()
    g = torch.Generator().manual_seed(7 * M + N)
    dA = [dev(torch.randn(M, K, generator=g) * 0.5) for K in Ks]
    dB = [dev(torch.randn(N, K, generator=g) * 0.03) for K in Ks]
    sc = torch.tensor([64.0, 2.0 ** 15], device="cuda")
    segs = [(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(dA, dB, Ks)]
    pA = [split_f16(a, scale=sc[0:1]) for a in dA]
    pB = [split_f16(b, scale=sc[1:2]) for b in dB]
    # the planes themselves: hi = x truncated to fp16, lo = (x - hi) truncated, zero padding
    a0 = (dA[0] * 64.0).cpu()
    hi = pA[0].cpu().view(torch.float16).view(M, -1, 2, 32)[:, :, 0, :].reshape(M, -1).float()
    lo = pA[0].cpu().view(torch.float16).view(M, -1, 2, 32)[:, :, 1, :].reshape(M, -1).float()
    K0 = Ks[0]
    assert torch.all(hi[:, K0:] == 0) and torch.all(lo[:, K0:] == 0)
    assert torch.all(hi[:, :K0].abs() <= a0.abs()) and torch.all((a0 - hi[:, :K0] - lo[:, :K0]).abs() <= a0.abs() * 2.0 ** -20 + 2.0 ** -24)
    scales = {"a_scale": sc[0:1], "b_scale": sc[1:2]}

    def run(planes, f16=1, extra=None, out=None):
        lib.ssc_debug_set(b"gemm_f16", f16)
        lib.ssc_debug_set(b"large_form", 2)
        try:
            out = torch.full((M, N), float("nan"), device="cuda") if out is None else out
            gemm(segs, M, N, 1, 1, out, splits=1, compact=dict(scales, **(extra or {})), planes=planes)
            torch.cuda.synchronize()
            return out
        finally:
            lib.ssc_debug_set(b"gemm_f16", 0)
            lib.ssc_debug_set(b"large_form", 1)

    want = run(None)
    none = [(None, None)] * len(Ks)
    for name, planes in (("A", [(a, None) for a in pA]), ("B", [(None, b) for b in pB]), ("both", list(zip(pA, pB))),
                         ("mixed", [(pA[i] if i % 2 == 0 else None, pB[i] if i % 2 == 1 else None) for i in range(len(Ks))])):
        assert torch.equal(run(planes), want), name
    # split-K: every workgroup starts in the middle of a segment list (the plane flags follow its cursor)
    ws = torch.empty(4 * M * N + 64, device="cuda")
    outs = []
    for planes in (none, list(zip(pA, pB))):
        lib.ssc_debug_set(b"gemm_f16", 1); lib.ssc_debug_set(b"large_form", 2)
        try:
            o = torch.full((M, N), float("nan"), device="cuda")
            gemm(segs, M, N, 1, 1, o, splits=3, ws=ws, compact=dict(scales), planes=planes)
            torch.cuda.synchronize()
            outs.append(o)
        finally:
            lib.ssc_debug_set(b"gemm_f16", 0); lib.ssc_debug_set(b"large_form", 1)
    assert torch.equal(outs[0], outs[1]) and float((outs[0] - want).abs().max()) < 1e-3
    # the 3xBF16 form of the same product reads the fp32 operands, whatever the planes hold
    assert torch.equal(run(list(zip(pA, pB)), f16=0), run(none, f16=0))
    # row lists on A (planes are addressed by the listed rows too)
    n = M // 3
    rows = torch.randperm(M, generator=g)[:n].sort().values.to(torch.int32).cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    ex = {"m_count": cnt, "a_rows": rows, "c_rows": rows}
    o1 = run(none, extra=ex, out=torch.zeros(M, N, device="cuda"))
    o2 = run(list(zip(pA, pB)), extra=ex, out=torch.zeros(M, N, device="cuda"))
    assert torch.equal(o1, o2) and torch.equal(o1[rows.long()], want[rows.long()])
    # planes of the listed rows only
    pl = [split_f16(a, scale=sc[0:1], rows=(rows, cnt)) for a in dA]
    assert all(torch.equal(x[rows.long()], y[rows.long()]) for x, y in zip(pl, pA))
    # records epilogue
    ntn = (N + 127) // 128
    r1 = torch.zeros(M, ntn, 6, device="cuda"); r2 = torch.zeros(M, ntn, 6, device="cuda")
    run(none, extra={"topk_part": r1}, out=torch.empty(1, N, device="cuda"))
    run(list(zip(pA, pB)), extra={"topk_part": r2}, out=torch.empty(1, N, device="cuda"))
    assert torch.equal(r1, r2)
    # malformed planes are refused
    d_bad = split_f16(dA[0], scale=sc[0:1])[:, :32 * ((Ks[0] + 31) // 32) - 32] if Ks[0] > 32 else None
    if d_bad is not None:
        bad = [(d_bad.contiguous(), None)] + [(None, None)] * (len(Ks) - 1)
        lib.ssc_debug_set(b"gemm_f16", 1)
        try:
            assert gemm(segs, M, N, 1, 1, torch.empty(M, N, device="cuda"), splits=1, planes=bad, check=False) == -1   # SSC_EINVAL
        finally:
            lib.ssc_debug_set(b"gemm_f16", 0)


@pytest.mark.parametrize("rows,K", [(70, 37), (5, 1001), (33, 31)])
def test_split_f16_of_unaligned_operands(rows, K):
    """ssc_split_f16 on an operand without 16-byte rows (odd K = ld): the scalar load path gives the pieces of the aligned path, and
    the columns K .. roundup(K, 32) are zero."""
    from gpuutil import split_f16
    g = torch.Generator().manual_seed(K)
    x = (torch.randn(rows, K, generator=g) * 0.7).cuda()
    sc = torch.tensor([32.0], device="cuda")
    got = split_f16(x, scale=sc)                       # ld = K: rows are not 16-byte aligned
    Ka = (K + 3) // 4 * 4
    xa = torch.zeros(rows, Ka, device="cuda"); xa[:, :K] = x
    want = split_f16(xa, K=K, scale=sc)                # aligned rows, same K
    assert torch.equal(got, want)
    halfs = got.cpu().view(torch.float16).view(rows, -1, 2, 32)
    hi, lo = halfs[:, :, 0, :].reshape(rows, -1).float(), halfs[:, :, 1, :].reshape(rows, -1).float()
    assert torch.all(hi[:, K:] == 0) and torch.all(lo[:, K:] == 0)
    assert float((hi[:, :K] + lo[:, :K] - x.cpu() * 32.0).abs().max()) < 2.0 ** -9

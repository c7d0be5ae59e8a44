"""GPU: exact-fp32 MFMA GEMM (ssc_gemm) against float64 matmul.  Tolerance: 2e-6 * sum|a||b| scale
(fp32 accumulation over K), checked as max-abs <= 1e-4 for K <= 6000 with N(0,1)/sqrt(K)-scaled data."""
import pytest
import torch

from gpuutil import dev, gemm, maxdiff

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

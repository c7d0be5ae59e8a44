#!/usr/bin/env python3
"""Random minibatch shapes of the fused train step at C2's model width: the default arithmetic (3xBF16 on the wave-specialised /
grouped kernels) against the exact-fp32-MFMA mode (ssc_set_gemm_mode(0): other kernels, other launch decisions) - loss, KL and
every gradient must agree to fp32 level for any (B, regions, caption lengths).   python tools/train_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from ssc_runtime import lib as L  # noqa: E402
from ssc_runtime.vocab import Vocabulary  # noqa: E402
from var_updown.models import UpDownCaptioner  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = L.load()
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    bad = 0
    for case in range(n):
        V = pick([10000, 10000, 10001, 9487, 30000])
        Z = pick([128, 128, 150, 64])
        L_ = pick([1, 2, 3, 5, 8])
        sv = pick([1, 1, 0])
        torch.manual_seed(3)
        model = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=2048, embedding_size=1000, hidden_size=1200,
                                attention_projection_size=768, max_caption_length=L_, beam_size=5, z_space=Z, prior_std=1.0,
                                simple_vae=False, latent_embedding="glove", sentiment_vae=sv, senti_prior_multip=0.5,
                                device=torch.device("cuda")).to("cuda")
        model.train()
        B = pick([1, 2, 7, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 150, 192, 200, 256])
        R = pick([5, 36, 36, 64, 100])
        T = L_ + 1
        feats = torch.randn(B, R, 2048, generator=g)
        if R > 4:
            feats[0, R - 2:] = 0
        caps = torch.zeros(B, L_, dtype=torch.long)
        for b in range(B):
            k = ri(1, L_)
            caps[b, :k] = torch.randint(2, V, (k,), generator=g)
        senti = torch.randint(-1, 2, (B, 1), generator=g).float()
        eps = torch.randn(T, B, Z, generator=g)
        res = []
        for mode in (1, 0):
            lib.ssc_set_gemm_mode(mode)
            try:
                eng = model._engine()
                loss, kld = eng.forward(feats.cuda(), caps.cuda(), senti.cuda(), eps.cuda())
                eng.backward(torch.full((B,), 1.0 / B, device="cuda"), torch.full((B,), 1.0 / (B * 750.0), device="cuda"))
                torch.cuda.synchronize()
                res.append((loss.clone(), kld.clone(), {k: v.clone() for k, v in eng.grad_dict().items()}))
            finally:
                lib.ssc_set_gemm_mode(1)
        dl = float((res[0][0] - res[1][0]).abs().max()) / (float(res[1][0].abs().max()) + 1.0)
        dk = float((res[0][1] - res[1][1]).abs().max()) / (float(res[1][1].abs().max()) + 1.0)
        dg, worst_k = 0.0, ""
        for k in res[0][2]:
            e = float((res[0][2][k] - res[1][2][k]).abs().max()) / (float(res[1][2][k].abs().max()) + 1e-3)
            if e > dg:
                dg, worst_k = e, k
        flag = "" if max(dl, dk) < 2e-5 and dg < 2e-3 else "   <-- MISMATCH"
        if flag:
            bad += 1
        print(f"case {case}: B={B} R={R} L={L_} V={V} Z={Z} sv={sv}: loss {dl:.1e} kld {dk:.1e} grads {dg:.1e} ({worst_k}){flag}", flush=True)
    print(f"{bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Random call shapes of the decode step at C4's model width: every large-call path on (per-token table, per-image attended-feature
table, products over distinct parents, un-gathered states where allowed) against the plain path (all switched off, states
re-ordered, no back-pointers) - log-probs, states and attention weights must agree to fp32 level for any (images, groups, beam,
regions).  The kernel-form / split / threshold decisions of the launchers over shapes no fixed test names.
  python tools/decode_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from ssc_runtime import lib as L  # noqa: E402
from ssc_runtime.vocab import Vocabulary  # noqa: E402
from var_updown.models import UpDownCaptioner  # noqa: E402

KEYS = (b"dec_dedup", b"dec_att_table", b"dec_ungathered")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = L.load()
    V, F, E, H, A, Z = 10000, 2048, 1000, 1200, 768, 128
    torch.manual_seed(2)
    m = UpDownCaptioner(Vocabulary.synthetic(V), image_feature_size=F, embedding_size=E, hidden_size=H, attention_projection_size=A,
                        max_caption_length=20, beam_size=5, z_space=Z, prior_std=1.0, simple_vae=False, latent_embedding="glove",
                        sentiment_vae=1, senti_prior_multip=0.5, device=torch.device("cuda")).to("cuda")
    m.eval()
    m._engine()
    dec = m._dec
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    bad, worst = 0, 0.0
    for case in range(n):
        nimg = pick([1, 2, 5, 7, 8, 9, 13, 20, 33, 50, 64])
        beam = pick([1, 2, 3, 5, 10])
        groups = pick([1, 2, 3, 4, 7, 13, 14, 20, 40])
        R = pick([9, 27, 36, 36, 50, 100])
        G = nimg * groups * beam
        if G > 6000 or G * R > 300_000:
            continue
        NG = nimg * groups
        feats = torch.randn(nimg, R, F, generator=g).cuda()
        tok = torch.randint(1, V, (G,), generator=g).cuda()
        sent = torch.randint(-1, 2, (nimg,), generator=g).float().repeat_interleave(groups * beam).cuda()
        eps = torch.randn(G, Z, generator=g).cuda()
        base = {k: (torch.randn(NG, beam, H, generator=g) * 0.3).cuda() for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")}
        parent = torch.randint(0, beam, (NG, beam), generator=g).cuda()
        gathered = {k: v.gather(1, parent.view(NG, beam, 1).expand(NG, beam, H)).reshape(G, H).contiguous() for k, v in base.items()}
        outs = []
        for on in (1, 0):
            for key in KEYS:
                lib.ssc_debug_set(key, on)
            try:
                ctx = dec.prepare(feats)
                ung = bool(on) and dec.ungathered_ok(ctx, G, beam)
                st = {k: v.reshape(G, H) for k, v in base.items()} if ung else dict(gathered)
                if on:
                    st["_parent"] = parent
                if ung:
                    st["_ungathered"] = True
                lp, so, al = dec.step(ctx, tok, st, sent, eps)
                torch.cuda.synchronize()
                outs.append((lp.clone(), {k: so[k].clone() for k in ("h1", "c1", "h_decoder", "c_decoder")}, al.clone(), ung))
            finally:
                for key in KEYS:
                    lib.ssc_debug_set(key, 1)
        d = max(float((outs[0][0] - outs[1][0]).abs().max()), float((outs[0][2] - outs[1][2]).abs().max()),
                max(float((outs[0][1][k] - outs[1][1][k]).abs().max()) for k in outs[0][1]))
        worst = max(worst, d)
        flag = "" if d < 5e-5 else "   <-- MISMATCH"
        if flag:
            bad += 1
        print(f"case {case}: {nimg} images x {groups} groups x beam {beam} = {G} rows, R = {R}, un-gathered {outs[0][3]}: max diff {d:.2e}{flag}", flush=True)
    print(f"worst {worst:.2e}, {bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Train-step time of the same engine at other configurations than the headline C2 (which bench.py measures), one GPU:
    python tools/step_probe.py c5   [steps]    BASELINE.json's stress config: B=128 per GPU, R=100, L=40 (T=41), V=30000; E/H/A/Z as C2
    python tools/step_probe.py yaml [steps]    the reference's shipped configs/config.yaml: E=600 (frozen table, tied output layer),
                                               H=900, A=768, Z=150, BATCH_SIZE=150, R=36, L=20, V=10000 (seeded table instead of GloVe)
    python tools/step_probe.py yaml 10 B=128   any dimension can be overridden
    python tools/step_probe.py <name>-dropin   the reference's training loop on the drop-in module API (autograd + torch optimiser)
    python tools/step_probe.py <name>-decode   diverse decode (50 images x 20 samples per call) of that model at beam 1 and 5
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from ssc_runtime.vocab import Vocabulary
from var_updown.models import UpDownCaptioner

CONFIGS = {"c5": dict(B=128, R=100, F=2048, L=40, Z=128, V=30000, E=1000, H=1200, A=768),
           "yaml": dict(B=150, R=36, F=2048, L=20, Z=150, V=10000, E=600, H=900, A=768),
           "c2": dict(bench.C2),
           "c2v": dict(bench.C2, V=10001)}   # a vocabulary size that is no multiple of 4 (real vocabularies rarely are)


class _SeededTable(UpDownCaptioner):
    def _initialize_glove(self):   # E in {300, 600}: frozen table + tied head (updown_captioner.py:75-119); no download here
        g = torch.Generator().manual_seed(3)
        return torch.randn(self._vocabulary.get_vocab_size(), self.embedding_size, generator=g) * 0.3


def decode_probe(model, c, dev, beam, images=50, n_z=20):
    """diverse decode of `images`-image chunks x n_z latent samples at beam width `beam` (the shipped yaml has BEAM_SIZE 1; C4 uses 5)"""
    from ssc_runtime.inference import count_tokens, diverse_decode
    model.eval()
    g = torch.Generator().manual_seed(4321)
    feats = [torch.randn(images, c["R"], c["F"], generator=g).to(dev) for _ in range(2)]
    senti = torch.ones(images, device=dev)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 2.0:   # (a GPU coming out of idle needs > 1 s to reach its sustained state)
        diverse_decode(model._dec, feats[0], senti, n_z, beam, c["L"], 1, early_stop=False)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    reps = 6
    for i in range(reps):
        pred, _ = diverse_decode(model._dec, feats[i % 2], senti, n_z, beam, c["L"], 1, early_stop=False)
        tokens += count_tokens(pred, 1)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"decode {images} images x {n_z} samples, beam {beam} ({images * n_z * beam} rows): {reps * images / el:.0f} images/s, "
          f"{reps * images * n_z * c['L'] / el / 1e3:.0f} k tokens/s, {el / reps * 1e3:.1f} ms per call", flush=True)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c5"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    decode = name.endswith("-decode")
    dropin = name.endswith("-dropin")
    name = name.replace("-decode", "").replace("-dropin", "")
    c = dict(CONFIGS[name])
    for kv in sys.argv[3:]:            # overrides: B=128 V=9487 ...
        k, v = kv.split("=")
        c[k] = int(v)
    dev = torch.device("cuda")
    torch.manual_seed(2)
    tied = c["E"] in (300, 600)
    model = _SeededTable(Vocabulary.synthetic(c["V"]), image_feature_size=c["F"], embedding_size=c["E"], hidden_size=c["H"],
                         attention_projection_size=c["A"], max_caption_length=c["L"], beam_size=5, use_cbs=tied, z_space=c["Z"],
                         prior_std=1.0, simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5,
                         device=dev).to(dev)
    eng = model._engine()
    if decode:
        shapes = ((1, 50, 20), (5, 50, 20), (10, 25, 20), (3, 50, 20), (5, 1, 20), (5, 8, 20), (5, 200, 1), (1, 1, 1))
        if os.environ.get("SSC_PROBE_SMALL"):   # the small-call end only (the reference's own inference loop decodes one image at a time)
            shapes = ((5, 1, 20), (5, 2, 20), (5, 4, 20), (1, 1, 20), (1, 4, 20))
        for beam, images, n_z in shapes:
            decode_probe(model, c, dev, beam, images, n_z)
        return
    if dropin:   # the reference's own loop (var_updown/scripts/train.py:154-176) on the module API: autograd, clip_grad_norm_, torch SGD
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.015, momentum=0.9, weight_decay=0.001)
        feats, caps, senti, _ = bench.synth_batch(1234, c["B"], c["R"], c["F"], c["L"], c["V"], c["Z"], dev)
        senti = senti.reshape(-1, 1)

        def it():
            out = model(feats, None, None, caps, senti)
            loss = out["loss"].mean() + out["kld"].mean() / 750.0
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 12.5)
            opt.step()
        model.train()
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            it()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{name} drop-in loop (model(...), loss.backward(), clip_grad_norm_, torch.optim.SGD): {ms:.2f} ms -> {c['B'] / ms * 1e3:.0f} captions/s", flush=True)
        return
    batches = [bench.synth_batch(1234 + i, c["B"], c["R"], c["F"], c["L"], c["V"], c["Z"], dev) for i in range(2)]

    def step(i):
        feats, caps, senti, eps = batches[i % 2]
        eng.train_step(feats, caps, senti, eps, lr=0.015, kld_weight=750.0, momentum=0.9, weight_decay=0.001, max_norm=12.5)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name} train step {c}: {ms:.2f} ms -> {c['B'] / ms * 1e3:.0f} captions/s", flush=True)


if __name__ == "__main__":
    main()

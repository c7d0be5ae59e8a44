#!/usr/bin/env python3
"""Train-step time at BASELINE.json's stress config C5 (B=128 per GPU, R=100, L=40 (T=41), V=30000; E/H/A/Z as C2) on one GPU:
python tools/c5_step.py [steps]   (bench.py measures the headline config C2; this is the same engine on C5's shapes)."""
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


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    c = dict(B=128, R=100, F=2048, L=40, Z=128, V=30000, E=1000, H=1200, A=768)
    dev = torch.device("cuda")
    torch.manual_seed(2)
    model = UpDownCaptioner(Vocabulary.synthetic(c["V"]), image_feature_size=c["F"], embedding_size=c["E"], hidden_size=c["H"],
                            attention_projection_size=c["A"], max_caption_length=c["L"], beam_size=5, z_space=c["Z"], prior_std=1.0,
                            simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5, device=dev).to(dev)
    eng = model._engine()
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
    print(f"C5 train step (B=128, R=100, T=41, V=30000): {ms:.2f} ms -> {c['B'] / ms * 1e3:.0f} captions/s")


if __name__ == "__main__":
    main()

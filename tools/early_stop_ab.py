"""A/B of the decode leg with and without early stop, alternating (order effects: the first leg of a process measures low)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from ssc_runtime.vocab import Vocabulary
from ssc_runtime.inference import diverse_decode, count_tokens
from var_updown.models import UpDownCaptioner
c = dict(bench.C2)
dev = torch.device("cuda", 0)
torch.manual_seed(2)
model = UpDownCaptioner(Vocabulary.synthetic(c["V"]), image_feature_size=c["F"], embedding_size=c["E"], hidden_size=c["H"],
                        attention_projection_size=c["A"], max_caption_length=c["L"], beam_size=5, z_space=c["Z"], prior_std=1.0,
                        simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5, device=dev).to(dev)
model.eval(); model._engine(); dec = model._dec; dec.weights_frozen = True
g = torch.Generator().manual_seed(4321)
feats = [torch.randn(100, c["R"], c["F"], generator=g).to(dev) for _ in range(4)]
senti = torch.ones(100, device=dev)
def leg(early, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tok = 0
    for i in range(n):
        pred, _ = diverse_decode(dec, feats[i % 4], senti, 20, 5, c["L"], 1, early_stop=early)
        tok += count_tokens(pred, 1)
    torch.cuda.synchronize(); return tok / (time.perf_counter() - t0)
t = time.perf_counter()
while time.perf_counter() - t < 3: leg(True, 2)
for rep in range(3):
    for early in (True, False):
        print(f"rep {rep} early_stop={early}: {leg(early)/1e3:.1f} k tokens/s", flush=True)

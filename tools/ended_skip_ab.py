"""C4 shapes with an output bias that makes captions end after ~8-12 tokens (what a trained model does; the random-init weights of
the bench never emit END): diverse decode with and without leaving ended beams out of the steps, alternating."""
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
with torch.no_grad():
    model._output_layer.bias[1] += float(sys.argv[1]) if len(sys.argv) > 1 else 7.0
model.eval(); model._engine(); dec = model._dec; dec.weights_frozen = True
g = torch.Generator().manual_seed(4321)
feats = [torch.randn(100, c["R"], c["F"], generator=g).to(dev) for _ in range(4)]
senti = torch.ones(100, device=dev)
def leg(skip, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tok = 0; steps = 0
    for i in range(n):
        pred, k = diverse_decode(dec, feats[i % 4], senti, 20, 5, c["L"], 1, early_stop=True, skip_dead=skip)
        tok += count_tokens(pred, 1); steps += k
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return tok / el, el / n * 1e3, tok / (n * 2000), steps / n
t = time.perf_counter()
while time.perf_counter() - t < 3: leg(True, 2)
for rep in range(2):
    for skip in (False, True):
        r = leg(skip)
        print(f"rep {rep} skip_ended={skip}: {r[0]/1e3:.1f} k tokens/s, {r[1]:.1f} ms per 100-image call, {r[2]:.1f} tokens per caption, {r[3]:.1f} steps per call", flush=True)

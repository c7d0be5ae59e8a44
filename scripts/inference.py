#!/usr/bin/env python3
"""Diverse captioning inference on MI355X - counterpart of the reference's var_updown/scripts/inference.py:19-191: same
flags and JSON output ([{"image_id", "caption"}...], N_Z_SAMPLES captions per image), but images x latent samples are
decoded as ONE batched beam search per chunk instead of a per-image Python loop of N_Z_SAMPLES model calls."""
import argparse
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssc_runtime.config import Config  # noqa: E402
from ssc_runtime.data import SyntheticCaptionData, TensorFileData  # noqa: E402
from ssc_runtime.inference import diverse_decode  # noqa: E402
from ssc_runtime.vocab import Vocabulary  # noqa: E402
from var_updown.models import UpDownCaptioner  # noqa: E402

parser = argparse.ArgumentParser("Run diverse-decoding inference with a trained Style-SeqCVAE captioner (MI355X).")
parser.add_argument("--config", required=True)
parser.add_argument("--config-override", default=[], nargs="*")
parser.add_argument("--gpu-ids", required=True, nargs="+", type=int)
parser.add_argument("--cpu-workers", type=int, default=0)
parser.add_argument("--in-memory", action="store_true")
parser.add_argument("--checkpoint-path", default="", help="checkpoint with a 'model' state_dict; empty: random init")
parser.add_argument("--output-path", default="predictions.json")
parser.add_argument("--evalai-submit", action="store_true", help="accepted for flag parity; EvalAI is a network service")
parser.add_argument("--infer-tensors", default="")
parser.add_argument("--synthetic", type=int, default=0)
parser.add_argument("--vocab-size", type=int, default=10000)
parser.add_argument("--num-boxes", type=int, default=36)
parser.add_argument("--images-per-call", type=int, default=100)
parser.add_argument("--sentiment", type=float, default=None, help="override the per-image sentiment (-1, 0, 1)")
parser.add_argument("--constraints-json", default="",
                    help='constrained beam search: {"<image_id>": ["dog", "fire hydrant", ...]} - up to DATA.CBS.MAX_GIVEN_CONSTRAINTS '
                         "constraint classes per image (what the reference's EvaluationDatasetWithConstraints derives from detector "
                         "boxes, updown-baseline/updown/data/datasets.py:470-620); needs --wordforms-tsv")
parser.add_argument("--wordforms-tsv", default="", help="class name <TAB> comma separated word forms (data/constraint_wordforms*.tsv)")
parser.add_argument("--boxes-json", default="",
                    help='instead of --constraints-json: raw detections {"<image_id>": {"boxes": [[x1, y1, x2, y2], ...], "class_names": [...], '
                         '"scores": [...]}}, filtered to constraints by ssc_runtime.constraints.ConstraintFilter '
                         "(updown-baseline/updown/utils/constraints.py:56-209); needs --hierarchy-json and --wordforms-tsv")
parser.add_argument("--hierarchy-json", default="", help="Open Images class hierarchy (bbox_labels_600_hierarchy_readable.json)")


class _LocalGlove(UpDownCaptioner):
    """Frozen embedding table comes from the checkpoint; no GloVe download at construction time."""

    def _initialize_glove(self):
        return torch.zeros(self._vocabulary.get_vocab_size(), self.embedding_size)


def main():
    _A = parser.parse_args()
    _C = Config(_A.config, _A.config_override)
    random.seed(_C.RANDOM_SEED)
    np.random.seed(_C.RANDOM_SEED)
    torch.manual_seed(_C.RANDOM_SEED)
    device = torch.device("cuda", _A.gpu_ids[0])
    torch.cuda.set_device(device)
    if _A.synthetic:
        vocabulary = Vocabulary.synthetic(_A.vocab_size)
        data = SyntheticCaptionData(_A.synthetic, _A.num_boxes, _C.MODEL.IMAGE_FEATURE_SIZE, _C.DATA.MAX_CAPTION_LENGTH,
                                    _A.vocab_size, seed=4321,
                                    obj_dim=_C.MODEL.Z_SPACE if (_C.MODEL.SENTIMENT_VAE == 2 and not _C.MODEL.SIMPLE_VAE) else 0)
    else:
        vocabulary = Vocabulary.from_files(_C.DATA.VOCABULARY)
        if not _A.infer_tensors:
            raise SystemExit("pass --infer-tensors file.pt or --synthetic N (h5 readers are out of scope)")
        data = TensorFileData(_A.infer_tensors)
    cls = _LocalGlove if (_C.MODEL.EMBEDDING_SIZE in (300, 600) and _A.checkpoint_path) else UpDownCaptioner
    extra = {"mean_choice": {}} if _C.MODEL.SENTIMENT_VAE == 2 else {}   # (per-region attribute MEANS come with the data: data.obj)
    model = cls.from_config(_C, vocabulary=vocabulary, device=device, **extra).to(device)
    if _A.checkpoint_path:
        model.load_state_dict(torch.load(_A.checkpoint_path, map_location=device, weights_only=True)["model"])
    model.eval()
    model._engine()
    model._dec.weights_frozen = True   # the checkpoint's weights stay as they are for the whole run: weight-only tables are formed once
    n_z = max(1, _C.MODEL.N_Z_SAMPLES)
    beam = _C.MODEL.BEAM_SIZE
    boundary = vocabulary.get_token_index("@@BOUNDARY@@")
    predictions = []
    id2word = np.array([vocabulary.get_token_from_index(i) for i in range(vocabulary.get_vocab_size())], dtype=object)
    constraints, builder = {}, None
    per_call = _A.images_per_call
    if _A.constraints_json or _A.boxes_json:
        from ssc_runtime.constraints import ConstraintFilter, FiniteStateMachineBuilder
        if not _A.wordforms_tsv:
            raise SystemExit("--constraints-json / --boxes-json need --wordforms-tsv")
        kmax = max(1, _C.DATA.CBS.MAX_GIVEN_CONSTRAINTS)
        if _A.boxes_json:
            if not _A.hierarchy_json:
                raise SystemExit("--boxes-json needs --hierarchy-json")
            cfilter = ConstraintFilter(_A.hierarchy_json, _C.DATA.CBS.NMS_THRESHOLD, kmax)
            constraints = {int(k): sorted(cfilter(np.asarray(v["boxes"], dtype=np.float32).reshape(-1, 4), v["class_names"],
                                                  np.asarray(v["scores"], dtype=np.float32)))
                           for k, v in json.load(open(_A.boxes_json)).items()}
        else:
            constraints = {int(k): v for k, v in json.load(open(_A.constraints_json)).items()}
        builder = FiniteStateMachineBuilder(vocabulary, _A.wordforms_tsv, None, max_given_constraints=kmax,
                                            max_words_per_constraint=_C.DATA.CBS.MAX_WORDS_PER_CONSTRAINT)
        # (one machine per IMAGE, shared by its N_Z samples and compiled on the device - ssc_fsm_compile -: a call is sized by its rows)
    ROW_BUDGET = 40000   # rows (image, sample, state, beam) per decode step of a constrained call
    with torch.no_grad():
        lo = 0
        while lo < len(data):
            built = None
            n_here = min(per_call, len(data) - lo)
            if builder is not None:
                # one machine per image, padded to the chunk's largest state count: the states an image does not use have no
                # incoming transition and never hold a finite beam (their rows are skipped: diverse_decode(skip_dead=True));
                # the chunk ends where its rows per step would exceed the budget
                built, S = [], 0
                for i in range(n_here):
                    m = builder.build([c for c in constraints.get(int(data.image_id[lo + i]), [])][:kmax])
                    if built and (len(built) + 1) * n_z * max(S, m[1]) * beam > ROW_BUDGET:
                        break
                    built.append(m)
                    S = max(S, m[1])
                n_here = len(built)
            feats = data.feats[lo: lo + n_here].to(device)
            senti = data.senti[lo: lo + n_here, 0].to(device)
            if _A.sentiment is not None:
                senti = torch.full_like(senti, _A.sentiment)
            fsm = ncons = None
            if built is not None:
                V = vocabulary.get_vocab_size()
                fsm = torch.zeros(n_here, S, S, V, dtype=torch.uint8)
                for i, (m, ns, _) in enumerate(built):
                    fsm[i, :ns, :ns] = m[:ns, :ns]
                fsm = fsm.to(device)
                ncons = torch.tensor([len(constraints.get(int(data.image_id[lo + i]), [])[:kmax]) for i in range(n_here)]
                                     ).repeat_interleave(n_z)
            obj = data.obj[lo: lo + n_here, : feats.size(1)].to(device) if getattr(data, "obj", None) is not None else None
            pred, _ = diverse_decode(model._dec, feats, senti, n_z, beam, _C.DATA.MAX_CAPTION_LENGTH, boundary, fsm=fsm,
                                     num_constraints=ncons, min_constraints_to_satisfy=_C.MODEL.MIN_CONSTRAINTS_TO_SATISFY,
                                     obj_means=obj)
            # ids -> words, cut at the first @@BOUNDARY@@ (inference.py:180-182): one table lookup for the whole chunk - the
            # per-token Python calls this replaces took as long as the chunk's 20 decode steps on the GPU
            ids = pred.cpu().numpy()                                   # (images, n_z, steps)
            words = id2word[ids]
            is_end = ids == boundary
            n_keep = np.where(is_end.any(-1), is_end.argmax(-1), ids.shape[-1])
            for i in range(ids.shape[0]):
                image_id = int(data.image_id[lo + i])
                for k in range(n_z):
                    predictions.append({"image_id": image_id, "caption": " ".join(words[i, k, : n_keep[i, k]])})
            lo += n_here
    json.dump(predictions, open(_A.output_path, "w", encoding="utf-8"))
    print(f"wrote {len(predictions)} captions to {_A.output_path}")


if __name__ == "__main__":
    main()

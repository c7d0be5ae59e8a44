"""UpDownCaptioner (Style-SeqCVAE): MI355X-native drop-in for
var_updown/var_updown/models/updown_captioner.py:20-466 of the reference - same constructor / from_config /
forward / _decode_step signatures, attribute names (``_updown_cell._language_lstm_cell_decoder`` is part of the
API: var_updown/scripts/train.py:157,160) and state_dict keys.

Underneath, parameters live in one flat HBM buffer (views with 16-B aligned rows), the teacher-forced training
forward and its BPTT are one fused C-ABI call each (ssc_train_fwd / ssc_train_bwd, exposed to autograd through a
single torch.autograd.Function), and eval decoding runs ssc_decode_step + on-device constrained beam search.
There is no CPU fallback: calling the model on a CPU tensor raises.
"""
import functools
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from ssc_runtime import lib as _lib
from ssc_runtime.cellops import cell_train_step
from ssc_runtime.decode import DecodeEngine
from ssc_runtime.decoding import select_best_beam_with_constraints
from ssc_runtime.engine import FIELD_OF, ModelDims, TrainEngine

from ..modules import ConstrainedBeamSearch, UpDownCell


class _SeqCVAETrainFn(torch.autograd.Function):
    """loss_b, kld_b = f(params; feats, caps, sentiment, eps) with the hand-derived BPTT as backward."""

    @staticmethod
    def forward(ctx, eng, names, feats, caps, sentiment, eps, obj_means, *params):
        loss, kld = eng.forward(feats, caps, sentiment, eps, obj_means)
        ctx.eng, ctx.names = eng, names
        ctx.version = eng.fwd_version
        return loss, kld

    @staticmethod
    def backward(ctx, gl, gk):
        eng = ctx.eng
        if eng.fwd_version != ctx.version:
            raise RuntimeError("UpDownCaptioner: backward() after a newer forward(); the activation workspace holds "
                               "only the latest forward")
        need = ctx.needs_input_grad[7:]
        skip = [n for n, k in zip(ctx.names, need) if not k]
        eng.backward(gl, gk, skip=skip)
        if eng.dp_autograd:   # data parallel on the autograd path: ONE sum all-reduce of the flat gradient buffer, then the mean
            world = eng.allreduce_grads()
            if world > 1:
                eng.grads.flat.mul_(1.0 / world)
        frozen = set(eng.frozen_names)
        grads = tuple(eng.grads.views[n].clone() if (k and n not in frozen) else None for n, k in zip(ctx.names, need))
        return (None,) * 7 + grads


class UpDownCaptioner(nn.Module):
    def __init__(self, vocabulary, image_feature_size, embedding_size, hidden_size, attention_projection_size,
                 max_caption_length=20, beam_size=1, use_cbs=False, min_constraints_to_satisfy=2, z_space=150,
                 prior_std=None, simple_vae=False, latent_embedding=None, latent_embedding_multip=1,
                 sentiment_vae=False, senti_prior_multip=1, cbs_simple=False, device=None, mean_choice=None):
        """Same parameters as the reference (updown_captioner.py:21-41) plus `mean_choice` (SENTIMENT_VAE = 2 only): the attribute
        word -> z_space-vector table the reference builds from files at hard-coded paths (`/path/to/sentiglove10.pkl`,
        `/path/to/wordform_swd_scores.json`, updown_captioner.py:79-93) - and cannot finish building as shipped (`self.senti_glove_5`
        is never defined, :89).  Here the caller supplies it: a dict of vectors, or use mean_choice_from_sentiglove /
        mean_choice_from_senti_wordnet, which restate :80-86."""
        super().__init__()
        self._vocabulary = vocabulary
        self.image_feature_size = image_feature_size
        self.embedding_size = embedding_size
        self.hidden_size = hidden_size
        self.attention_projection_size = attention_projection_size
        self._max_caption_length = max_caption_length
        self._use_cbs = use_cbs
        self._min_constraints_to_satisfy = min_constraints_to_satisfy
        self.z_space = z_space
        _vocab_size = vocabulary.get_vocab_size()
        self._pad_index = vocabulary.get_token_index("@@UNKNOWN@@")
        self._boundary_index = vocabulary.get_token_index("@@BOUNDARY@@")
        self.prior_std = 1.0 if prior_std is None else prior_std
        self.sentiment_vae = int(sentiment_vae)
        self.senti_prior_multip = senti_prior_multip
        self.simple_vae = simple_vae
        self.latent_embedding = latent_embedding
        self.latent_embedding_multip = latent_embedding_multip
        self.mean_choice = None
        if self.sentiment_vae == 2:
            # attention-grounded style prior (updown_captioner.py:76-93, updown_cell.py:160-163,185-188,219-222)
            if latent_embedding not in ("glove", "senti_word_net"):
                raise NotImplementedError()          # (updown_captioner.py:92-93)
            if mean_choice is None:
                mean_choice = self._load_mean_choice()
            self.mean_choice = {k: np.asarray(v, dtype=np.float64).reshape(-1) for k, v in mean_choice.items()}
            bad = [k for k, v in self.mean_choice.items() if v.shape[0] != z_space]
            if bad:
                raise ValueError(f"mean_choice vectors must have z_space = {z_space} entries (e.g. {bad[0]!r} has {self.mean_choice[bad[0]].shape[0]})")
            if not simple_vae and latent_embedding == "glove" and z_space != 150:
                raise ValueError("SENTIMENT_VAE = 2 with LATENT_EMBEDDING 'glove' conditions the language LSTMs on 150 columns "
                                 "(updown_cell.py:63-70): Z_SPACE must be 150")
        self._tied = self.embedding_size in (300, 600)
        if self._tied:  # frozen GloVe(+deps) table, output layer tied to it (updown_captioner.py:75-100,112-119)
            glove_vectors = self._initialize_glove()
            self._embedding_layer = nn.Embedding.from_pretrained(glove_vectors, freeze=True, padding_idx=self._pad_index)
        else:
            self._embedding_layer = nn.Embedding(_vocab_size, embedding_size, padding_idx=self._pad_index)
            assert not use_cbs, "CBS is not supported without Frozen GloVe embeddings (300d)"
        self._updown_cell = UpDownCell(image_feature_size, embedding_size, hidden_size, attention_projection_size, z_space,
                                       self.sentiment_vae, simple_vae, device, latent_embedding)
        self._updown_cell._host = _HostRef(self)
        if self._tied:
            self._output_projection = nn.Sequential(nn.Linear(hidden_size, self.embedding_size), nn.Tanh())
            self._output_layer = nn.Linear(self.embedding_size, _vocab_size, bias=False)
            self._output_layer.weight = self._embedding_layer.weight
        else:
            self._output_projection = nn.Identity()
            self._output_layer = nn.Linear(hidden_size, _vocab_size)
        self._log_softmax = nn.LogSoftmax(dim=1)
        # The reference's non-CBS branch (allennlp BeamSearch) cannot unpack the 5-tuple step;
        # here non-CBS decoding is CBS over the trivial one-state FSM.
        self._beam_search = ConstrainedBeamSearch(self._boundary_index, max_steps=max_caption_length,
                                                  beam_size=beam_size, per_node_beam_size=beam_size // 2)
        self.device = device
        self.cbs_simple = cbs_simple
        # noise source: "cpu" reproduces the reference's stream (one CPU torch.randn((rows, Z)) per step,
        # updown_cell.py:206) without its per-step host sync; "device" draws on the GPU (not bit-comparable).
        self.eps_source = "cpu"
        self._eps_override = None      # test hook: explicit (T,B,Z) / list of (G,Z)
        self._eng: Optional[TrainEngine] = None
        self._dec: Optional[DecodeEngine] = None
        self._ctx_cache = None

    # ------------------------------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config, **kwargs):
        """Instantiate from a Config (updown_captioner.py:141-166); extra kwargs such as cbs_simple are ignored as in
        the reference.  mean_choice=... (SENTIMENT_VAE = 2) is handed to the constructor."""
        _C = config
        return cls(vocabulary=kwargs.pop("vocabulary"), image_feature_size=_C.MODEL.IMAGE_FEATURE_SIZE,
                   embedding_size=_C.MODEL.EMBEDDING_SIZE, hidden_size=_C.MODEL.HIDDEN_SIZE,
                   attention_projection_size=_C.MODEL.ATTENTION_PROJECTION_SIZE, beam_size=_C.MODEL.BEAM_SIZE,
                   max_caption_length=_C.DATA.MAX_CAPTION_LENGTH, use_cbs=_C.MODEL.USE_CBS,
                   min_constraints_to_satisfy=_C.MODEL.MIN_CONSTRAINTS_TO_SATISFY, z_space=_C.MODEL.Z_SPACE,
                   prior_std=_C.MODEL.PRIOR_STD, simple_vae=_C.MODEL.SIMPLE_VAE, latent_embedding=_C.MODEL.LATENT_EMBEDDING,
                   sentiment_vae=_C.MODEL.SENTIMENT_VAE, senti_prior_multip=_C.MODEL.SENTI_PRIOR_MULTIP,
                   latent_embedding_multip=_C.MODEL.LATENT_EMBEDDING_MULTIP, cbs_simple=_C.MODEL.CBS_SIMPLE,
                   device=kwargs["device"], mean_choice=kwargs.get("mean_choice"))

    def _initialize_glove(self):
        """GloVe 42B (+ dependency embeddings for 600-d) rows for the vocabulary (updown_captioner.py:168-226).
        Needs torchtext and its vector cache, like the reference; override in a subclass to supply a table."""
        try:
            from torchtext.vocab import GloVe, Vectors
        except ImportError as e:  # no silent substitute for pretrained vectors
            raise ImportError("EMBEDDING_SIZE in {300,600} initialises frozen GloVe embeddings through torchtext "
                              "(not installed); subclass and override _initialize_glove() to provide the table") from e
        V = self._vocabulary.get_vocab_size()
        glove = GloVe(name="42B", dim=300, cache="/path/to/.vector_cache")
        deps = Vectors(name="deps.words", cache="/path/to/.vector_cache") if self.embedding_size == 600 else None
        table = torch.zeros(V, self.embedding_size)
        for word, i in self._vocabulary.get_token_to_index_vocabulary().items():
            parts = []
            for src in ([glove] if deps is None else [glove, deps]):
                parts.append(src.vectors[src.stoi[word]] if word in src.stoi else 2 * torch.randn(300) - 1)
            table[i] = torch.cat(parts, 0)
        return table

    # ---- SENTIMENT_VAE = 2: attribute table and per-object means --------------------------------------------------
    def _load_mean_choice(self):
        """Hook for subclasses: the attribute table when none was passed to the constructor (the reference reads pickles at
        hard-coded paths here, updown_captioner.py:79-86)."""
        raise ValueError("SENTIMENT_VAE = 2 needs the attribute table: pass mean_choice={word: vector of z_space floats} "
                         "(see mean_choice_from_sentiglove / mean_choice_from_senti_wordnet) or override _load_mean_choice()")

    @staticmethod
    def mean_choice_from_sentiglove(senti_glove_10, z_space):
        """updown_captioner.py:79-81: each 10-d sentiment-GloVe vector repeated z_space / 10 times per entry (np.repeat)."""
        return {k: np.repeat(np.asarray(v), int(z_space / 10)) for k, v in senti_glove_10.items()}

    @staticmethod
    def mean_choice_from_senti_wordnet(scores, z_space):
        """updown_captioner.py:83-86: (positive - negative) SentiWordNet score of a word form, z_space times
        (data/wordform_swd_scores.json: word -> [pos, obj, neg])."""
        return {k: np.repeat(v[0] - v[2], z_space) for k, v in scores.items()}

    def translate_obj_atts2obj_means(self, obj_atts):
        """Per image a list of objects `(name, [attribute strings])` -> (B, max objects, z_space) float32: the mean of the table
        vectors of an object's attributes (first word of each attribute string; unknown words are skipped; no known attribute ->
        zeros), zero-padded over objects, times LATENT_EMBEDDING_MULTIP (updown_captioner.py:509-532)."""
        Z = self.z_space
        per_image = []
        for im in obj_atts:
            means = np.zeros((len(im), Z))
            for i_o, obj in enumerate(im):
                vecs = []
                for att in obj[1]:
                    try:                                   # (the reference skips whatever it cannot look up, :517-520)
                        vecs.append(self.mean_choice[att.split(" ")[0]])
                    except Exception:                      # noqa: BLE001
                        pass
                if vecs:
                    means[i_o] = np.mean(vecs, axis=0)
            per_image.append(means)
        out = np.zeros((len(per_image), max(len(x) for x in per_image), Z))
        for i_i, m in enumerate(per_image):
            out[i_i, : len(m)] = m
        dev = self._embedding_layer.weight.device
        return torch.Tensor(out * self.latent_embedding_multip).to(dev)

    def _obj_means(self, obj_atts, batch_size, num_boxes):
        """The per-region attribute means the cell pools (updown_cell.py:160-163) from what forward() was given: the reference's
        nested lists (translated as above), or - an extension - a (B, R, z_space) tensor of means as is."""
        if self.sentiment_vae != 2 or self.simple_vae:
            return None
        if obj_atts is None:
            raise ValueError("SENTIMENT_VAE = 2: forward() needs obj_atts (the reference multiplies the attention weights with it, "
                             "updown_cell.py:160-163)")
        means = obj_atts if torch.is_tensor(obj_atts) else self.translate_obj_atts2obj_means(obj_atts)
        if tuple(means.shape) != (batch_size, num_boxes, self.z_space):
            raise ValueError(f"obj_atts means {tuple(means.shape)} do not line up with the image features: expected "
                             f"({batch_size}, {num_boxes}, {self.z_space}) - one entry per region, as the attention weights have")
        return means.to(self._embedding_layer.weight.device, torch.float32).contiguous()

    # ---- engine plumbing -------------------------------------------------------------------------------------
    def _dims(self) -> ModelDims:
        sv1 = self.sentiment_vae == 1 and not self.simple_vae
        sv2 = self.sentiment_vae == 2 and not self.simple_vae
        return ModelDims(V=self._vocabulary.get_vocab_size(), E=self.embedding_size, H=self.hidden_size,
                         A=self.attention_projection_size, F=self.image_feature_size, Z=self.z_space,
                         S=self._updown_cell.senti_cols, tied=self._tied, kld_mode=0 if self.sentiment_vae == 0 else (2 if sv2 else 1),
                         pm_scale=float(self.senti_prior_multip) if sv1 else 0.0, prior_var=float(self.prior_std) ** 2,
                         pad=self._pad_index, boundary=self._boundary_index)

    def _named(self) -> Dict[str, nn.Parameter]:
        out = {}
        for n, p in self.named_parameters():  # named_parameters() de-duplicates the tied output weight
            if n in FIELD_OF:
                out[n] = p
        return out

    def _engine(self) -> TrainEngine:
        dev = self._embedding_layer.weight.device
        if dev.type != "cuda":
            raise RuntimeError("UpDownCaptioner (MI355X build) runs on a ROCm GPU only: move the model with "
                               ".to('cuda'); there is no CPU fallback")
        if self._eng is None or self._eng.device != dev:
            self._eng = TrainEngine(self._dims(), dev)
            self._dec = DecodeEngine(self._eng.dims, self._eng.params.c_struct, dev)
        self._eng.adopt(self._named())
        return self._eng

    def _draw_eps(self, steps: int, rows: int, device) -> torch.Tensor:
        if self._eps_override is not None:
            e = self._eps_override
            assert tuple(e.shape) == (steps, rows, self.z_space), (e.shape, (steps, rows, self.z_space))
            return e.to(device, torch.float32).contiguous()
        if self.eps_source == "device":
            return torch.randn(steps, rows, self.z_space, device=device)
        host = torch.empty(steps, rows, self.z_space, pin_memory=True)
        for t in range(steps):  # one (rows, Z) draw per step: the reference's CPU stream (updown_cell.py:206)
            host[t] = torch.randn(rows, self.z_space)
        return host.to(device, non_blocking=True)

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self, image_features: torch.Tensor, obj_atts=None, image_attributes=None, caption_tokens=None,
                sentiment=None, fsm: torch.Tensor = None, num_constraints: torch.Tensor = None, constraints=None,
                constraint2states=None):
        batch_size, num_boxes, _ = image_features.size()
        eng = self._engine()
        dev = eng.device
        if self.training and caption_tokens is not None:
            # training branch (updown_captioner.py:263-323): boundary tokens, T-step loop, KLD, masked NLL - all fused
            L = caption_tokens.size(1)
            eps = self._draw_eps(L + 1, batch_size, dev)
            names = list(self._named().keys())
            params = [self._named()[n] for n in names]
            sent = sentiment if sentiment is not None else None
            obj_means = self._obj_means(obj_atts, batch_size, num_boxes)
            loss, kld = _SeqCVAETrainFn.apply(eng, names, image_features.contiguous().float(),
                                              caption_tokens.contiguous().long(), sent, eps, obj_means, *params)
            return {"loss": loss, "kld": kld}
        # eval branch (updown_captioner.py:324-366)
        start_predictions = torch.full((batch_size,), self._boundary_index, dtype=torch.long, device=dev)
        obj_means = self._obj_means(obj_atts, batch_size, num_boxes)   # (updown_captioner.py:246-247)
        step = functools.partial(self._decode_step, image_features, obj_means, sentiment=sentiment)
        with torch.no_grad():
            if self._use_cbs and fsm is not None:
                fsm_d = fsm.to(dev).to(torch.uint8)
                beams, lps = self._beam_search.search(start_predictions, None, step, fsm_d)
                best, _valid = select_best_beam_with_constraints(beams, lps, num_constraints, constraints, constraint2states,
                                                                 self._min_constraints_to_satisfy, self.cbs_simple)
            else:
                V = self._vocabulary.get_vocab_size()
                fsm_d = torch.ones(batch_size, 1, 1, V, dtype=torch.uint8, device=dev)
                beams, lps = self._beam_search.search(start_predictions, None, step, fsm_d)
                best = beams[:, 0, 0, :]
        return {"predictions": best}

    def _image_context(self, image_features, obj_means=None):
        """Per-image terms (mask, averaged features, projected features, hoisted gate term) for the eval decode step, computed
        once per image set and parameter version.  The cache entry keeps a REFERENCE to the caller's tensor and is hit only
        for that very tensor object at the same in-place version: a freed tensor's address can be handed to the next batch
        by the caching allocator (same pointer, same shape, version 0), so a key made of data_ptr/shape would alias."""
        c = self._ctx_cache
        if (c is None or c[0] is not image_features or c[1] != image_features._version or
                c[3] != self._eng.param_version() or c[4] is not obj_means):
            ctx = self._dec.prepare(image_features.float(), obj_means)
            self._ctx_cache = (image_features, image_features._version, ctx, self._eng.param_version(), obj_means)
        return self._ctx_cache[2]

    def _rows(self, t, B, G):
        """(B, k) per-image tensor -> (G, k) per-row, batch-major."""
        if t is None:
            return None
        t = t.reshape(B, -1)
        return t.unsqueeze(1).expand(B, G // B, t.size(1)).reshape(G, t.size(1))

    def _decode_step(self, image_features, obj_atts, previous_predictions, states=None, sentiment=None, attrib_cond=None,
                     prior_mean=None, prior_var=None):
        """One decoding step (updown_captioner.py:371-455).  Eval mode: rows = B * net_beam_size, returns the 5-tuple
        (log_probs, states, prior_mean, prior_log_var, attention_weights).  prior_mean / prior_var are derived from
        `sentiment` (SENTI_PRIOR_MULTIP, PRIOR_STD) exactly as forward() does (:249-261)."""
        if self.training:
            # stand-alone training-mode step (no autograd): the 7-tuple of updown_captioner.py:452-453.  The
            # differentiable path is forward(), which fuses all T steps.
            eng = self._engine()
            G = previous_predictions.size(0)
            emb = self._embedding_layer.weight.detach()[previous_predictions.to(eng.device)]
            eps = self._eps_override.pop(0) if self._eps_override is not None else (
                torch.randn(G, self.z_space, device=eng.device) if self.eps_source == "device" else torch.randn(G, self.z_space))
            h_dec, states, mean, log_var, alpha, _ = cell_train_step(eng.dims, eng.params.views, image_features.to(eng.device),
                                                                     emb, states, sentiment, eps)
            d = eng.dims
            logits = torch.empty(G, d.V, device=eng.device)
            lib = _lib.load()
            if self._tied:
                raise NotImplementedError("stand-alone training step with the tied head: use forward()")
            ow = eng.params.views["_output_layer.weight"]
            from ssc_runtime.cellops import _gemm
            _gemm(lib, [(h_dec.data_ptr(), d.H, ow.data_ptr(), ow.stride(0), d.H)], G, d.V, logits,
                  bias=eng.params.views["_output_layer.bias"])
            pm = (sentiment.reshape(G, 1).to(eng.device) * d.pm_scale).expand(G, self.z_space) if (
                sentiment is not None and d.pm_scale != 0.0) else torch.zeros(G, self.z_space, device=eng.device)
            plv = torch.full((G, self.z_space), float(torch.log(torch.tensor(d.prior_var))), device=eng.device)
            return logits, states, mean, log_var, pm, plv, alpha
        eng = self._engine()
        dev = eng.device
        B = image_features.size(0)
        G = previous_predictions.size(0)
        d = eng.dims
        obj_means = obj_atts if (d.kld_mode == 2 and torch.is_tensor(obj_atts)) else self._obj_means(obj_atts, B, image_features.size(1))
        ctx = self._image_context(image_features, obj_means)
        sent_rows = self._rows(sentiment, B, G) if sentiment is not None else None
        if self._eps_override is not None:
            eps = self._eps_override.pop(0)
        elif self.eps_source == "device":
            eps = torch.randn(G, self.z_space, device=dev)
        else:
            eps = torch.randn(G, self.z_space)
        # prior_mean / prior_var: what forward() derives from `sentiment` (updown_captioner.py:249-261) unless the caller hands its own
        # (:371-381); per image (B, Z) -> one row per beam, batch-major like the image features, or already (G, Z)
        def per_row(t):
            if t is None:
                return None
            t = t.to(dev, torch.float32)
            return t if t.size(0) == G else self._rows(t, B, G)
        pm_in, pv_in = per_row(prior_mean), per_row(prior_var)
        if self.simple_vae and pm_in is not None:
            pm_in = torch.zeros_like(pm_in)            # updown_cell.py:165-166
        pm_out = torch.empty(G, self.z_space, device=dev) if d.kld_mode == 2 else None
        lp, states, alpha = self._dec.step(ctx, previous_predictions, states, sent_rows, eps, prior_mean_out=pm_out,
                                           prior_mean=None if d.kld_mode == 2 else pm_in, prior_var=pv_in)
        if pm_out is not None:
            pm = pm_out                                  # the attention-pooled attribute means (updown_cell.py:160-163)
        elif pm_in is not None:
            pm = pm_in
        elif sent_rows is not None and d.pm_scale != 0.0:
            pm = (sent_rows * d.pm_scale).expand(G, self.z_space)
        else:
            pm = torch.zeros(G, self.z_space, device=dev)
        plv = pv_in.log() if pv_in is not None else torch.full((G, self.z_space), float(torch.log(torch.tensor(d.prior_var))), device=dev)
        return lp, states, pm, plv, alpha

    def _cell_forward(self, image_features, token_embedding, states, training, sentiment, prior_mean, prior_var, eps):
        """UpDownCell.forward backend (eval branch; updown_cell.py:86-231 with training=False)."""
        if training:  # stand-alone training-mode cell step (encoder LSTM + posterior sample), no autograd
            eng = self._engine()
            G = token_embedding.size(0)
            if eps is None:
                eps = torch.randn(G, self.z_space)
            h_dec, new_states, mean, log_var, alpha, _ = cell_train_step(eng.dims, eng.params.views,
                                                                         image_features.to(eng.device),
                                                                         token_embedding.to(eng.device), states, sentiment, eps)
            d = eng.dims
            pm = prior_mean if prior_mean is not None else torch.zeros(G, self.z_space, device=eng.device)
            if self.simple_vae:
                pm = torch.zeros_like(pm)
            pv = prior_var if prior_var is not None else torch.full((G, self.z_space), d.prior_var, device=eng.device)
            return h_dec, new_states, mean, log_var, pm, pv.log(), alpha
        eng = self._engine()
        G = token_embedding.size(0)
        B = image_features.size(0)
        ctx = self._image_context(image_features)
        if eps is None:
            eps = torch.randn(G, self.z_space)
        sent_rows = self._rows(sentiment, B, G) if sentiment is not None and sentiment.size(0) == B else sentiment
        d = eng.dims
        # the cell samples z from the prior it is HANDED (updown_cell.py:200-208)
        pm = prior_mean.to(eng.device, torch.float32) if prior_mean is not None else torch.zeros(G, self.z_space, device=eng.device)
        if self.simple_vae:
            pm = torch.zeros_like(pm)                  # updown_cell.py:165-166
        pv = prior_var.to(eng.device, torch.float32) if prior_var is not None else torch.full((G, self.z_space), d.prior_var, device=eng.device)
        _, new_states, alpha = self._dec.step_from_embedding(ctx, token_embedding, states, sent_rows, eps,
                                                             prior_mean=pm if prior_mean is not None else None,
                                                             prior_var=pv if prior_var is not None else None)
        return new_states["h_decoder"], new_states, pm, pv.log(), pm, pv.log(), alpha

    def _get_loss(self, logits: torch.Tensor, targets: torch.Tensor, target_mask: torch.Tensor) -> torch.Tensor:
        """(batch,) summed masked negative log-likelihood of the targets (updown_captioner.py:457-466: target length times allennlp's
        per-sequence average) on the HIP path (ssc_ce_fwd: row-wise log-sum-exp + NLL, the kernel inside ssc_train_fwd).  logits
        (B, T, V), targets (B, T) int64, target_mask (B, T).  The training forward computes this inside its fused call; this is the
        stand-alone entry the reference exposes.  No autograd."""
        lib = _lib.load()
        dev = logits.device
        if dev.type != "cuda":
            raise RuntimeError("UpDownCaptioner._get_loss runs on a ROCm GPU only (no CPU fallback)")
        B, T, V = logits.shape
        lg = logits.detach().to(torch.float32).transpose(0, 1).contiguous().view(T * B, V)     # time-major rows (t, b)
        tg = targets.to(dev, torch.int64).t().contiguous()
        w = target_mask.to(dev, torch.float32).t().contiguous()
        nvalid = w.sum(0).contiguous()
        lse = torch.empty(2 * T * B, dtype=torch.float32, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        lib.ssc_ce_fwd(_lib.ptr(lg), V, _lib.ptr(tg), _lib.ptr(w), _lib.ptr(nvalid), T, B, V, _lib.ptr(lse), _lib.ptr(loss), _lib.stream_ptr())
        return loss


class _HostRef:
    """Weak-ish back reference from the cell to its captioner that nn.Module will not register as a submodule."""

    def __init__(self, host):
        object.__setattr__(self, "_h", host)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_h"), name)

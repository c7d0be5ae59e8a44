"""Pure-torch CPU restatement of the Style-SeqCVAE ``var_updown`` hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Written from the algorithm
specification in SURVEY.md Appendix A; every function cites the reference
file:line (relative to /root/reference) whose behaviour it restates.

Conventions
-----------
* ``params``: dict keyed by the reference's ``state_dict`` names
  (``_updown_cell._attention_lstm_cell.weight_ih`` ...), fp32 CPU tensors.
* ``cfg``: :class:`OracleConfig` (sentiment_vae, simple_vae, prior_std, ...).
* ``eps``: explicit standard-normal noise ``(T, B, Z)`` (training) or
  ``(B_rows, Z)`` per decode step.  The reference draws it with
  ``torch.randn(var.shape)`` on the CPU generator once per step
  (var_updown/var_updown/modules/updown_cell.py:206).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

__all__ = [
    "OracleConfig",
    "masked_softmax",
    "masked_mean",
    "add_sentence_boundary_token_ids",
    "sequence_cross_entropy_with_logits",
    "lstm_cell",
    "average_image_features",
    "butd_attention",
    "cell_step",
    "decode_step",
    "output_logits",
    "prior_from_sentiment",
    "kld_step",
    "train_forward",
    "train_objective",
    "sgd_clip_step",
    "cbs_search",
    "select_best_beam",
    "select_best_beam_with_constraints",
    "eval_forward",
    "init_params",
    "param_shapes",
]

P_CELL = "_updown_cell."
P_ATT = P_CELL + "_attention_lstm_cell."
P_ENC = P_CELL + "_language_lstm_cell_encoder."
P_DEC = P_CELL + "_language_lstm_cell_decoder."
P_BUTD = P_CELL + "_butd_attention."


@dataclass
class OracleConfig:
    """Model hyper-parameters consumed on the hot path.

    Mirrors the keyword arguments of ``UpDownCaptioner.__init__``
    (var_updown/var_updown/models/updown_captioner.py:21-41).
    """

    vocab_size: int
    image_feature_size: int
    embedding_size: int
    hidden_size: int
    attention_projection_size: int
    z_space: int = 150
    max_caption_length: int = 20
    sentiment_vae: int = 0
    simple_vae: bool = False
    prior_std: float = 1.0
    senti_prior_multip: float = 1.0
    tied: bool = False  # E in {300, 600} in the reference (:75, :112-119)
    pad_index: int = 0  # "@@UNKNOWN@@"
    boundary_index: int = 1  # "@@BOUNDARY@@"
    beam_size: int = 5
    kld_weight: float = 750.0

    @property
    def senti_cols(self) -> int:
        """Extra conditioning columns on the language LSTMs (updown_cell.py:47-81)."""
        if self.simple_vae or self.sentiment_vae == 0:
            return 0
        if self.sentiment_vae == 1:
            return 1
        if self.sentiment_vae == 2:
            return 150  # the attention-pooled attribute means c (updown_cell.py:62-69: hard-coded 150; z_space must equal it)
        raise NotImplementedError(f"SENTIMENT_VAE={self.sentiment_vae}")


# --------------------------------------------------------------------------- #
# allennlp==0.8.4 tensor utilities (third-party, absent; restated semantics).  #
# Call sites: updown_cell.py:6,266 ; attention.py:6,93 ;                       #
#             updown_captioner.py:12,265,464.                                  #
# --------------------------------------------------------------------------- #
def masked_softmax(vector: torch.Tensor, mask: Optional[torch.Tensor], dim: int = -1) -> torch.Tensor:
    """allennlp.nn.util.masked_softmax (memory_efficient=False branch).

    Masked logits are zeroed (not -inf'd) before the softmax, the result is
    re-masked and renormalised with ``+1e-13``.
    """
    if mask is None:
        return torch.softmax(vector, dim=dim)
    m = mask.float()
    while m.dim() < vector.dim():
        m = m.unsqueeze(1)
    r = torch.softmax(vector * m, dim=dim)
    r = r * m
    return r / (r.sum(dim=dim, keepdim=True) + 1e-13)


def masked_mean(vector: torch.Tensor, mask: torch.Tensor, dim: int, eps: float = 1e-8) -> torch.Tensor:
    """allennlp.nn.util.masked_mean: sum of unmasked / clamp(count, min=eps)."""
    m = mask.float()
    s = torch.sum(vector * m, dim=dim)  # == masked_fill(1-mask, 0).sum for 0/1 masks
    n = torch.sum(m, dim=dim)
    return s / n.clamp(min=eps)


def add_sentence_boundary_token_ids(
    tensor: torch.Tensor, mask: torch.Tensor, begin: int, end: int
) -> Tuple[torch.Tensor, torch.Tensor]:
    """allennlp.nn.util.add_sentence_boundary_token_ids for 2-D id tensors.

    ``out[:, 0] = begin``; ``out[:, 1:-1] = tensor``; ``out[b, n_b + 1] = end``
    where ``n_b`` is the COUNT of unmasked tokens.
    """
    B, L = tensor.shape
    out = tensor.new_zeros((B, L + 2))
    out[:, 1:-1] = tensor
    out[:, 0] = begin
    n = mask.long().sum(dim=1)
    out[torch.arange(B), n + 1] = end
    return out, (out != 0).long()


def sequence_cross_entropy_with_logits(logits: torch.Tensor, targets: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """allennlp.nn.util.sequence_cross_entropy_with_logits(average=None)."""
    lp = torch.log_softmax(logits.reshape(-1, logits.size(-1)), dim=-1)
    nll = -lp.gather(1, targets.reshape(-1, 1).long()).view(*targets.shape)
    nll = nll * weights.float()
    return nll.sum(1) / (weights.float().sum(1) + 1e-13)


# --------------------------------------------------------------------------- #
# Cell pieces                                                                  #
# --------------------------------------------------------------------------- #
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell semantics (gate order i,f,g,o).  SURVEY §8(a)-10."""
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, gg, o = g.chunk(4, dim=1)
    i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
    c2 = f * c + i * gg
    h2 = o * torch.tanh(c2)
    return h2, c2


def average_image_features(feats: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """updown_cell.py:233-270: region mask from |v|-sum > 0, masked mean over regions."""
    mask = feats.abs().sum(dim=-1) > 0
    avg = masked_mean(feats, mask.unsqueeze(-1), dim=1)
    return avg, mask


def butd_attention(params, h1, feats, mask, pv=None):
    """updown-baseline/updown/modules/attention.py:36-97 (+ :99-125 projection)."""
    wq = params[P_BUTD + "_query_vector_projection_layer.weight"]
    wv = params[P_BUTD + "_image_features_projection_layer.weight"]
    wa = params[P_BUTD + "_attention_layer.weight"]
    q = h1 @ wq.t()
    if pv is None:
        pv = feats @ wv.t()
    logits = (torch.tanh(q.unsqueeze(1) + pv) @ wa.t()).squeeze(-1)
    return masked_softmax(logits, mask, dim=-1), pv


def zero_states(B: int, H: int, like: torch.Tensor) -> Dict[str, torch.Tensor]:
    z = like.new_zeros((B, H))
    return {k: z.clone() for k in ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")}


def cell_step(params, cfg: OracleConfig, feats, emb, states, training, sentiment, prior_mean, prior_var, eps,
              cache=None, obj_atts=None, return_prior=False):
    """One ``UpDownCell.forward`` (var_updown/var_updown/modules/updown_cell.py:86-231).

    ``cache`` (dict) carries the per-sequence terms the reference lru-caches
    (avg, mask, pv).  Returns (h_dec, states, mean, log_var, alpha)
    (+ the step's prior mean with ``return_prior``: for SENTIMENT_VAE=2 the
    attention-pooled ``obj_atts`` (B,R,150), :160-163, which also conditions both
    language LSTMs, :185-188,219-222).
    """
    B = feats.size(0)
    if cache is None:
        cache = {}
    if "avg" not in cache:
        cache["avg"], cache["mask"] = average_image_features(feats)
    avg, mask = cache["avg"], cache["mask"]
    if states is None:
        states = zero_states(B, cfg.hidden_size, feats)
    states = dict(states)
    hd_prev = states["h_decoder"]

    x_a = torch.cat([emb, avg, states["h1"], hd_prev], dim=1)  # :143-145
    states["h1"], states["c1"] = lstm_cell(
        x_a, states["h1"], states["c1"],
        params[P_ATT + "weight_ih"], params[P_ATT + "weight_hh"], params[P_ATT + "bias_ih"], params[P_ATT + "bias_hh"])
    alpha, cache["pv"] = butd_attention(params, states["h1"], feats, mask, cache.get("pv"))  # :151
    att = torch.sum(alpha.unsqueeze(-1) * feats, dim=1)  # :156-158

    if cfg.sentiment_vae == 2:
        prior_mean = torch.sum(alpha.unsqueeze(-1) * obj_atts, dim=1)  # :160-163
    if cfg.simple_vae:
        prior_mean = torch.zeros_like(prior_mean)  # :165-166
    s = cfg.senti_cols
    extra = [sentiment] if s == 1 else ([prior_mean] if s > 1 else [])  # latent_embedding == "glove": c = prior_mean (:169-170)
    if training:  # :176-198
        x_e = torch.cat([att, states["h1"], hd_prev] + extra, dim=1)
        states["h_encoder"], states["c_encoder"] = lstm_cell(
            x_e, states["h_encoder"], states["c_encoder"],
            params[P_ENC + "weight_ih"], params[P_ENC + "weight_hh"], params[P_ENC + "bias_ih"], params[P_ENC + "bias_hh"])
        mean = states["h_encoder"] @ params[P_CELL + "fc_mean.weight"].t() + params[P_CELL + "fc_mean.bias"]
        log_var = states["h_encoder"] @ params[P_CELL + "fc_log_var.weight"].t() + params[P_CELL + "fc_log_var.bias"]
        var = log_var.exp()
    else:  # :200-203
        mean, var = prior_mean, prior_var
        log_var = var.log()
    z = eps * var.sqrt() + mean  # :206-208
    x_d = torch.cat([att, states["h1"], hd_prev] + extra + [z], dim=1)  # :211-222
    states["h_decoder"], states["c_decoder"] = lstm_cell(
        x_d, states["h_decoder"], states["c_decoder"],
        params[P_DEC + "weight_ih"], params[P_DEC + "weight_hh"], params[P_DEC + "bias_ih"], params[P_DEC + "bias_hh"])
    if return_prior:
        return states["h_decoder"], states, mean, log_var, alpha, prior_mean
    return states["h_decoder"], states, mean, log_var, alpha


def output_logits(params, cfg: OracleConfig, h_dec):
    """updown_captioner.py:444-445 (+ :112-124 for the tied / untied heads)."""
    if cfg.tied:
        p = torch.tanh(h_dec @ params["_output_projection.0.weight"].t() + params["_output_projection.0.bias"])
        return p @ params["_embedding_layer.weight"].t()
    return h_dec @ params["_output_layer.weight"].t() + params["_output_layer.bias"]


def prior_from_sentiment(cfg: OracleConfig, sentiment, B, like):
    """updown_captioner.py:249-261."""
    if cfg.sentiment_vae == 0:
        prior_mean = like.new_zeros((B, cfg.z_space))
    elif cfg.sentiment_vae == 1:
        prior_mean = sentiment.repeat(1, cfg.z_space) * cfg.senti_prior_multip
    elif cfg.sentiment_vae == 2:
        prior_mean = like.new_zeros((B, cfg.z_space))  # :258-260 (replaced by the attention-pooled obj_atts in every cell step)
    else:
        raise NotImplementedError
    prior_var = (torch.ones_like(prior_mean) * cfg.prior_std).pow(2)
    return prior_mean, prior_var


def kld_step(cfg: OracleConfig, mean, log_var, prior_mean, prior_var):
    """updown_captioner.py:295-303."""
    if cfg.sentiment_vae == 0:
        return -0.5 * torch.sum(1 + log_var - mean.pow(2) - log_var.exp(), dim=1)
    k = 1 + log_var - prior_var.log() - ((mean - prior_mean).pow(2) + log_var.exp()) / (prior_var + 0.00001)
    return -0.5 * k.sum(1)


def embed(params, cfg: OracleConfig, tokens):
    return torch.nn.functional.embedding(tokens, params["_embedding_layer.weight"], padding_idx=cfg.pad_index)


def decode_step(params, cfg: OracleConfig, feats, tokens, states, training, sentiment, prior_mean, prior_var, eps,
                cache=None, obj_atts=None, return_prior=False):
    """``UpDownCaptioner._decode_step`` (updown_captioner.py:371-455).

    In eval mode the caller passes ``feats`` already expanded to one row per
    beam row, batch-major (batch-major for all).
    Returns (logits | log_probs, states, mean, log_var, alpha).
    """
    emb = embed(params, cfg, tokens)
    h_dec, states, mean, log_var, alpha, pm = cell_step(
        params, cfg, feats, emb, states, training, sentiment, prior_mean, prior_var, eps, cache, obj_atts, True)
    logits = output_logits(params, cfg, h_dec)
    out = logits if training else torch.log_softmax(logits, dim=1)
    if return_prior:
        return out, states, mean, log_var, alpha, pm
    return out, states, mean, log_var, alpha


def train_forward(params, cfg: OracleConfig, feats, caption_tokens, sentiment, eps, return_steps=False, obj_atts=None):
    """Training branch of ``UpDownCaptioner.forward`` (updown_captioner.py:228-323).

    Returns ``{"loss": (B,), "kld": (B,)}`` (+ per-step intermediates).
    ``obj_atts`` (B,R,150): per-region attribute means, SENTIMENT_VAE=2 only (what
    ``translate_obj_atts2obj_means`` :509-532 hands to the cell; the KL term uses the
    prior mean the cell RETURNS, :287-303).
    """
    B = feats.size(0)
    prior_mean, prior_var = prior_from_sentiment(cfg, sentiment, B, feats)
    tokens, _ = add_sentence_boundary_token_ids(
        caption_tokens, caption_tokens != cfg.pad_index, cfg.boundary_index, cfg.boundary_index)
    tokens_mask = tokens != cfg.pad_index
    T = tokens.size(1) - 1
    states, cache = None, {}
    step_logits, step_klds, steps = [], [], []
    for t in range(T):
        logits, states, mean, log_var, alpha, prior_mean = decode_step(
            params, cfg, feats, tokens[:, t], states, True, sentiment, prior_mean, prior_var, eps[t], cache, obj_atts, True)
        pm = prior_mean
        step_klds.append(kld_step(cfg, mean, log_var, pm, prior_var).unsqueeze(1))
        step_logits.append(logits.unsqueeze(1))
        if return_steps:
            steps.append({**{k: v for k, v in states.items()}, "alpha": alpha, "mean": mean, "log_var": log_var,
                          "logits": logits, "prior_mean": pm})
    logits = torch.cat(step_logits, 1)
    klds = torch.cat(step_klds, 1) * tokens_mask[:, 1:].float()
    tmask = tokens_mask[:, 1:].contiguous()
    lengths = tmask.sum(-1).float()
    loss = lengths * sequence_cross_entropy_with_logits(logits, tokens[:, 1:].contiguous(), tmask)  # :457-466
    out = {"loss": loss, "kld": klds.sum(1)}
    if return_steps:
        out["steps"] = steps
        out["tokens"] = tokens
    return out


def train_objective(out, cfg: OracleConfig):
    """var_updown/scripts/train.py:168-171."""
    return out["loss"].mean() + out["kld"].mean() / cfg.kld_weight


def sgd_clip_step(params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]],
                  momentum_buf: Dict[str, torch.Tensor], lr: float, momentum: float, weight_decay: float,
                  max_norm: float):
    """clip_grad_norm_ + torch.optim.SGD(momentum, weight_decay) step (train.py:126-131,173-175).

    torch>=2 semantics: parameters whose grad is None are skipped entirely
    (frozen decoder-LSTM row).  First step: buf = d_p.
    Returns (new_params, new_momentum, total_norm).
    """
    gs = [g for g in grads.values() if g is not None]
    total_norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
    coef = torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)
    new_p, new_m = {}, dict(momentum_buf)
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            new_p[k] = p.clone()
            continue
        d = g * coef + weight_decay * p
        if k in momentum_buf:
            buf = momentum * momentum_buf[k] + d
        else:
            buf = d.clone()
        new_m[k] = buf
        new_p[k] = p - lr * buf
    return new_p, new_m, total_norm


# --------------------------------------------------------------------------- #
# Constrained beam search (torch-1.1 semantics)                                #
# --------------------------------------------------------------------------- #
StepFn = Callable[[torch.Tensor, Optional[Dict[str, torch.Tensor]]], Tuple[torch.Tensor, Dict[str, torch.Tensor]]]


def cbs_search(start_predictions: torch.Tensor, start_state, step: StepFn, fsm: torch.Tensor, end_index: int,
               max_steps: int = 20, beam_size: int = 5, per_node_beam_size: Optional[int] = None,
               early_stop: bool = True):
    """``ConstrainedBeamSearch.search`` (updown-baseline/updown/modules/cbs.py:59-277).

    Restated with torch-1.1 semantics made explicit: uint8 masks act as
    booleans (:134-136, :204-206), ``/`` on int64 is floor division (:231).
    ``step(tokens, state) -> (log_probs (G,V), state)``; row order is
    (batch, fsm_state, beam) flattened.  Returns
    ``(predictions (B,S,beam,steps), log_probs (B,S,beam))``.
    """
    per_node = per_node_beam_size or beam_size
    B, S, _, V = fsm.shape
    fsm_b = fsm.bool()
    predictions: List[torch.Tensor] = []
    backpointers: List[torch.Tensor] = []

    start_lp, state = step(start_predictions, start_state)
    V = start_lp.size(-1)
    sp = start_lp.view(B, 1, V).expand(B, S, V).masked_fill(~fsm_b[:, 0, :, :], float("-inf"))  # :130-136
    last_lp, start_cls = sp.topk(beam_size)  # (B,S,beam)
    predictions.append(start_cls.reshape(B, -1))

    after_end = torch.full((1, V), float("-inf"))
    after_end[:, end_index] = 0.0

    def enlarge(t):  # cbs.py:10-17
        _, *rest = t.shape
        return t.view(B, 1, 1, *rest).expand(B, S, beam_size, *rest).reshape(-1, *rest)

    state = {k: enlarge(v) for k, v in state.items()}
    step_mask = fsm_b.view(B, S, S, 1, V).expand(B, S, S, beam_size, V)

    for _ in range(max_steps - 1):
        last = predictions[-1].reshape(B * beam_size * S)
        if early_stop and bool((last == end_index).all()):  # :167
            break
        lp, state = step(last, state)
        lp = torch.where(last.view(-1, 1).expand(-1, V) == end_index, after_end, lp).view(B, S, beam_size, V)
        r_cls = torch.zeros((B, S, beam_size), dtype=torch.long)
        r_lp = torch.zeros((B, S, beam_size))
        r_idx = torch.zeros((B, S, beam_size), dtype=torch.long)
        exp_last = last_lp.view(B, S, beam_size, 1).expand(B, S, beam_size, per_node)
        for i in range(S):  # :200-226
            slp = lp.masked_fill(~step_mask[:, :, i, :, :], -1e20)
            top_lp, cls = slp.topk(per_node)
            summed = (top_lp + exp_last).reshape(B, -1)
            cls = cls.reshape(B, -1)
            b_lp, b_idx = summed.topk(beam_size)
            r_cls[:, i, :] = cls.gather(1, b_idx)
            r_idx[:, i, :] = b_idx
            r_lp[:, i, :] = b_lp
        predictions.append(r_cls.view(B, -1))
        backpointer = torch.div(r_idx, per_node, rounding_mode="floor")  # :231 (int64 `/` in torch 1.1)
        backpointers.append(backpointer.view(B, -1))
        last_lp = r_lp.view(B, S, -1)

        def track_back(t):  # :236-250
            _, *rest = t.shape
            bp = backpointer.view(B, S * beam_size, *([1] * len(rest))).expand(B, S * beam_size, *rest)
            return t.reshape(B, S * beam_size, *rest).gather(1, bp).reshape(B * S * beam_size, *rest)

        state = {k: track_back(v) for k, v in state.items()}

    # back-trace (:252-277)
    if not backpointers:
        allp = predictions[0].unsqueeze(2)
        return allp.view(B, S, beam_size, -1), last_lp
    recon = [predictions[-1].unsqueeze(2)]
    cur = backpointers[-1]
    for t in range(len(predictions) - 2, 0, -1):
        recon.append(predictions[t].gather(1, cur).unsqueeze(2))
        cur = backpointers[t - 1].gather(1, cur)
    recon.append(predictions[0].gather(1, cur).unsqueeze(2))
    allp = torch.cat(list(reversed(recon)), 2)
    return allp.view(B, S, beam_size, -1), last_lp


def select_best_beam(beams, beam_log_probabilities):
    """updown-baseline/updown/utils/decoding.py:10-27."""
    return beams[:, 0, :]


def select_best_beam_with_constraints(beams, beam_log_probabilities, given_constraints, min_constraints_to_satisfy=2):
    """decoding.py:30-138, ``cbs_simple=True`` branch (the shipped yaml: CBS_SIMPLE True)."""
    B = beams.size(0)
    best = []
    for i in range(B):
        k = int(given_constraints[i])
        valid = [s for s in range(2 ** k) if bin(s).count("1") >= min(k, min_constraints_to_satisfy)]
        vb = beams[i, valid, 0, :]
        vlp = beam_log_probabilities[i, valid, 0]
        best.append(vb[int(torch.argmax(vlp))])
    return torch.stack(best).long()


def eval_forward(params, cfg: OracleConfig, feats, sentiment, fsm, num_constraints, eps_steps, beam_size=None,
                 min_constraints_to_satisfy=0, early_stop=True):
    """Eval branch of ``UpDownCaptioner.forward`` with CBS (updown_captioner.py:324-366).

    ``eps_steps[k]`` is the (rows_k, Z) noise of the k-th step call
    (rows_0 = B, rows_k = B*S*beam afterwards), batch-major rows.
    Returns ``{"predictions": (B, steps)}`` plus beams / log-probs for tests.
    """
    beam = beam_size or cfg.beam_size
    B = feats.size(0)
    prior_mean, prior_var = prior_from_sentiment(cfg, sentiment, B, feats)
    counter = {"k": 0}

    def step(tokens, state):
        G = tokens.size(0)
        rep = G // B
        f = feats.unsqueeze(1).expand(B, rep, *feats.shape[1:]).reshape(G, *feats.shape[1:])
        se = sentiment.unsqueeze(1).expand(B, rep, 1).reshape(G, 1) if sentiment is not None else None
        pm = prior_mean.unsqueeze(1).expand(B, rep, -1).reshape(G, -1)
        pvv = prior_var.unsqueeze(1).expand(B, rep, -1).reshape(G, -1)
        e = eps_steps[counter["k"]]
        counter["k"] += 1
        lp, st, _, _, _ = decode_step(params, cfg, f, tokens, state, False, se, pm, pvv, e)
        return lp, st

    start = torch.full((B,), cfg.boundary_index, dtype=torch.long)
    beams, lps = cbs_search(start, None, step, fsm, cfg.boundary_index, cfg.max_caption_length, beam,
                            (beam // 2) or beam, early_stop)
    best = select_best_beam_with_constraints(beams, lps, num_constraints, min_constraints_to_satisfy)
    return {"predictions": best, "beams": beams, "log_probs": lps, "num_step_calls": counter["k"]}


# --------------------------------------------------------------------------- #
# Parameter construction (default torch inits; updown_captioner.py:21-139)     #
# --------------------------------------------------------------------------- #
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    V, E, H, A, F, Z, s = (cfg.vocab_size, cfg.embedding_size, cfg.hidden_size, cfg.attention_projection_size,
                           cfg.image_feature_size, cfg.z_space, cfg.senti_cols)
    sh = {"_embedding_layer.weight": (V, E)}
    for pre, k in ((P_ATT, E + F + 2 * H), (P_ENC, s + F + 2 * H), (P_DEC, s + F + 2 * H + Z)):
        sh[pre + "weight_ih"] = (4 * H, k)
        sh[pre + "weight_hh"] = (4 * H, H)
        sh[pre + "bias_ih"] = (4 * H,)
        sh[pre + "bias_hh"] = (4 * H,)
    sh[P_BUTD + "_query_vector_projection_layer.weight"] = (A, H)
    sh[P_BUTD + "_image_features_projection_layer.weight"] = (A, F)
    sh[P_BUTD + "_attention_layer.weight"] = (1, A)
    sh[P_CELL + "fc_mean.weight"] = (Z, H)
    sh[P_CELL + "fc_mean.bias"] = (Z,)
    sh[P_CELL + "fc_log_var.weight"] = (Z, H)
    sh[P_CELL + "fc_log_var.bias"] = (Z,)
    if cfg.tied:
        sh["_output_projection.0.weight"] = (E, H)
        sh["_output_projection.0.bias"] = (E,)
    else:
        sh["_output_layer.weight"] = (V, H)
        sh["_output_layer.bias"] = (V,)
    return sh


def init_params(cfg: OracleConfig, seed: int = 2, scale: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Seeded synthetic parameters with torch's default init *distributions*
    (LSTM/Linear U(+-1/sqrt(fan)), Embedding N(0,1), pad row zero).  Not the
    reference's exact RNG consumption order - fixtures carry explicit weights."""
    g = torch.Generator().manual_seed(seed)
    H = cfg.hidden_size
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name == "_embedding_layer.weight":
            w = torch.randn(shape, generator=g)
            w[cfg.pad_index] = 0
        else:
            if "lstm" in name:
                bound = 1.0 / math.sqrt(H)
            elif len(shape) == 2:
                bound = 1.0 / math.sqrt(shape[1])
            else:  # Linear bias: fan_in of its weight
                fan = {"fc_mean.bias": H, "fc_log_var.bias": H, "_output_projection.0.bias": H,
                       "_output_layer.bias": H}[name.replace(P_CELL, "")]
                bound = 1.0 / math.sqrt(fan)
            w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        out[name] = w * scale if scale is not None and name != "_embedding_layer.weight" else w
    return out

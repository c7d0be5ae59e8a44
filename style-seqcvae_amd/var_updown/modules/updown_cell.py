"""UpDownCell: same constructor, attributes, state_dict keys and forward signature as the reference
(var_updown/var_updown/modules/updown_cell.py:11-231).  The sub-modules are parameter holders created in the
reference's order (identical default-init RNG stream); the arithmetic runs in libssc_hip.so."""
from typing import Dict, Optional

import torch
from torch import nn

from .attention import BottomUpTopDownAttention


class UpDownCell(nn.Module):
    def __init__(self, image_feature_size: int, embedding_size: int, hidden_size: int, attention_projection_size: int,
                 z_space: int, sentiment_vae: int, simple_vae, device, latent_embedding):
        super().__init__()
        self.image_feature_size = image_feature_size
        self.embedding_size = embedding_size
        self.hidden_size = hidden_size
        self.attention_projection_size = attention_projection_size
        self.device = device
        F, E, H = image_feature_size, embedding_size, hidden_size
        self._attention_lstm_cell = nn.LSTMCell(E + F + 2 * H, H)
        self._butd_attention = BottomUpTopDownAttention(H, F, attention_projection_size)
        self.z_space = z_space
        self.sentiment_vae = sentiment_vae
        self.simple_vae = simple_vae
        self.latent_embedding = latent_embedding
        # extra conditioning columns (updown_cell.py:47-72)
        if sentiment_vae == 0:
            s = 0
        elif latent_embedding == "senti_word_net" or sentiment_vae == 1:
            s = 1
        elif sentiment_vae == 2:
            s = 150
        else:
            raise NotImplementedError()
        self._language_lstm_cell_encoder = nn.LSTMCell(s + F + 2 * H, H)
        self._language_lstm_cell_decoder = nn.LSTMCell(s + F + 2 * H + z_space, H)
        if self.simple_vae:  # re-created without the conditioning columns (updown_cell.py:74-81)
            s = 0
            self._language_lstm_cell_encoder = nn.LSTMCell(F + 2 * H, H)
            self._language_lstm_cell_decoder = nn.LSTMCell(F + 2 * H + z_space, H)
        self.senti_cols = s
        self.fc_mean = nn.Linear(H, z_space)
        self.fc_log_var = nn.Linear(H, z_space)
        self._host = None  # set by UpDownCaptioner: provides the engine over the flat parameter store

    def forward(self, image_features: torch.Tensor, obj_atts, token_embedding: torch.Tensor,
                states: Optional[Dict[str, torch.Tensor]] = None, training=True, sentiment=None, attrib_cond=None,
                prior_mean=None, prior_var=None, eps: Optional[torch.Tensor] = None):
        """One cell step -> (h_decoder, states, mean, log_var, prior_mean, log(prior_var), attention_weights)
        (updown_cell.py:231).  Runs the HIP decode step without autograd; the differentiable training path is the
        fused sequence kernel behind UpDownCaptioner.forward.  `eps` (G,Z) may be injected; default: CPU
        torch.randn as in updown_cell.py:206."""
        if self._host is not None and (self.sentiment_vae != 2 or self.simple_vae):
            return self._host._cell_forward(image_features, token_embedding, states, training, sentiment, prior_mean,
                                            prior_var, eps)
        return self._standalone_forward(image_features, obj_atts, token_embedding, states, training, sentiment, prior_mean,
                                        prior_var, eps)

    def _standalone_forward(self, image_features, obj_atts, token_embedding, states, training, sentiment, prior_mean,
                            prior_var, eps):
        """The cell on its own parameters through the op-level C ABI (ssc_runtime/cellops.py): what a bare `UpDownCell`
        (no captioner around it) runs, and the only path of SENTIMENT_VAE = 2 - whose captioner wiring cannot be constructed
        in the reference either (it reads /path/to/sentiglove10.pkl, updown_captioner.py:79) while the cell is complete:
        c = sum_r alpha_r obj_atts_r conditions both language LSTMs and is the prior mean (updown_cell.py:160-163,185-188,
        219-222)."""
        from ssc_runtime.cellops import cell_train_step
        from ssc_runtime.engine import ModelDims
        w = self._attention_lstm_cell.weight_ih
        if not w.is_cuda:
            raise RuntimeError("UpDownCell.forward runs on the HIP path only (no CPU fallback): move the module to a ROCm device")
        dev = w.device
        P = {"_updown_cell." + n: p.detach() for n, p in self.named_parameters()}
        G = token_embedding.size(0)
        Z = self.z_space
        dims = ModelDims(V=2, E=self.embedding_size, H=self.hidden_size, A=self.attention_projection_size,
                         F=self.image_feature_size, Z=Z, S=self.senti_cols)
        if eps is None:
            eps = torch.randn(G, Z)          # CPU generator, as updown_cell.py:206
        if self.latent_embedding not in ("glove", "senti_word_net"):
            raise NotImplementedError()
        sv2 = self.sentiment_vae == 2 and not self.simple_vae
        # the one conditioning column is `sentiment` in mode 1; in mode 2 with "senti_word_net" it is the pooled prior mean's first
        # entry (updown_cell.py:160-163,171-172), which cell_train_step derives from obj_atts
        sent = sentiment.reshape(G) if (sentiment is not None and self.senti_cols == 1 and self.sentiment_vae == 1) else None
        pm_in = None if prior_mean is None else (torch.zeros_like(prior_mean) if self.simple_vae else prior_mean)
        h_dec, new_states, mean, log_var, alpha, cond = cell_train_step(
            dims, P, image_features.to(dev), token_embedding.to(dev), states, sent, eps, obj_atts=obj_atts if sv2 else None, training=training,
            prior_mean=pm_in, prior_var=prior_var)
        pm = cond if cond is not None else (pm_in.to(dev) if pm_in is not None else torch.zeros(G, Z, device=dev))
        if self.simple_vae:
            pm = torch.zeros_like(pm)
        pv = prior_var.to(dev) if prior_var is not None else torch.ones(G, Z, device=dev)
        return h_dec, new_states, mean, log_var, pm, pv.log(), alpha

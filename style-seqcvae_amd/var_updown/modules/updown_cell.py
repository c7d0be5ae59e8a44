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
        if self._host is None:
            raise RuntimeError("UpDownCell.forward needs its UpDownCaptioner host (engine over the parameter store)")
        return self._host._cell_forward(image_features, token_embedding, states, training, sentiment, prior_mean,
                                        prior_var, eps)

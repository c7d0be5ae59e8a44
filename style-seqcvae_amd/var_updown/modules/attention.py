"""Parameter holder mirroring updown-baseline/updown/modules/attention.py:27-34 (three bias-free Linear layers,
same attribute names => same state_dict keys and the same default-init RNG consumption)."""
from torch import nn


class BottomUpTopDownAttention(nn.Module):
    def __init__(self, query_size: int, image_feature_size: int, projection_size: int):
        super().__init__()
        self._query_vector_projection_layer = nn.Linear(query_size, projection_size, bias=False)
        self._image_features_projection_layer = nn.Linear(image_feature_size, projection_size, bias=False)
        self._attention_layer = nn.Linear(projection_size, 1, bias=False)

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise RuntimeError("BottomUpTopDownAttention is fused into the HIP attention step (ssc_attn_fwd); "
                           "call UpDownCell / UpDownCaptioner instead")

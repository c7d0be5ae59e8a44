from .attention import BottomUpTopDownAttention
from .cbs import ConstrainedBeamSearch
from .updown_cell import UpDownCell

__all__ = ["UpDownCell", "BottomUpTopDownAttention", "ConstrainedBeamSearch"]

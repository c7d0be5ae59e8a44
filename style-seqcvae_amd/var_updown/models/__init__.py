from .updown_captioner import UpDownCaptioner

__all__ = ["UpDownCaptioner"]

from .helper import to_2tuple, Format, nchw_to

__all__ = ["to_2tuple", "Format", "nchw_to"]

"""GUPPI frames (guppi/frame.py in the reference): one ASCII-card header plus
one int8 payload block; always valid."""
from ..base.frame import block_frame_class
from .header import GUPPIHeader
from .payload import GUPPIPayload

__all__ = ['GUPPIFrame']

GUPPIFrame = block_frame_class(
    'GUPPIFrame', GUPPIHeader, GUPPIPayload,
    "GUPPI frame: header cards + (channels-first or time-first) int8 block.")

"""DADA frames (dada/frame.py in the reference): one 4096-byte ASCII header
plus one int8 payload; always valid."""
from ..base.frame import block_frame_class
from .header import DADAHeader
from .payload import DADAPayload

__all__ = ['DADAFrame']

DADAFrame = block_frame_class(
    'DADAFrame', DADAHeader, DADAPayload,
    "DADA frame: ASCII header + int8 payload (MKBF heaps when INSTRUMENT says so).")

"""GMRT GSB format: GPU-decoded reader with the reference's call shapes."""
from .header import GSBHeader
from .payload import GSBPayload
from .frame import GSBFrame
from .base import GSBStreamReader, GSBStreamWriter, open

__all__ = ['GSBHeader', 'GSBPayload', 'GSBFrame', 'GSBStreamReader',
           'GSBStreamWriter', 'open']

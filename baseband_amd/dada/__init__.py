"""DADA format: GPU-decoded reader with the reference's call shapes."""
from .header import DADAHeader
from .payload import DADAPayload, MKBFPayload
from .frame import DADAFrame
from .base import DADAFileReader, DADAStreamReader, DADAFileNameSequencer, open

__all__ = ['DADAFileNameSequencer', 'DADAHeader', 'DADAPayload', 'MKBFPayload', 'DADAFrame',
           'DADAFileReader', 'DADAStreamReader', 'open']

"""DADA format: GPU-decoded reader with the reference's call shapes."""
from .header import DADAHeader
from .payload import DADAPayload, MKBFPayload
from .frame import DADAFrame
from .base import DADAFileWriter, DADAFileReader, DADAStreamReader, DADAStreamWriter, DADAFileNameSequencer, open

__all__ = ['DADAFileWriter', 'DADAStreamWriter', 'DADAFileNameSequencer', 'DADAHeader', 'DADAPayload', 'MKBFPayload', 'DADAFrame',
           'DADAFileReader', 'DADAStreamReader', 'open']

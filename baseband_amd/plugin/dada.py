"""baseband.io entry point dada_hip: `baseband_amd.dada` with the reference's types at the seam."""
from ._proxy import make_module_api

open, info = make_module_api('dada')
__all__ = ['open', 'info']

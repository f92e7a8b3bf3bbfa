"""baseband.io entry point guppi_hip: `baseband_amd.guppi` with the reference's types at the seam."""
from ._proxy import make_module_api

open, info = make_module_api('guppi')
__all__ = ['open', 'info']

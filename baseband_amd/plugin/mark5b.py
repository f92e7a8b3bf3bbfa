"""baseband.io entry point mark5b_hip: `baseband_amd.mark5b` with the reference's types at the seam."""
from ._proxy import make_module_api

open, info = make_module_api('mark5b')
__all__ = ['open', 'info']

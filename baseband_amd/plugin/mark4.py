"""baseband.io entry point mark4_hip: `baseband_amd.mark4` with the reference's types at the seam."""
from ._proxy import make_module_api

open, info = make_module_api('mark4')
__all__ = ['open', 'info']

"""baseband.io entry point gsb_hip: `baseband_amd.gsb` with the reference's types at the seam."""
from ._proxy import make_module_api

open, info = make_module_api('gsb')
__all__ = ['open', 'info']

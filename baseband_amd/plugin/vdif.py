"""baseband.io entry point vdif_hip: `baseband_amd.vdif` with the reference's types at the seam."""
from ._proxy import make_module_api

open, info = make_module_api('vdif')
__all__ = ['open', 'info']

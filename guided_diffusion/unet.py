"""Alias of dps_ttc_amd.unet (reference module path guided_diffusion/unet.py)."""
import sys

from dps_ttc_amd import unet as _impl

sys.modules[__name__] = _impl      # the same module object: registries and monkey-patches are shared

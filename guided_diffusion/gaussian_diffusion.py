"""Alias of dps_ttc_amd.gaussian_diffusion (reference module path guided_diffusion/gaussian_diffusion.py)."""
import sys

from dps_ttc_amd import gaussian_diffusion as _impl

sys.modules[__name__] = _impl      # the same module object: registries and monkey-patches are shared

"""Alias of dps_ttc_amd.posterior_mean_variance (reference module path guided_diffusion/posterior_mean_variance.py)."""
import sys

from dps_ttc_amd import posterior_mean_variance as _impl

sys.modules[__name__] = _impl      # the same module object: registries and monkey-patches are shared

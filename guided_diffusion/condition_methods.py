"""Alias of dps_ttc_amd.condition_methods (reference module path guided_diffusion/condition_methods.py)."""
import sys

from dps_ttc_amd import condition_methods as _impl

sys.modules[__name__] = _impl      # the same module object: registries and monkey-patches are shared

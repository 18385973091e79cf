"""Alias of dps_ttc_amd.measurements (reference module path guided_diffusion/measurements.py)."""
import sys

from dps_ttc_amd import measurements as _impl

sys.modules[__name__] = _impl      # the same module object: registries and monkey-patches are shared

"""dps_ttc_amd -- MI355X-native hot path of vishnutez/dps-ttc (batched DPS test-time-compute loop).

Same registry API as the reference's guided_diffusion package:

    from dps_ttc_amd.measurements import get_operator, get_noise
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import create_sampler

All per-step device work outside the UNet is hand-written HIP for gfx950 behind the C ABI
in include/dpsx.h (dps_ttc_amd/lib/libdpsx.so).  There is no CPU or torch-op fallback.
"""
from ._lib import SO_PATH, DpsxError, lib  # noqa: F401

__all__ = ["lib", "SO_PATH", "DpsxError"]

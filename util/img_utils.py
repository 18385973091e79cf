"""Alias of dps_ttc_amd.img_utils (reference module path util/img_utils.py: clear_color, mask_generator)."""
import sys

from dps_ttc_amd import img_utils as _impl

sys.modules[__name__] = _impl

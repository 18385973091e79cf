"""Alias of dps_ttc_amd.data (reference module path data/dataloader.py: get_dataset, get_dataloader)."""
import sys

from dps_ttc_amd import data as _impl

sys.modules[__name__] = _impl

"""Reference module paths `util.*` -> dps_ttc_amd (see guided_diffusion/__init__.py)."""

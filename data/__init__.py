"""Reference module path `data.dataloader` -> dps_ttc_amd.data (see guided_diffusion/__init__.py)."""

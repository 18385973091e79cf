"""Drop-in module paths of the reference package (`guided_diffusion.*`): with this repository's root on
sys.path, the import block of the reference's sample_condition_batched_ttc.py (:11-18) resolves to the MI355X
hot path unchanged.  Every module here is a thin re-export of its `dps_ttc_amd` counterpart -- same objects, same
registries (route A)."""

"""Drop-in for the reference import path `from model.smap import SMAP` (model/smap.py:313)."""
from smap_amd.model.smap import SMAP  # noqa: F401

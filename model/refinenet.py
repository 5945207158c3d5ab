"""Drop-in for `from model.refinenet import RefineNet` (model/refinenet.py:29)."""
from smap_amd.model.refinenet import RefineNet, RefineNet_base  # noqa: F401

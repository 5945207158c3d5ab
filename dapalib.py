"""`import dapalib` drop-in (reference: extensions/association.cpp:236-241, built by extensions/setup.py).

Thin re-export of smap_amd.dapalib, whose functions run the HIP kernels of libsmap_hip.so.
"""
from smap_amd.dapalib import (connect, extract, connect_batch, extract_batch, lift_batch,  # noqa: F401
                              refine_batch, refine_mlp, scale_hms_, flip_merge_, register_gt_batch)

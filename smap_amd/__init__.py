"""smap_amd -- MI355X (gfx950) native implementation of the SMAP inference hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/smap_hip.h),
the ctypes loader, and the host-side mirrors of the reference interface
(`dapalib`, `model.smap.SMAP`, `model.refinenet.RefineNet`, `exps/stage3_root2`).
"""
__all__ = ["lib"]

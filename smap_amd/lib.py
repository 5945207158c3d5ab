"""ctypes loader for libsmap_hip.so -- the ONLY compute backend of this package.

There is deliberately no fallback: if the HIP library is missing or a symbol of
include/smap_hip.h is absent, importing / calling fails loudly.
"""
import ctypes as C
import os

# torch first: it ships its own libamdhip64.so (SONAME libamdhip64.so.7).  libsmap_hip.so must bind
# to THAT runtime instance (streams / device pointers are handed over from PyTorch); loading our
# library before torch would pull a second HIP runtime from /opt/rocm into the process.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SMAP_HIP_LIB") or os.path.join(HERE, "libsmap_hip.so")   # env override: kernel experiments only

# every symbol include/smap_hip.h declares
SYMBOLS = [
    "smap_version", "smap_scale_hms", "smap_flip_merge", "smap_nms", "smap_paf_score", "smap_group", "smap_lift",
    "smap_refine", "smap_register_gt", "smap_lift_gt", "smap_refine_gt", "smap_refine_mlp", "smap_preprocess", "smap_sizeof_op", "smap_conv_tile_dims", "smap_conv_tile_bk", "smap_conv_tile_tail_bn", "smap_plan_create", "smap_plan_destroy", "smap_plan_run", "smap_plan_run_range",
    "smap_plan_run_inputs", "smap_workspace_bytes", "smap_plan_create_from_blob", "smap_plan_set_lanes",
    "smap_nms_workspace_bytes", "smap_nms_ws",
]
MAX_INPUTS = 8                         # SMAP_MAX_INPUTS


class SmapOp(C.Structure):
    """Mirror of `struct smap_op` (include/smap_hip.h)."""
    _fields_ = [
        ("kind", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("in_stride_c", C.c_int32), ("in_c_off", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("relu", C.c_int32), ("cout_pad", C.c_int32), ("out_stride_c", C.c_int32),
        ("out_c_off", C.c_int32), ("out_fp32", C.c_int32), ("tile", C.c_int32), ("n_aux", C.c_int32),
        ("in_off", C.c_int64), ("out_off", C.c_int64),
        ("w_off", C.c_int64), ("bias_off", C.c_int64),
        ("res_off", C.c_int64), ("add1_off", C.c_int64), ("add2_off", C.c_int64),
        ("aux_off", C.c_int64 * 3), ("aux_h", C.c_int32 * 3), ("aux_w", C.c_int32 * 3),
        ("ext_off", C.c_int64),
        ("precision", C.c_int32), ("acc_scale", C.c_float),
        ("flip_from", C.c_int32), ("w_pairs", C.c_int32), ("status_off", C.c_int32),
        ("tail_cout", C.c_int32), ("tail_cout_pad", C.c_int32), ("tail_acc_scale", C.c_float),
        ("tail_w_off", C.c_int64), ("tail_bias_off", C.c_int64),
        ("head_cin", C.c_int32), ("head_acc_scale", C.c_float), ("head_w_off", C.c_int64), ("head_bias_off", C.c_int64),
        ("short_w_off", C.c_int64), ("short_acc_scale", C.c_float), ("scale_hms", C.c_int32),
        ("seg_n", C.c_int32 * 2), ("seg_cout", C.c_int32 * 2), ("seg_relu", C.c_int32 * 2), ("seg_out_stride_c", C.c_int32 * 2),
        ("seg_acc_scale", C.c_float * 2), ("seg_out_off", C.c_int64 * 2),
        ("ksplit", C.c_int32), ("reserved1", C.c_int32), ("kpart_off", C.c_int64), ("kcount_off", C.c_int64),
        ("lane", C.c_int32), ("n_wait", C.c_int32), ("wait_op", C.c_int32 * 4),
        ("in2_off", C.c_int64), ("in2_H", C.c_int32), ("in2_W", C.c_int32), ("in2_C", C.c_int32), ("in2_stride_c", C.c_int32),
        ("in2_stride", C.c_int32), ("in2_mode", C.c_int32), ("in2_acc_scale", C.c_float), ("in2_bias_off", C.c_int64), ("tap_n", C.c_int32), ("tap_scale", C.c_float), ("tap_w_off", C.c_int64),
    ]


class BlobInfo(C.Structure):
    """Mirror of `struct smap_blob_info`."""
    _fields_ = [("frames", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("n_hms", C.c_int32), ("n_det", C.c_int32), ("n_root", C.c_int32), ("precision", C.c_int32), ("reserved", C.c_int32),
                ("arena_bytes", C.c_int64), ("out_bytes", C.c_int64), ("weights_offset", C.c_int64), ("weights_bytes", C.c_int64),
                ("hms_off", C.c_int64), ("det_off", C.c_int64), ("root_off", C.c_int64), ("status_off", C.c_int64)]


BLOB_VERSION = 2          # include/smap_hip.h SMAP_BLOB_VERSION


class BlobHeader(C.Structure):
    """Mirror of `struct smap_blob_header`."""
    _fields_ = [("magic", C.c_char * 8), ("version", C.c_uint32), ("sizeof_op", C.c_uint32), ("header_bytes", C.c_uint32),
                ("n_ops", C.c_int32), ("ops_offset", C.c_int64), ("weights_offset", C.c_int64), ("weights_bytes", C.c_int64),
                ("arena_bytes", C.c_int64), ("out_bytes", C.c_int64), ("info", BlobInfo)]


_lib = None


class SmapError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -m smap_amd.build` "
            "(hipcc --offload-arch=gfx950). smap_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(SO_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{SO_PATH} lacks symbols {missing}; rebuild with `python -m smap_amd.build --force`")
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    lib.smap_version.restype = C.c_char_p
    lib.smap_scale_hms.argtypes = [vp, ip, ip, ip, vp]
    lib.smap_flip_merge.argtypes = [vp, vp, C.POINTER(C.c_int), ip, ip, ip, vp]
    lib.smap_nms.argtypes = [vp, ip, ip, ip, ip, fp, vp, vp]
    lib.smap_nms_workspace_bytes.argtypes = [ip, ip, ip]
    lib.smap_nms_ws.argtypes = [vp, ip, ip, ip, ip, fp, vp, vp, C.c_int64, vp]
    lib.smap_paf_score.argtypes = [vp, vp, ip, ip, ip, vp, vp]
    lib.smap_group.argtypes = [vp, vp, vp, ip, ip, ip, ip, ip, vp, vp, vp]
    lib.smap_lift.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp, vp, vp, vp]
    lib.smap_refine.argtypes = [vp, vp, vp, ip, C.POINTER(vp), C.POINTER(vp), vp, vp]
    lib.smap_refine_mlp.argtypes = [vp, ip, C.POINTER(vp), C.POINTER(vp), vp, vp]
    lib.smap_register_gt.argtypes = [vp, vp, vp, vp, ip, ip, vp, vp, vp]
    lib.smap_lift_gt.argtypes = lib.smap_lift.argtypes
    lib.smap_refine_gt.argtypes = lib.smap_refine.argtypes
    lib.smap_preprocess.argtypes = [vp, ip, ip, ip, ip, ip, ip, vp, ip, ip, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double, C.c_double, vp]
    lib.smap_conv_tile_dims.argtypes = [ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.smap_conv_tile_bk.argtypes = [ip, ip]
    lib.smap_conv_tile_tail_bn.argtypes = [ip]
    lib.smap_plan_create.argtypes = [C.POINTER(SmapOp), ip, C.POINTER(vp)]
    lib.smap_plan_destroy.argtypes = [vp]
    lib.smap_plan_destroy.restype = None
    lib.smap_plan_run.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.smap_plan_run_range.argtypes = [vp, ip, ip, vp, vp, vp, vp, vp]
    lib.smap_plan_run_inputs.argtypes = [vp, C.POINTER(vp), ip, vp, vp, vp, vp]
    lib.smap_plan_set_lanes.argtypes = [vp, ip]
    lib.smap_workspace_bytes.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.smap_plan_create_from_blob.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(BlobInfo)]
    for s in SYMBOLS:
        if s not in ("smap_version", "smap_plan_destroy", "smap_nms_workspace_bytes"):  # everything else returns int
            getattr(lib, s).restype = ip
    lib.smap_nms_workspace_bytes.restype = C.c_int64
    if lib.smap_sizeof_op() != C.sizeof(SmapOp):
        raise ImportError(f"smap_op layout mismatch: C {lib.smap_sizeof_op()} vs ctypes {C.sizeof(SmapOp)}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        detail = "argument error" if rc == -1 else f"hipError_t {-rc - 1000}" if rc <= -1000 else f"code {rc}"
        raise SmapError(f"{what} failed: {detail}")


def version():
    return load().smap_version().decode()

"""ctypes binding of libbenerf_hip.so (C ABI declared in include/benerf_hip.h).

The product path has NO fallback: if the library is missing or a symbol cannot be
resolved this module raises, and every kernel wrapper raises on a non-zero return code.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# BENERF_HIP_LIB: development override (kernel variants built next to the shipped library)
LIB_PATH = os.environ.get("BENERF_HIP_LIB") or os.path.join(_HERE, "libbenerf_hip.so")

NLAYERS = 12
L_VIEWS, L_FEAT, L_ALPHA, L_RGB = 8, 9, 10, 11
LOSS_NSTATS = 16
# status words of the split-f16 range guard (include/benerf_hip.h)
ST_ACT, ST_GRAD, ST_MODE, ST_AUTO, ST_SKIP, ST_SKIPPED, ST_CONSECUTIVE, ST_STEPS, ST_LAST_ACT, ST_LAST_GRAD, ST_STEP_SCRATCH, ST_WORDS = \
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 16
ST_SKIPPED_TOTAL, ST_MAX_CONSECUTIVE = 12, 13


class MlpParams(Structure):
    _fields_ = [("w", c_void_p * NLAYERS), ("b", c_void_p * NLAYERS), ("pe_weights", c_void_p)]


class MlpGrads(Structure):
    _fields_ = [("w", c_void_p * NLAYERS), ("b", c_void_p * NLAYERS)]


class LossCfg(Structure):
    _fields_ = [("channels", c_int32), ("linlog", c_int32), ("n_evt_pix", c_int32), ("n_rgb_pix", c_int32),
                ("n_poses", c_int32), ("n_evt_pix_global", c_int32), ("n_rgb_pix_global", c_int32),
                ("event_threshold", c_float), ("event_coeff", c_float), ("rgb_coeff", c_float)]


P = c_void_p
ABI_VERSION = 103       # include/benerf_hip.h: BENERF_ABI_VERSION
_SIGNATURES = {
    "benerf_version": (c_int, []),
    "benerf_last_error": (c_char_p, []),
    "benerf_spline_poses_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "benerf_spline_poses_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P]),
    "benerf_spline_poses_fwd_pair": (c_int, [P, P, P, c_int, P, c_int, c_int, P, P, P]),
    "benerf_spline_poses_bwd_pair": (c_int, [P, P, P, c_int, P, c_int, c_int, P, P, P, P, P, P]),
    "benerf_spline_op_fwd": (c_int, [c_int, P, c_int64, P, P]),
    "benerf_spline_op_bwd": (c_int, [c_int, P, c_int64, P, P, P]),
    "benerf_rays_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P, P, P, P, P]),
    "benerf_rays_bwd_workspace_floats": (c_size_t, [c_int, c_int]),
    "benerf_rays_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P, P, P, P, P, P, c_size_t, P]),
    "benerf_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "benerf_stratified_z": (c_int, [c_int, c_int, c_float, c_float, P, c_uint64, c_uint64, P, P]),
    "benerf_ray_grad_reduce": (c_int, [c_int, c_int, P, P, P, c_int, P, P, P, P]),
    "benerf_mlp_packed_floats": (c_size_t, []),
    "benerf_mlp_pack_weights": (c_int, [POINTER(MlpParams), c_int, P, P]),
    "benerf_mlp_pack_weights_pair": (c_int, [POINTER(MlpParams), P, POINTER(MlpParams), P, c_int, P]),
    "benerf_mlp_act_floats": (c_size_t, [c_int64]),
    "benerf_mlp_dact_floats_per_point": (c_size_t, []),
    "benerf_mlp_dact_floats": (c_size_t, [c_int64]),
    "benerf_mlp_act_floats_for": (c_size_t, [c_int64, c_int]),
    "benerf_mlp_dact_floats_for": (c_size_t, [c_int64, c_int]),
    "benerf_mlp_dw_workspace_floats": (c_size_t, [c_int64]),
    "benerf_mlp_fwd": (c_int, [POINTER(MlpParams), P, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P, P]),
    "benerf_mlp_status_check": (c_int, [P, P]),
    "benerf_mlp_h8_roundtrip": (c_int, [P, c_int64, c_int, P, P, P, P]),
    "benerf_step_gate": (c_int, [P, P, c_int, P]),
    "benerf_mlp_bwd_dx": (c_int, [POINTER(MlpParams), P, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P]),
    "benerf_mlp_bwd_dw": (c_int, [POINTER(MlpParams), c_int, c_int, c_int, P, P, P, P, c_size_t, POINTER(MlpGrads), c_int, c_int, P, P]),
    "benerf_composite_fwd": (c_int, [P, P, P, P, c_float, c_uint64, c_uint64, c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "benerf_composite_bwd": (c_int, [P, P, P, P, c_float, c_uint64, c_uint64, c_int, c_int, c_int, P, P, P, P, P, P,
                                     c_int, P, P]),
    "benerf_sample_pdf_merge": (c_int, [P, P, P, c_uint64, c_uint64, c_int, c_int, c_int, P, P, P, P]),
    "benerf_sample_pdf": (c_int, [P, P, P, c_uint64, c_uint64, c_int, c_int, c_int, P, P, P]),
    "benerf_pixel_rays": (c_int, [P, c_int, P, P, c_int64, c_float, c_float, c_float, c_float, P, P, P]),
    "benerf_ndc_rays": (c_int, [c_int, c_int, c_float, c_float, P, P, c_int64, P, P, P]),
    "benerf_posenc": (c_int, [P, c_int64, c_int, c_int, c_int, P, P]),
    "benerf_mse_fwd": (c_int, [P, P, c_int64, P, P]),
    "benerf_mse_bwd": (c_int, [P, P, c_int64, P, P, P, P]),
    "benerf_bright_log_fwd": (c_int, [P, c_int64, c_int, P, P]),
    "benerf_bright_log_bwd": (c_int, [P, P, c_int64, c_int, P, P]),
    "benerf_rgb2gray_fwd": (c_int, [P, c_int64, P, P]),
    "benerf_rgb2gray_bwd": (c_int, [P, c_int64, P, P]),
    "benerf_loss_stats": (c_int, [POINTER(LossCfg), P, P, P, P, P, P, P, P]),
    "benerf_loss_grads": (c_int, [POINTER(LossCfg), P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "benerf_event_accumulate": (c_int, [P, P, P, c_int64, c_int, c_int, P, P]),
    "benerf_event_window_accumulate": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_int, c_int, P, P]),
    "benerf_gather_rows": (c_int, [P, P, c_int64, c_int, P, P]),
    "benerf_sample_pixels": (c_int, [c_int64, c_int64, ctypes.c_uint64, ctypes.c_uint64, P, P]),
    "benerf_adam_step": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, c_int, c_double, P, P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


class BenerfHipError(RuntimeError):
    pass


def load():
    """dlopen libbenerf_hip.so (built in-tree by benerf_amd/build.py).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BenerfHipError(
            "libbenerf_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C benerf_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  - first: the library must share the HIP runtime torch ships (loading /opt/rocm's copy
    #                              before torch's leaves the process with two runtimes and no visible device)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.benerf_version() != ABI_VERSION:       # a stale in-tree build, or a library of another revision on the path
        raise BenerfHipError("libbenerf_hip.so at %s reports ABI revision %d, this binding is written against %d (include/benerf_hip.h: "
                             "BENERF_ABI_VERSION) - rebuild with `make -C benerf_amd/csrc`" % (LIB_PATH, lib.benerf_version(), ABI_VERSION))
    _lib = lib
    return lib


class BenerfRangeError(BenerfHipError):
    """BENERF_ERANGE: a value left the range of the split-f16 MLP mode (include/benerf_hip.h, K3 `status`)."""


ERANGE = -4


def check(rc, what):
    if rc != 0:
        msg = load().benerf_last_error()
        cls = BenerfRangeError if rc == ERANGE else BenerfHipError
        raise cls("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))

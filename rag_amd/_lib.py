"""ctypes binding of librag_amd.so (C ABI declared in include/rag_amd.h)."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LOCK = threading.Lock()

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
c_int32_p = ctypes.POINTER(ctypes.c_int32)

class TailSpec(ctypes.Structure):
    """ragmi_tail_t of include/rag_amd.h: a consumer 1x1x1 ConvBR_3d fused into a 3x3x3 kernel's epilogue."""
    _fields_ = [("weight", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("relu", ctypes.c_int32),
                ("y", c_void_p), ("y_bstride", c_int64), ("y_ch0", ctypes.c_int32), ("cout", ctypes.c_int32)]


c_tail_p = ctypes.POINTER(TailSpec)


class K1RSpec(ctypes.Structure):
    """ragmi_k1r_t of include/rag_amd.h: one resample + 1x1x1 ConvBR_3d of a paired launch."""
    _fields_ = [("x", c_void_p), ("x_bstride", c_int64), ("Di", ctypes.c_int32), ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32),
                ("weight", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("relu", ctypes.c_int32),
                ("y_ch0", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32)]


c_k1r_p = ctypes.POINTER(K1RSpec)


class Cell2dIn(ctypes.Structure):
    """ragmi_cell2d_in_t of include/rag_amd.h: one input of a fused Cell_2d launch (tensor + its 1x1 ConvBR_2d)."""
    _fields_ = [("x", c_void_p), ("x_bstride", c_int64), ("Cin", ctypes.c_int32), ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32),
                ("weight", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("relu", ctypes.c_int32)]


c_cell2d_p = ctypes.POINTER(Cell2dIn)

# name -> (restype, argtypes); mirrors include/rag_amd.h one-to-one
SIGNATURES = {
    "ragmi_version": (c_int, []),
    "ragmi_last_error": (ctypes.c_char_p, []),
    "ragmi_graph_node_census": (c_int, [c_void_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "ragmi_costvol_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_costvol_stem_weights_elems": (c_int64, [c_int, c_int]),
    "ragmi_costvol_stem_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ragmi_costvol_stem_workspace_elems": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ragmi_costvol_stem_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_tail_p, c_int, c_void_p]),
    "ragmi_costvol_stem_conv3d_supported": (c_int, [c_int] * 9),
    "ragmi_costvol_stem_conv3d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_tail_p,
                                              c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int32_p, c_int, c_int, c_tail_p,
                                              c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_uses_x3": (c_int, [c_int] * 10),
    "ragmi_conv3d_k3_g4_caps": (c_int, [c_int] * 10),
    "ragmi_conv3d_k3_packed_elems": (c_int64, [c_int, c_int]),
    "ragmi_conv3d_k3_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_int64, c_int32_p, c_void_p, c_int64, c_int32_p,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_small_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int,
                                          c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_small_fwd_ex": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int,
                                             c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_dual_fwd": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int32_p,
                                         c_void_p, c_int64, c_int32_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_fwd_ex": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int64, c_int32_p, c_void_p, c_int64, c_int32_p,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_tail_p, c_int, c_void_p]),
    "ragmi_conv3d_k3_dual_fwd_ex": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                            c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int32_p,
                                            c_void_p, c_int64, c_int32_p, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_tail_p, c_int, c_void_p]),
    "ragmi_conv3d_k3_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int32_p, c_int32_p, c_int32_p, c_int]),
    "ragmi_conv3d_k1_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "ragmi_conv3d_k1_chain_supported": (c_int, [c_int] * 3),
    "ragmi_conv3d_k1_chain_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "ragmi_conv3d_k1_fwd_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "ragmi_conv3d_k1_resample_fwd": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                             c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k1_resample_pair_fwd": (c_int, [c_k1r_p, c_k1r_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k1_resample_multi_fwd": (c_int, [ctypes.POINTER(c_k1r_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_int, c_int,
                                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_down2_tail_supported": (c_int, [c_int, c_int, c_int]),
    "ragmi_cell2d_supported": (c_int, [c_int] * 7),
    "ragmi_cell2d_fwd": (c_int, [c_cell2d_p, c_cell2d_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int64, c_int32_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_upconv3d_c1_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "ragmi_upconv3d_c1_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_trilinear3d_act_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_trilinear3d_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    "ragmi_conv2d_k3_strided_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_void_p]),
    "ragmi_add_fwd": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
                              c_int, c_int, c_int64, c_int, c_void_p]),
    "ragmi_bn_workspace_elems": (c_int64, [c_int, c_int, c_int64]),
    "ragmi_bn_train_stats_fwd": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ragmi_conv3d_k3_pack_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k3_pack_for": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_sgd_workspace_bytes": (c_int64, []),
    "ragmi_sgd_clip_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                                    c_void_p]),
    "ragmi_bn_train_act_fwd": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                       c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                       c_void_p, c_int64, c_int, c_void_p]),
    "ragmi_bn_act_bwd": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "ragmi_bn_act_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
                                 c_int, c_int, c_int64, c_void_p]),
    "ragmi_bn_act_bwd_coeffs": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ragmi_bn_act_bwd_apply": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p]),
    "ragmi_conv3d_k3_wgrad_workspace_elems": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ragmi_conv3d_k3_wgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_void_p,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv3d_k1_wgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "ragmi_trilinear3d_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_costvol_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv2d_k3_strided_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_conv2d_k3_strided_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_disparity_regression_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_stereo_metrics_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ragmi_masked_smooth_l1_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "ragmi_disp_softargmin_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ragmi_disp_softargmin_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p]),
    "ragmi_disparity_regression_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
}


def lib_path() -> str:
    return os.environ.get("RAG_AMD_LIB", os.path.join(_HERE, "lib", "librag_amd.so"))


def load_library():
    """Load librag_amd.so and bind every exported entry point.  Raises if the HIP library
    has not been built (there is deliberately no fallback path)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is None:
            path = lib_path()
            if not os.path.exists(path):
                raise RuntimeError(
                    f"rag_amd: HIP library not found at {path}. Build it with `make` at the repo root "
                    "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
            lib = ctypes.CDLL(path)
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
                fn.restype = restype
                fn.argtypes = argtypes
            _LIB = lib
    return _LIB


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().ragmi_last_error()
        raise RuntimeError(f"rag_amd.{what} failed with status {status}: {msg.decode() if msg else ''}")

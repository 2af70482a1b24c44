"""ctypes binding of libdiffsbdd_hip.so (include/diffsbdd_hip.h).

The library is built in-tree by `diffsbdd_amd.build.build()` (hipcc,
--offload-arch=gfx950).  There is NO fallback: if the shared object is missing
or does not export the expected ABI, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSBDD_LIB: load another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("DSBDD_LIB") or os.path.join(_HERE, "libdiffsbdd_hip.so")
ABI_VERSION = 6

# error / status codes (include/diffsbdd_hip.h)
OK, ERR_ARG, ERR_STATE, ERR_CAPACITY, ERR_LAUNCH = 0, -1, -2, -3, -4
STATUS_NAN, STATUS_EDGE_OVERFLOW = 1, 2

# weight-slot enums
G_NAMES = ["ATOM_ENC_W0T", "ATOM_ENC_B0", "ATOM_ENC_W1T", "ATOM_ENC_B1",
           "RES_ENC_W0T", "RES_ENC_B0", "RES_ENC_W1T", "RES_ENC_B1",
           "ATOM_DEC_W0T", "ATOM_DEC_B0", "ATOM_DEC_W1T", "ATOM_DEC_B1",
           "RES_DEC_W0T", "RES_DEC_B0", "RES_DEC_W1T", "RES_DEC_B1",
           "EMB_WT", "EMB_B", "EMBOUT_WT", "EMBOUT_B"]
GCL_NAMES = ["E1_WT", "E1_WD", "E1_WD0", "E1_TAB", "E2_WT", "E2_B", "ATT_W", "ATT_B",
             "N1_WT", "N1_B", "N2_WT", "N2_B"]
EQ_NAMES = ["C1_WT", "C_WD", "C_WD0", "C_TAB", "C_W2T", "C_B2",
            "X_WD", "X_WD0", "X_TAB", "X_W2T", "X_B2", "W3"]
OPT_PRUNE, OPT_CONE, OPT_GRANULE16, OPT_EMU, OPT_SPLITK = 0, 1, 2, 3, 4
(BUF_EDGE_ROW, BUF_EDGE_COL, BUF_EDGE_D0, BUF_ROW_PTR, BUF_H, BUF_X, BUF_NODE_BATCH, BUF_DEG,
 BUF_LEVEL, BUF_LEVEL_LIST, BUF_LEVEL_COUNT, BUF_LEVEL_END, BUF_LROW_PTR, BUF_LEDGE_ROW, BUF_LEDGE_COL,
 BUF_LEDGE_D0, BUF_LEVEL_STATS) = range(17)


class Config(C.Structure):
    """struct dsbdd_config"""
    _fields_ = [(n, C.c_int32) for n in (
        "atom_nf", "residue_nf", "joint_nf", "hidden_nf", "n_layers", "inv_sublayers",
        "attention", "use_tanh", "update_pocket_coords", "reflection_equivariant",
        "edge_embedding_dim", "has_cutoff_ligand", "has_cutoff_pocket",
        "has_cutoff_interaction")] + [(n, C.c_float) for n in (
            "cutoff_ligand", "cutoff_pocket", "cutoff_interaction",
            "norm_constant", "normalization_factor", "coords_range")]


class TrainGraph(C.Structure):
    """struct dsbdd_train_graph"""
    _fields_ = [(n, C.c_void_p) for n in ("erow", "ecol", "ed0", "row_ptr", "deg", "rev", "node_batch", "lig_off",
                                          "poc_off")] + [(n, C.c_int64) for n in ("n_lig", "n_nodes", "n_edges", "batch")]


class LossCfg(C.Structure):
    """struct dsbdd_loss_cfg"""
    _fields_ = [(n, C.c_int32) for n in ("batch", "n_lig", "n_pocket", "atom_nf", "residue_nf", "timesteps", "remove_com",
                                         "vnode_idx")] + \
               [(n, C.c_float) for n in ("norm_value_x", "norm_value_h", "norm_bias_h")] + \
               [(n, C.c_int32) for n in ("n1_tab", "n2_tab")]


class TrainMlp(C.Structure):
    """struct dsbdd_train_mlp"""
    _fields_ = [("P", C.c_void_p), ("Q", C.c_void_p), ("ldpq", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("wd", "wd0", "tab", "W2", "W2T", "b2", "head", "head_b")]


class TrainMlpGrad(C.Structure):
    """struct dsbdd_train_mlp_grad"""
    _fields_ = [("dP", C.c_void_p), ("dQ", C.c_void_p), ("ldo", C.c_int32), ("d_vec", C.c_void_p), ("d_W2", C.c_void_p),
                ("gd0", C.c_void_p)]


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float

# name -> (restype, argtypes); every symbol include/diffsbdd_hip.h declares
SIGNATURES = {
    "dsbdd_abi_version": (C.c_int, []),
    "dsbdd_last_error": (C.c_char_p, []),
    "dsbdd_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "dsbdd_engine_destroy": (None, [_P]),
    "dsbdd_engine_weight_slots": (C.c_int, [_P]),
    "dsbdd_engine_set_weights": (C.c_int, [_P, C.POINTER(_P), C.c_int]),
    "dsbdd_engine_workspace_bytes": (C.c_size_t, [_P, _I64, _I64, _I64, _I64]),
    "dsbdd_engine_bind_workspace": (C.c_int, [_P, _P, C.c_size_t, _I64, _I64, _I64, _I64]),
    "dsbdd_engine_set_trace": (C.c_int, [_P, _P, _P]),
    "dsbdd_engine_set_pocket_frame": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64]),
    "dsbdd_engine_clear_pocket_frame": (C.c_int, [_P]),
    "dsbdd_dynamics_forward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64,
                                         _P, _P, _I64, _P, _P, _P]),
    "dsbdd_engine_buffer": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "dsbdd_engine_graph_stats": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "dsbdd_engine_last_plan": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dsbdd_engine_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "dsbdd_engine_get_option": (C.c_int, [_P, C.c_int]),
    "dsbdd_engine_profile": (C.c_int, [_P, C.c_int, C.c_int]),
    "dsbdd_engine_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "dsbdd_cond_reverse_update": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64,
                                            _I32, _I32, _F, _F, _F, _I32]),
    "dsbdd_joint_reverse_update": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64,
                                             _I32, _I32, _F, _F, _F, _I32]),
    "dsbdd_segment_mean3": (C.c_int, [_P, _P, _I32, _P, _I64, _I64, _P]),
    "dsbdd_cond_affine_noise": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _F, _F, _I32]),
    "dsbdd_joint_affine_noise": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _F, _F,
                                           _I32, _I32]),
    "dsbdd_cond_repaint_update": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64,
                                            _I32, _I32, _F, _F, _F, _F, _I32, _I32]),
    "dsbdd_joint_repaint_update": (C.c_int, [_P] * 15 + [_I64, _I64, _I64, _I32, _I32, _F, _F, _F, _F, _I32]),
    "dsbdd_cond_step_keyed": (C.c_int, [_P] * 10 + [_I64, _I64, _I64, _I32, _I32, _F, _F, _F, _I32, _F, _F, _F, _I32,
                                        C.c_uint64, C.c_uint64, _I64, _P, _P, _F]),
    "dsbdd_randn_keyed": (C.c_int, [_P, _P, _P, _I64, _I32, _I64, _I64, _P, C.c_uint64, C.c_uint64,
                                    C.c_uint32]),
    "dsbdd_node_linear": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _I32, _P, _P, _I32,
                                    _P, _I32, _I64, _I32, _I32]),
    "dsbdd_build_edges": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, C.POINTER(Config),
                                    _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "dsbdd_bond_orders": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P, C.c_float, C.c_float,
                                    C.c_float, _I32, _P]),
    # training-step building blocks (struct arguments are passed by pointer: ctypes.byref / arrays)
    "dsbdd_train_scratch_bytes": (C.c_size_t, [_I32, _I64, _I64]),
    "dsbdd_train_wgrad_scratch_bytes": (C.c_size_t, [_I64, _I64, _I64]),
    "dsbdd_train_wgrad_plan_bytes": (C.c_size_t, [_I64, _I64, _I64]),
    "dsbdd_train_edge_rev": (C.c_int, [_P, _P, _P]),
    "dsbdd_train_sample_mean": (C.c_int, [_P, _P, _P, _P]),
    "dsbdd_train_gcl_forward": (C.c_int, [_P, _I32, _P, _P, _P, _F, _P, _P, C.c_size_t]),
    "dsbdd_train_gcl_backward": (C.c_int, [_P, _I32, _P, _P, _P, _F, _P, _P, _P, _P, C.c_size_t]),
    "dsbdd_train_coord_forward": (C.c_int, [_P, _I32, _P, _P, _I32, _P, _P, _I64, _F, _F, _I32, _F, _P, _P,
                                            C.c_size_t]),
    "dsbdd_train_coord_backward": (C.c_int, [_P, _I32, _P, _P, _I32, _P, _P, _I64, _I64, _F, _F, _I32, _F, _P, _P, _P,
                                             _P, _P, C.c_size_t]),
    "dsbdd_train_radial_backward": (C.c_int, [_P, _P, _P, _P, _P]),
    "dsbdd_train_wgrad": (C.c_int, [_P, _P, _I32, _P, _I32, _I64, _I32, _I32, _P, _P, C.c_size_t]),
    "dsbdd_train_colsum": (C.c_int, [_P, _P, _I32, _I64, _I32, _P, _P, C.c_size_t]),
    "dsbdd_train_net_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "dsbdd_train_net_destroy": (None, [_P]),
    "dsbdd_train_net_param_count": (C.c_int, [_P]),
    "dsbdd_train_net_pack_bytes": (C.c_size_t, [_P]),
    "dsbdd_train_net_workspace_bytes": (C.c_size_t, [_P, C.POINTER(TrainGraph)]),
    "dsbdd_train_net_forward": (C.c_int, [_P, _P, C.POINTER(TrainGraph), C.POINTER(_P), _P, C.c_size_t, _P, C.c_size_t, _P, _P,
                                          _P, _I64, _I32, _P, _P, _P]),
    "dsbdd_train_net_backward": (C.c_int, [_P, _P, C.POINTER(TrainGraph), C.POINTER(_P), C.POINTER(_P), _P, C.c_size_t, _P,
                                           C.c_size_t, _I64, _P, _P, _P, _P]),
    # the loss terms of the pocket-conditioned training step around the network call (csrc/loss_head.h)
    "dsbdd_edge_capacity": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _P]),
    "dsbdd_loss_rows": (C.c_int, []),
    "dsbdd_loss_out_rows": (C.c_int, []),
    "dsbdd_loss_cond_pre": (C.c_int, [_P, C.POINTER(LossCfg)] + [_P] * 17),
    "dsbdd_loss_cond_post": (C.c_int, [_P, C.POINTER(LossCfg)] + [_P] * 8),
    "dsbdd_loss_cond_post_backward": (C.c_int, [_P, C.POINTER(LossCfg)] + [_P] * 9),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load libdiffsbdd_hip.so and bind every declared symbol.  Raises
    HipLibraryError if the library is missing or incompatible."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own copy of the HIP runtime (libamdhip64); loading our library before it
    # would pull in /opt/rocm's copy as well, and kernels launched through one runtime do not see the
    # device the other one initialised ("no ROCm-capable device is detected")
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the product path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    ver = lib.dsbdd_abi_version()
    if ver != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {ver}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc == OK:
        return
    msg = load().dsbdd_last_error()
    msg = msg.decode() if msg else ""
    raise HipLibraryError(f"{what} failed with code {rc}: {msg}")

"""ctypes binding of libdpf_hip.so (include/dpf_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be
resolved, every product entry point raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or
`make -C dpf_nets_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_l = ctypes.c_long

# name -> (restype, argtypes); must list every symbol include/dpf_hip.h declares
SIGNATURES = {
    "dpf_nndistance": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpf_nndistance_auto": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpf_nndistance_strided": (_i, [_i, _i, _vp, ctypes.c_long, _i, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp]),
    "dpf_nndistance_strided_auto": (_i, [_i, _i, _vp, ctypes.c_long, _i, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp]),
    "dpf_nndistance_mfma_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpf_nndistance_mfma": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpf_nndistance_cd_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpf_nndistance_cd": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "dpf_pairwise_cd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dpf_pairwise_cd": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpf_nndistancegrad": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpf_approxmatch": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dpf_approxmatch_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpf_emd_set_matrix_path": (_i, [_i]),
    "dpf_approxmatch_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpf_approxmatch_cost_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpf_matchcost": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dpf_matchcostgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpf_matchcostgrad_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dpf_matchcostgrad": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpf_flow_canon_floats": (_sz, [_i]),
    "dpf_flow_packed_bytes": (_sz, [_i, _i, _i]),
    "dpf_flow_film_floats": (_sz, [_i, _i]),
    "dpf_flow_pack": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dpf_flow_film": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _f, _vp]),
    "dpf_nn_small_mode": (_i, [_i]),
    "dpf_flow_set_tile16": (_i, [_i]),
    "dpf_flow_tile16_launches": (ctypes.c_long, []),
    "dpf_flow_forward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "dpf_flow_forward_base": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _f, _vp]),
    "dpf_flow_train_canon_floats": (_sz, []),
    "dpf_flow_train_packed_bytes": (_sz, [_i, _i]),
    "dpf_flow_train_stats_floats": (_sz, []),
    "dpf_flow_train_film_floats": (_sz, [_i]),
    "dpf_flow_train_workspace_bytes": (_sz, [_i, _i]),
    "dpf_flow_train_pack": (_i, [_i, _i, _vp, _vp, _vp]),
    "dpf_flow_train_forward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "dpf_flow_train_backward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _f, _vp, _vp]),
    "dpf_flow_train_backward_lists": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _f, _vp, _vp]),
    "dpf_fscore_reduce": (_i, [_i, _i, _i, _vp, _vp, _f, _vp, _vp]),
    "dpf_chamfer_reduce": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dpf_encoder_canon_floats": (_sz, []),
    "dpf_encoder_packed_bytes": (_sz, [_i]),
    "dpf_encoder_pack": (_i, [_i, _vp, _vp, _vp]),
    "dpf_encoder_forward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dpf_debug_nn_surrogate": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp]),
    "dpf_debug_emd_exponents": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dpf_train_graph_replays": (_l, []),
    "dpf_train_colsum_fallbacks": (_l, []),
    "dpf_train_graph_stats": (None, [_vp]),
    "dpf_train_graph_set_enabled": (_i, [_i]),
    "dpf_train_kernel_times": (_i, [_i, _vp, _vp]),
    "dpf_encoder_train_workspace_bytes": (_sz, [_i, _i]),
    "dpf_encoder_train_forward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "dpf_encoder_train_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpf_gprior_canon_floats": (_sz, [_i, _i]),
    "dpf_gprior_packed_floats": (_sz, [_i, _i, _i]),
    "dpf_gprior_pack": (_i, [_i, _i, _i, _f, _vp, _vp, _vp]),
    "dpf_gprior_forward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "dpf_gprior_train_workspace_floats": (_sz, [_i, _i, _i]),
    "dpf_gprior_train_forward": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp]),
    "dpf_gprior_train_backward": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _f, _f, _vp]),
    "dpf_adam_step": (_i, [_sz, _vp, _vp, _vp, _vp, _vp] + [ctypes.c_double] * 7 + [_vp]),
    "dpf_flow_train_update_running": (_i, [_i, ctypes.c_double] + [_vp] * 7),
    "dpf_film_train_max_batch": (_i, []),
    "dpf_film_train_forward": (_i, [_i, _i, _i] + [_vp] * 6 + [_f] + [_vp] * 5 + [_vp]),
    "dpf_film_train_backward": (_i, [_i, _i, _i] + [_vp] * 14 + [_i, _vp]),
    "dpf_pointflow_nll_workspace_floats": (_sz, []),
    "dpf_pointflow_nll": (_i, [_i, _i, _i, _vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _vp]),
    "dpf_pointflow_nll_backward": (_i, [_i, _i, _i, _vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dpf_version": (ctypes.c_char_p, []),
}

PREC = {"bf16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 4}
MODE = {"direct": 0, "inverse": 1}


def lib_path():
    return os.path.join(_HERE, "libdpf_hip.so")


def have_lib():
    return os.path.exists(lib_path())


def lib():
    """The loaded library; raises RuntimeError (never falls back) if it is absent."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "dpf_nets_amd: %s is missing -- the HIP kernels are not built. "
                "Run __graft_entry__.build() or `make -C dpf_nets_amd/csrc`. There is no CPU fallback." % path)
        # PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, same soname as
        # /opt/rocm's).  It must be the one resident when libdpf_hip.so is loaded, or the process ends up with
        # two runtimes and every launch on torch's streams fails with hipErrorNoDevice: import torch first.
        import torch  # noqa: F401
        handle = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape"}.get(rc, "hipError_t %d" % rc)
        raise RuntimeError("dpf_hip: %s failed: %s" % (what, kind))


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream

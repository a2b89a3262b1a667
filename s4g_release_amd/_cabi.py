"""ctypes binding of ``libs4g_hip.so`` (the C ABI declared in ``include/s4g_ops.h``).

There is deliberately NO fallback: if the HIP library is missing or does not
export a declared symbol, importing the operators fails loudly.  The oracle
under ``oracle/`` is test infrastructure and is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# S4G_HIP_LIB: another build of the SAME sources (measurement builds with parts compiled out,
# tools/ablate_kernels.sh); never a different implementation -- the ABI version and every
# declared symbol are checked either way
LIB_PATH = os.environ.get("S4G_HIP_LIB") or os.path.join(_HERE, "libs4g_hip.so")

S4G_ABI_VERSION = 12
S4G_EINVAL = -1
S4G_EWORKSPACE = -2
S4G_EUNSUPPORTED = -3
S4G_FLAG_FMAD = 1
S4G_OP_FPS, S4G_OP_BALL_QUERY, S4G_OP_THREE_NN = 1, 2, 3

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_int = ctypes.c_int
_f32 = ctypes.c_float

_i32 = ctypes.c_int32


class GemmDesc(ctypes.Structure):
    """ctypes mirror of `s4g_gemm_desc_t` (include/s4g_ops.h)."""
    _fields_ = [
        ("loader", _i32), ("epilogue", _i32), ("groups", _i32), ("relu", _i32),
        ("P", _i32), ("Cin", _i32), ("Kpad", _i32), ("Cout", _i32),
        ("W", _vp), ("bias", _vp), ("w_gstride", _i32), ("b_gstride", _i32),
        ("A", _vp), ("lda", _i32), ("a_coff", _i32), ("a_gcol", _i32),
        ("gidx", _vp), ("feat", _vp), ("xyz", _vp), ("ctr", _vp),
        ("Cf", _i32), ("N", _i32), ("M", _i32), ("K", _i32),
        ("nidx", _vp), ("nw", _vp), ("sparse", _vp), ("dense", _vp),
        ("C2", _i32), ("C1", _i32), ("N2", _i32), ("N1", _i32),
        ("out", _vp), ("ldc", _i32), ("c_coff", _i32), ("c_gcol", _i32),
        ("cf_ptr", _vp * 4), ("cf_start", _i32 * 5), ("cf_sigmoid_from", _i32), ("cf_N", _i32),
        ("precision", _i32), ("Kpad16", _i32), ("W_bf16x3", _vp), ("mlp1_w", _vp),
        ("W_f16x2", _vp), ("w_inv_scale", _vp), ("a_amax", _vp), ("a_amax2", _vp),
        ("a_amax_floor", _f32), ("out_amax", _vp), ("W_f16x2_frag", _vp),
        ("W2_f16x2_frag", _vp), ("w2_inv_scale", _vp), ("bias2", _vp), ("Cout2", _i32), ("relu2", _i32),
        ("W3_f16x2_frag", _vp), ("w3_inv_scale", _vp), ("bias3", _vp), ("Cout3", _i32), ("relu3", _i32),
        ("loader_bias", _vp), ("rows_per_scene", _i32), ("rel_xyz4", _vp), ("seg4", _vp), ("seg_rows", _vp),
        ("out2", _vp), ("ldc2", _i32), ("split_n", _i32), ("out_amax2", _vp),
    ]


class HeadsDesc(ctypes.Structure):
    """ctypes mirror of `s4g_heads_desc_t` (include/s4g_ops.h)."""
    _fields_ = [
        ("precision", _i32), ("P", _i32), ("N", _i32), ("ldx", _i32),
        ("C", _i32), ("H0", _i32), ("H1", _i32), ("H2", _i32), ("H3", _i32),
        ("X", _vp), ("W_frag", _vp * 5), ("bias", _vp * 5), ("w_inv_scale", _vp * 5),
        ("out", _vp * 4), ("channels", _i32 * 4), ("sigmoid_head", _i32),
        ("a_amax", _vp), ("a_amax_floor", _f32), ("rows_per_scene", _i32),
        ("pre_W_frag", _vp * 2), ("pre_bias", _vp * 2), ("pre_w_inv_scale", _vp * 2),
        ("pre_nidx", _vp), ("pre_nw", _vp), ("pre_sparse", _vp), ("pre_dense", _vp),
        ("pre_lbias", _vp), ("pre_a_amax2", _vp), ("pre_N2", _i32), ("out_batch_stride", _i64), ("head_mask", _i32),
    ]


# name -> (restype, argtypes); mirrors include/s4g_ops.h one to one.
SIGNATURES = {
    "s4g_mlp_gemm_f32": (_int, [ctypes.POINTER(GemmDesc), _vp]),
    "s4g_heads_chain_f32": (_int, [ctypes.POINTER(HeadsDesc), _vp]),
    "s4g_ball_query_i32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _sz,
                                  _int, _vp]),
    "s4g_three_nn_weights_i32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _sz,
                                        _int, _vp]),
    "s4g_three_nn_grid_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "s4g_three_nn_grid_header_offset": (_sz, [_i64, _i64]),
    "s4g_three_nn_weights_grid_i32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _f32, _vp, _vp, _vp,
                                             _sz, _int, _vp]),
    "s4g_three_nn_grid_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _sz, _int, _vp]),
    "s4g_group_points_xyz_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "s4g_gemm_chain_supported": (_int, [_int, _int, _int, _int]),
    "s4g_interp_add_cl_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "s4g_group_points_ws_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "s4g_three_interpolate_ws_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _int, _vp]),
    "s4g_fps_gather_i32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _int, _vp]),
    "s4g_fps_prepass_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "s4g_fps_gather_ex_i32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp]),
    "s4g_fps_prefix_check_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _int, _vp]),
    "s4g_build_variants": (_int, []),
    "s4g_test_knobs_enabled": (_int, []),
    # double dispatch of the five operators (csrc/ops_f64.hip)
    "s4g_fps_f64": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _int, _vp]),
    "s4g_ball_query_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _int, _vp]),
    "s4g_three_nn_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp]),
    "s4g_group_points_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_group_points_backward_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_three_interpolate_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "s4g_three_interpolate_backward_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_scatter_det_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "s4g_scatter_det_workspace_bytes_c": (_sz, [_i64, _i64, _i64, _i64, _int]),
    "s4g_group_points_backward_det_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "s4g_three_interpolate_backward_det_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "s4g_group_rel_xyz_i32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_group_rel_xyz_unique_i32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "s4g_expected_score_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "s4g_decode_poses_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "s4g_collision_counts_f32": (_int, [_vp, _vp, _i64, _i64, _i64, ctypes.POINTER(ctypes.c_float),
                                        _vp, _vp]),
    "s4g_collision_counts_n_f32": (_int, [_vp, _vp, _i64, _i64, _i64, ctypes.POINTER(ctypes.c_float),
                                          _vp, _int, _vp, _vp]),
    "s4g_sort_pairs_workspace_bytes": (_sz, [_i64]),
    "s4g_sort_pairs_u32": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "s4g_exclusive_scan_workspace_bytes": (_sz, [_i64]),
    "s4g_exclusive_scan_i32": (_int, [_vp, _vp, _i64, _vp, _sz, _vp]),
    "s4g_query_group_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _sz,
                                   _int, _vp]),
    "s4g_crop_indices_f32": (_int, [_vp, _i64, ctypes.POINTER(ctypes.c_float), _vp, _vp, _vp]),
    "s4g_voxel_down_sample_workspace_bytes": (_sz, [_i64]),
    "s4g_voxel_down_sample_f32": (_int, [_vp, _i64, _f32, ctypes.POINTER(ctypes.c_float),
                                         ctypes.POINTER(ctypes.c_int32), _vp, _vp, _vp, _sz, _vp]),
    "s4g_radius_outlier_workspace_bytes": (_sz, [_i64]),
    "s4g_radius_outlier_mask_f32": (_int, [_vp, _i64, _f32, _i32, _vp, _vp, _sz, _int, _vp]),
    "s4g_abi_version": (_int, []),
    "s4g_error_string": (ctypes.c_char_p, [_int]),
    "s4g_workspace_bytes": (_sz, [_int, _i64, _i64, _i64, _i64]),
    "s4g_fps_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _int, _vp]),
    "s4g_ball_query_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _sz,
                                  _int, _vp]),
    "s4g_group_points_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_group_points_backward_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_gather_points_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "s4g_three_nn_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _int, _vp]),
    "s4g_three_interpolate_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "s4g_three_interpolate_backward_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp,
                                                  _vp]),
    "s4g_interp_weights_f32": (_int, [_vp, _i64, _i64, _f32, _vp, _vp]),
}

_lib = None


def lib():
    """Load the HIP library once; raise if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libs4g_hip.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C s4g_release_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7 and the system ROCm has one with the
    # same SONAME: whichever is loaded first serves the whole process.  Import
    # torch first so the kernels and torch's allocator/streams share ONE runtime
    # (loading ours first gives "no ROCm-capable device" on the first launch).
    import torch  # noqa: F401
    if os.environ.get("S4G_HIP_LIB"):
        # a measurement build (possibly with parts compiled out) is about to serve every operator of
        # this process: say so, loudly -- a leftover variable must never go unnoticed
        import warnings
        warnings.warn("S4G_HIP_LIB overrides the shipped library: loading %s" % LIB_PATH, RuntimeWarning,
                      stacklevel=2)
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise RuntimeError("libs4g_hip.so does not export %s" % name)
        fn.restype = res
        fn.argtypes = args
    ver = L.s4g_abi_version()
    if ver != S4G_ABI_VERSION:
        raise RuntimeError("libs4g_hip.so ABI version %d != expected %d" % (ver, S4G_ABI_VERSION))
    _lib = L
    return _lib


def knob(name, default=None):
    """An A/B / test knob of include/s4g_ops.h's second list: read from the environment ONLY when the process also sets
    S4G_TEST_KNOBS=1 (tests/conftest.py does), else `default` -- the host-side twin of csrc/s4g_common.h's s4g::knob.
    A production process cannot have its launch plan changed by a stray variable."""
    if os.environ.get("S4G_TEST_KNOBS") != "1":
        return default
    return os.environ.get(name, default)


def check(code, what):
    if code != 0:
        msg = lib().s4g_error_string(code)
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))

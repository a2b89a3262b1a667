"""ctypes/numpy front-end of the CPU oracle (``oracle/s4g_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see the header of ``s4g_oracle.c``.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module, and only as the checker / reported CPU baseline.  The
product package (``s4g_release_amd``) never imports it.

Every function takes and returns numpy arrays in the reference's Python-level
layouts (``(B,3,N)`` fp32 -- or, like the reference's extension, fp64 -- clouds, int64 indices) and mirrors one reference
operator (PN2U = inference/grasp_proposal/network_models/models/pointnet2_utils):

  fps / fps_literal     PN2U/csrc/sampling_kernel.cu:49-172
  ball_query            PN2U/csrc/ball_query_kernel.cu:33-133
  group_points(+bwd)    PN2U/csrc/grouping_kernel.cu:32-152
  gather_points         PN2U/functions.py:10-25
  three_nn              PN2U/csrc/interpolate_kernel.cu:32-132
  three_interpolate(+bwd) PN2U/csrc/interpolate_kernel.cu:138-341
  interp_weights        PN2U/modules.py:118-120
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# S4G_ORACLE_LIB (tests/test_oracle_sanitizers.py only): another build of the same source, e.g. the
# AddressSanitizer / UBSan library `make -C oracle asan` produces
_LIB_PATH = os.environ.get("S4G_ORACLE_LIB") or os.path.join(_HERE, "libs4g_oracle.so")
_lib = None
_double = False      # see double_dispatch

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i64 = ctypes.c_int64


def build(force=False):
    """Compile the C restatement with gcc (no GPU needed)."""
    src = os.path.join(_HERE, "s4g_oracle.c")
    if os.environ.get("S4G_ORACLE_LIB"):
        return _LIB_PATH            # the caller built it
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libs4g_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        L = _lib
        L.s4g_oracle_set_threads.argtypes = [ctypes.c_int]
        L.s4g_oracle_set_threads.restype = ctypes.c_int
        for name in ("s4g_oracle_fps", "s4g_oracle_fps_literal"):
            getattr(L, name).argtypes = [_f32p, _i64, _i64, _i64, _i64p, ctypes.c_int]
        L.s4g_oracle_ball_query.argtypes = [_f32p, _f32p, _i64, _i64, _i64, ctypes.c_float,
                                            _i64, _i64p, _i64p, ctypes.c_int]
        L.s4g_oracle_group_points.argtypes = [_f32p, _i64p, _i64, _i64, _i64, _i64, _i64, _f32p]
        L.s4g_oracle_group_points_backward.argtypes = [_f32p, _i64p, _i64, _i64, _i64, _i64, _i64, _f32p]
        L.s4g_oracle_gather_points.argtypes = [_f32p, _i64p, _i64, _i64, _i64, _i64, _f32p]
        L.s4g_oracle_three_nn.argtypes = [_f32p, _f32p, _i64, _i64, _i64, _i64p, _f32p, ctypes.c_int]
        L.s4g_oracle_three_interpolate.argtypes = [_f32p, _i64p, _f32p, _i64, _i64, _i64, _i64,
                                                   _f32p, ctypes.c_int]
        L.s4g_oracle_three_interpolate_backward.argtypes = [_f32p, _i64p, _f32p, _i64, _i64, _i64,
                                                            _i64, _f32p]
        L.s4g_oracle_interp_weights.argtypes = [_f32p, _i64, _i64, ctypes.c_float, _f32p]
        # scalar_t = double: the same source compiled with -DS4G_ORACLE_F64 (AT_DISPATCH_FLOATING_TYPES' other case)
        d, dbl = _f64p, ctypes.c_double
        for name in ("s4g_oracle_fps_f64", "s4g_oracle_fps_literal_f64"):
            getattr(L, name).argtypes = [d, _i64, _i64, _i64, _i64p, ctypes.c_int]
        L.s4g_oracle_ball_query_f64.argtypes = [d, d, _i64, _i64, _i64, dbl, _i64, _i64p, _i64p, ctypes.c_int]
        L.s4g_oracle_group_points_f64.argtypes = [d, _i64p, _i64, _i64, _i64, _i64, _i64, d]
        L.s4g_oracle_group_points_backward_f64.argtypes = [d, _i64p, _i64, _i64, _i64, _i64, _i64, d]
        L.s4g_oracle_gather_points_f64.argtypes = [d, _i64p, _i64, _i64, _i64, _i64, d]
        L.s4g_oracle_three_nn_f64.argtypes = [d, d, _i64, _i64, _i64, _i64p, d, ctypes.c_int]
        L.s4g_oracle_three_interpolate_f64.argtypes = [d, _i64p, d, _i64, _i64, _i64, _i64, d, ctypes.c_int]
        L.s4g_oracle_three_interpolate_backward_f64.argtypes = [d, _i64p, d, _i64, _i64, _i64, _i64, d]
        L.s4g_oracle_interp_weights_f64.argtypes = [d, _i64, _i64, dbl, d]
    return _lib


class double_dispatch:
    """Opt in to the operators' second dispatch case (`AT_DISPATCH_FLOATING_TYPES`: scalar_t = double):

        with oracle.double_dispatch():
            idx = oracle.fps(points_f64, 512)      # the -DS4G_ORACLE_F64 build, float64 in / out

    Outside the block every floating array is coerced to float32 -- numpy's default float64 (`np.random.rand`,
    `pts * 256 / 256`) must not silently select double-precision semantics in a checker.  Inside it, float64
    arrays keep their type and a call that mixes float32 and float64 arrays raises TypeError."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _double
        self.prev, _double = _double, self.on
        return self

    def __exit__(self, *exc):
        global _double
        _double = self.prev
        return False


def _f32(a, *more):
    """float32 -- or, inside `double_dispatch()`, float64 where the caller passes float64.  With several arrays:
    all coerced alike, mixed float32 / float64 refused."""
    arrs = [np.asarray(x) for x in (a,) + more]
    kinds = {x.dtype == np.float64 for x in arrs} if _double else {False}
    if len(kinds) > 1:
        raise TypeError("oracle: float32 and float64 arrays in one call (the reference's extension dispatches "
                        "on ONE scalar type)")
    dt = np.float64 if kinds == {True} else np.float32
    out = [np.ascontiguousarray(x, dtype=dt) for x in arrs]
    return out[0] if not more else out


def _fn(name, a):
    """The entry point for a's scalar type."""
    return getattr(lib(), name + ("_f64" if a.dtype == np.float64 else ""))


def _scalar(v, a):
    """A C `float` argument of the float build; the double build's `scalar_t` otherwise."""
    return ctypes.c_double(v) if a.dtype == np.float64 else ctypes.c_float(v)


def _i64a(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _fp(a):
    return a.ctypes.data_as(_f64p if a.dtype == np.float64 else _f32p)


def _ip(a):
    return a.ctypes.data_as(_i64p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (what, rc))


def fps(points, num_centroids, fmad=0, literal=False):
    points = _f32(points)
    B, C, N = points.shape
    if C != 3:
        raise RuntimeError("points.size(1) must be 3")
    idx = np.zeros((B, num_centroids), dtype=np.int64)
    fn = _fn("s4g_oracle_fps_literal" if literal else "s4g_oracle_fps", points)
    _check(fn(_fp(points), B, N, num_centroids, _ip(idx), fmad), "fps")
    return idx


def fps_literal(points, num_centroids, fmad=0):
    return fps(points, num_centroids, fmad, literal=True)


def ball_query(points, centroids, radius, num_neighbours, fmad=0):
    points, centroids = _f32(points, centroids)
    B, _, N = points.shape
    M = centroids.shape[2]
    K = int(num_neighbours)
    idx = np.zeros((B, M, K), dtype=np.int64)
    cnt = np.zeros((B, M), dtype=np.int64)
    # (the extension's `radius` is a C float whatever the tensors' type: ball_query.h; it is cast to scalar_t)
    _check(_fn("s4g_oracle_ball_query", points)(_fp(points), _fp(centroids), B, N, M,
                                                _scalar(float(np.float32(radius)), points), K, _ip(idx), _ip(cnt), fmad),
           "ball_query")
    return idx, cnt


def group_points(points, index):
    points, index = _f32(points), _i64a(index)
    B, C, N = points.shape
    _, M, K = index.shape
    out = np.empty((B, C, M, K), dtype=points.dtype)
    _check(_fn("s4g_oracle_group_points", points)(_fp(points), _ip(index), B, C, N, M, K, _fp(out)),
           "group_points")
    return out


def group_points_backward(grad_out, index, num_points):
    grad_out, index = _f32(grad_out), _i64a(index)
    B, C, M, K = grad_out.shape
    gin = np.empty((B, C, num_points), dtype=grad_out.dtype)
    _check(_fn("s4g_oracle_group_points_backward", grad_out)(_fp(grad_out), _ip(index), B, C, num_points,
                                                  M, K, _fp(gin)), "group_points_backward")
    return gin


def gather_points(points, index):
    points, index = _f32(points), _i64a(index)
    B, C, N = points.shape
    M = index.shape[1]
    out = np.empty((B, C, M), dtype=points.dtype)
    _check(_fn("s4g_oracle_gather_points", points)(_fp(points), _ip(index), B, C, N, M, _fp(out)),
           "gather_points")
    return out


def three_nn(query_xyz, key_xyz, fmad=0):
    """Returns (index (B,N1,3) int64, SQUARED distance (B,N1,3) fp32)."""
    q, k = _f32(query_xyz, key_xyz)
    B, _, N1 = q.shape
    N2 = k.shape[2]
    idx = np.empty((B, N1, 3), dtype=np.int64)
    d2 = np.empty((B, N1, 3), dtype=q.dtype)
    _check(_fn("s4g_oracle_three_nn", q)(_fp(q), _fp(k), B, N1, N2, _ip(idx), _fp(d2), fmad),
           "three_nn")
    return idx, d2


def interp_weights(d2, eps=1e-10):
    d2 = _f32(d2)
    B, N1, _ = d2.shape
    w = np.empty_like(d2)
    _check(_fn("s4g_oracle_interp_weights", d2)(_fp(d2), B, N1, _scalar(eps, d2), _fp(w)),
           "interp_weights")
    return w


def three_interpolate(feature, index, weight, fmad=0):
    (feature, weight), index = _f32(feature, weight), _i64a(index)
    B, C, N2 = feature.shape
    N1 = index.shape[1]
    weight = weight.astype(feature.dtype, copy=False)
    out = np.empty((B, C, N1), dtype=feature.dtype)
    _check(_fn("s4g_oracle_three_interpolate", feature)(_fp(feature), _ip(index), _fp(weight), B, C, N2,
                                              N1, _fp(out), fmad), "three_interpolate")
    return out


def three_interpolate_backward(grad_out, index, weight, num_inst):
    (grad_out, weight), index = _f32(grad_out, weight), _i64a(index)
    B, C, N1 = grad_out.shape
    weight = weight.astype(grad_out.dtype, copy=False)
    gin = np.empty((B, C, num_inst), dtype=grad_out.dtype)
    _check(_fn("s4g_oracle_three_interpolate_backward", grad_out)(_fp(grad_out), _ip(index), _fp(weight),
                                                       B, C, num_inst, N1, _fp(gin)),
           "three_interpolate_backward")
    return gin


def set_threads(n):
    """Team size of the C operators' OpenMP loops from now on (OMP_NUM_THREADS is only read when
    the OpenMP runtime starts); returns the size in effect."""
    return int(lib().s4g_oracle_set_threads(int(n)))

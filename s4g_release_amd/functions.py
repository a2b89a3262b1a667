"""Operator API of the PointNet++ hot path on MI355X.

Mirror of the reference's
``inference/grasp_proposal/network_models/models/pointnet2_utils/functions.py``
(same callables, positional signatures, return order and autograd behaviour:
``gather_points`` :10-25, ``farthest_point_sample`` :28-50, ``ball_query``
:53-80, ``group_points`` :83-109, ``search_nn_distance`` :112-135,
``feature_interpolate`` :145-174) over ``libs4g_hip.so`` instead of ``pn2_ext``.
North-star spellings (``furthest_point_sample``, ``three_nn``,
``three_interpolate``) are aliases.

Like the reference, there is no CPU path: tensors must live on a HIP device
(the reference's CHECK_CUDA), and a missing HIP library is an error.
Differences, all deliberate:
  * double tensors are accepted like in the reference (AT_DISPATCH_FLOATING_TYPES) and take plain
    kernels (csrc/ops_f64.hip); the tuned kernels and the fused fast path are fp32 -- S4G never leaves it;
  * kernels run on torch's CURRENT stream and on the tensor's device (the
    reference launches on the legacy default stream without a device guard);
  * inputs are consumed channel-first as they come -- no transposed copies.
"""
import os
import warnings

import torch

from . import _cabi

# 0 = canonical arithmetic (every fp32 op rounded), 1 = nvcc -fmad emulation.
_DIST_FLAGS = _cabi.S4G_FLAG_FMAD if os.environ.get("S4G_DIST_MODE", "strict") == "fmad" else 0


def set_distance_mode(mode):
    """'strict' (default; what the oracle pins) or 'fmad' (nvcc contraction)."""
    global _DIST_FLAGS
    if mode not in ("strict", "fmad"):
        raise ValueError("mode must be 'strict' or 'fmad'")
    _DIST_FLAGS = _cabi.S4G_FLAG_FMAD if mode == "fmad" else 0


# backward of group_points / three_interpolate: "deterministic" (default; sorted-segment sums in ascending position
# order, csrc/scatter.hip: run-to-run bit-identical, equal to the sequential sum) or "atomic" (the reference's
# atomicAdd scatter: fewer passes, undefined order)
_BACKWARD_MODE = "atomic" if os.environ.get("S4G_BACKWARD", "deterministic") == "atomic" else "deterministic"


def set_backward_mode(mode):
    """'deterministic' (default) or 'atomic' (grouping_kernel.cu:94 / interpolate_kernel.cu:283's scheme)."""
    global _BACKWARD_MODE
    if mode not in ("deterministic", "atomic"):
        raise ValueError("mode must be 'deterministic' or 'atomic'")
    _BACKWARD_MODE = mode


_DET_FALLBACK_WARNED = False


def _scatter_ws(dev, B, N, T, C=0, weighted=False):
    """Workspace of the deterministic scatters (with room for the channels-last copy of the gradients from 32 channels
    on), or (None, 0) where the sizes are outside their 32-bit keys."""
    nbytes = _cabi.lib().s4g_scatter_det_workspace_bytes_c(B, C, N, T, int(weighted))
    if nbytes == 0:
        global _DET_FALLBACK_WARNED
        if not _DET_FALLBACK_WARNED:      # once: the mode still says 'deterministic', the sums below are not
            _DET_FALLBACK_WARNED = True
            warnings.warn("s4g backward: B*T or B*N reaches 2^31 (B=%d, N=%d, T=%d): outside the deterministic scatter's "
                          "32-bit keys, falling back to the atomic (order-nondeterministic) kernels for this size"
                          % (B, N, T), RuntimeWarning, stacklevel=3)
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def _check_dev(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)  # CHECK_CUDA of the reference


def _f32c(t, name):
    _check_dev(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t.contiguous()


def _is_f64(*tensors):
    """The reference's extension dispatches over float and double (AT_DISPATCH_FLOATING_TYPES): double
    tensors take the *_f64 entry points (csrc/ops_f64.hip: plain kernels, same semantics; the tuned
    kernels are the float ones).  Mixed float / double arguments are an error, as in the reference."""
    kinds = {t.dtype for t in tensors if t is not None and t.is_floating_point()}
    if kinds == {torch.float64}:
        for t in tensors:
            if t is not None:
                _check_dev(t, "tensor")
        return True
    if torch.float64 in kinds:
        raise RuntimeError("expected all floating-point arguments to share one dtype (float or double)")
    return False


def _i64c(t, name):
    _check_dev(t, name)
    if t.dtype != torch.int64:
        raise RuntimeError("%s must be int64 (got %s)" % (name, t.dtype))
    return t.contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class OpTimer:
    """Optional HIP-event timing of every native launch (used by bench.py).

    Events are recorded on the stream the kernel is launched on (torch's
    current stream), immediately before and after the launch."""
    enabled = False
    records = []   # (op name, start event, end event, algorithmic bytes, flops)
    every = 1      # time the launches of every `every`-th forward pass only (bench.py: the event
    _pass = 0      # pairs are barrier packets in the queues they time; sampling keeps them out of
    _armed = True  # most steps of the timed region)

    @classmethod
    def reset(cls, enabled, every=1):
        cls.enabled = enabled
        cls.records = []
        cls.every = max(1, int(every))
        cls._pass = 0
        cls._armed = True

    @classmethod
    def begin_pass(cls):
        """Called once per submitted forward pass: arms the timers for one pass in `every`."""
        cls._armed = cls._pass % cls.every == 0
        cls._pass += 1

    @classmethod
    def active(cls):
        return cls.enabled and cls._armed

    @classmethod
    def summary(cls):
        """{op: (launches, mean ms, algorithmic bytes, flops per launch)}; call after a sync."""
        acc = {}
        for name, e0, e1, nbytes, flops in cls.records:
            n, t, b, f = acc.get(name, (0, 0.0, 0, 0.0))
            if callable(flops):        # data-dependent work: evaluated now, after the region's fence
                flops = flops()
            acc[name] = (n + 1, t + e0.elapsed_time(e1), b + nbytes, f + flops)
        return {k: (n, t / n, b / n, f / n) for k, (n, t, b, f) in acc.items()}


class _timed:
    def __init__(self, name, nbytes, flops=0.0):
        self.name, self.nbytes, self.flops = name, nbytes, flops

    def __enter__(self):
        self.on = OpTimer.active()
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            OpTimer.records.append((self.name, self.e0, self.e1, self.nbytes, self.flops))
        return False


def _ptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


def _workspace(op, dev, B, d0, d1, d2):
    nbytes = _cabi.lib().s4g_workspace_bytes(op, B, d0, d1, d2)
    if nbytes == 0:
        return None, 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return ws, nbytes


# ----------------------------------------------------------------------------
# raw (non-autograd) entry points: the seven pn2_ext functions
# (reference csrc/main.cpp:7-13)
# ----------------------------------------------------------------------------
def _farthest_point_sample(points, num_centroids):
    if _is_f64(points):
        points = points.contiguous()
        if points.dim() != 3 or points.size(1) != 3:
            raise RuntimeError("points.size(1) does not equal to 3")
        B, _, N = points.shape
        M = int(num_centroids)
        if not M > 0:
            raise RuntimeError("num_centroids is not greater than 0")
        if not N >= M:
            raise RuntimeError("num_points is not greater than or equal to num_centroids")
        index = torch.empty((B, M), dtype=torch.int64, device=points.device)
        ws = torch.empty((B, N), dtype=torch.float64, device=points.device)     # `temp` of sampling_kernel.cu:144
        with torch.cuda.device(points.device):
            rc = _cabi.lib().s4g_fps_f64(_ptr(points), B, N, M, _ptr(index), _ptr(ws), ws.numel() * 8,
                                         _DIST_FLAGS, _stream())
        _cabi.check(rc, "farthest_point_sample (double)")
        return index
    points = _f32c(points, "points")
    if points.dim() != 3 or points.size(1) != 3:
        raise RuntimeError("points.size(1) does not equal to 3")  # sampling_kernel.cu:137
    B, _, N = points.shape
    M = int(num_centroids)
    if not M > 0:
        raise RuntimeError("num_centroids is not greater than 0")  # :138
    if not N >= M:
        raise RuntimeError("num_points is not greater than or equal to num_centroids")  # :139
    index = torch.empty((B, M), dtype=torch.int64, device=points.device)
    with torch.cuda.device(points.device):
        ws, nbytes = _workspace(_cabi.S4G_OP_FPS, points.device, B, N, M, 0)
        with _timed("fps[N=%d,M=%d]" % (N, M), B * (12 * N + 8 * M)):
            rc = _cabi.lib().s4g_fps_f32(_ptr(points), B, N, M, _ptr(index), _ptr(ws), nbytes,
                                         _DIST_FLAGS, _stream())
    _cabi.check(rc, "farthest_point_sample")
    return index


def _ball_query(points, centroids, radius, num_neighbours):
    f64 = _is_f64(points, centroids)
    if f64:
        points, centroids = points.contiguous(), centroids.contiguous()
    else:
        points = _f32c(points, "points")
        centroids = _f32c(centroids, "centroids")
    if points.dim() != 3 or points.size(1) != 3:
        raise RuntimeError("points.size(1) does not equal to 3")  # ball_query_kernel.cu:102
    if centroids.dim() != 3 or centroids.size(1) != 3:
        raise RuntimeError("centroids.size(1) does not equal to 3")  # :103
    if centroids.size(0) != points.size(0):
        raise RuntimeError("points and centroids must share the batch size")
    B, _, N = points.shape
    M = centroids.size(2)
    K = int(num_neighbours)
    if K <= 0:
        raise RuntimeError("num_neighbours must be positive")
    index = torch.empty((B, M, K), dtype=torch.int64, device=points.device)
    count = torch.empty((B, M), dtype=torch.int64, device=points.device)
    if f64:
        with torch.cuda.device(points.device):
            rc = _cabi.lib().s4g_ball_query_f64(_ptr(points), _ptr(centroids), B, N, M, float(radius), K,
                                                _ptr(index), _ptr(count), _DIST_FLAGS, _stream())
        _cabi.check(rc, "ball_query (double)")
        return index, count
    with torch.cuda.device(points.device):
        ws, nbytes = _workspace(_cabi.S4G_OP_BALL_QUERY, points.device, B, N, M, K)
        with _timed("ball_query[N=%d,M=%d,K=%d]" % (N, M, K),
                    B * (12 * N + 12 * M + 8 * M * K + 8 * M)):
            rc = _cabi.lib().s4g_ball_query_f32(_ptr(points), _ptr(centroids), B, N, M,
                                                float(radius), K, _ptr(index), _ptr(count),
                                                _ptr(ws), nbytes, _DIST_FLAGS, _stream())
    _cabi.check(rc, "ball_query")
    return index, count


def query_and_group(points, centroids, radius, num_neighbours):
    """ball_query + group_points(points, index) in one pass over the outputs.

    Returns (index (B,M,K) int64, count (B,M) int64, grouped xyz (B,3,M,K) fp32),
    identical to the two reference operators called in sequence
    (QueryGrouper.forward, modules.py:39-42)."""
    if _is_f64(points, centroids):
        index, count = _ball_query(points, centroids, radius, num_neighbours)
        return index, count, _group_points_forward(points, index)
    points = _f32c(points, "points")
    centroids = _f32c(centroids, "centroids")
    if points.dim() != 3 or points.size(1) != 3:
        raise RuntimeError("points.size(1) does not equal to 3")  # ball_query_kernel.cu:102
    if centroids.dim() != 3 or centroids.size(1) != 3:
        raise RuntimeError("centroids.size(1) does not equal to 3")  # :103
    if centroids.size(0) != points.size(0):       # (the C entry takes ONE batch size: a mismatch would read out of bounds)
        raise RuntimeError("points and centroids must share the batch size")
    B, _, N = points.shape
    M = centroids.size(2)
    K = int(num_neighbours)
    if K <= 0:
        raise RuntimeError("num_neighbours must be positive")
    index = torch.empty((B, M, K), dtype=torch.int64, device=points.device)
    count = torch.empty((B, M), dtype=torch.int64, device=points.device)
    grouped = torch.empty((B, 3, M, K), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        ws, nbytes = _workspace(_cabi.S4G_OP_BALL_QUERY, points.device, B, N, M, K)
        with _timed("query_group[N=%d,M=%d,K=%d]" % (N, M, K),
                    B * (12 * N + 12 * M + 8 * M * K + 8 * M) +
                    B * (4 * 3 * N + 8 * M * K + 4 * 3 * M * K)):
            rc = _cabi.lib().s4g_query_group_f32(_ptr(points), _ptr(centroids), B, N, M,
                                                 float(radius), K, _ptr(index), _ptr(count),
                                                 _ptr(grouped), _ptr(ws), nbytes, _DIST_FLAGS,
                                                 _stream())
    _cabi.check(rc, "query_group")
    return index, count, grouped


def _group_points_forward(points, index):
    if _is_f64(points):
        points, index = points.contiguous(), _i64c(index, "index")
        if points.dim() != 3 or index.dim() != 3 or index.size(0) != points.size(0):
            raise RuntimeError("input / index must be 3-d and share the batch size")
        B, C, N = points.shape
        _, M, K = index.shape
        out = torch.empty((B, C, M, K), dtype=torch.float64, device=points.device)
        with torch.cuda.device(points.device):
            rc = _cabi.lib().s4g_group_points_f64(_ptr(points), _ptr(index), B, C, N, M, K, _ptr(out), _stream())
        _cabi.check(rc, "group_points_forward (double)")
        return out
    points = _f32c(points, "input")
    index = _i64c(index, "index")
    if points.dim() != 3:
        raise RuntimeError("input.dim() does not equal to 3")  # grouping_kernel.cu:44
    if index.dim() != 3:
        raise RuntimeError("index.dim() does not equal to 3")  # :45
    if index.size(0) != points.size(0):
        raise RuntimeError("index.size(0) does not equal to batch_size")  # :46
    B, C, N = points.shape
    _, M, K = index.shape
    out = torch.empty((B, C, M, K), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        if C == 3 and B > 0 and (M * K) % 4 == 0 and M * K >= 4096:
            # xyz grouping: one 16-byte gather per neighbour out of an (x, y, z, 0) copy
            aos = torch.empty((B * N, 4), dtype=torch.float32, device=points.device)
            with _timed("group_points[C=%d,N=%d,M=%d,K=%d]" % (C, N, M, K),
                        B * (4 * C * N + 8 * M * K + 4 * C * M * K)):
                rc = _cabi.lib().s4g_group_points_xyz_f32(_ptr(points), _ptr(index), B, N, M, K,
                                                          _ptr(out), aos.data_ptr(), aos.numel() * 4,
                                                          _stream())
        elif C % 4 == 0 and C >= 16 and B > 0 and M * K >= 4096:
            # feature grouping: 64 channels of a neighbour are one 256-byte read of a
            # channels-last copy, the channel-first rows leave through an LDS tile
            cl = torch.empty((B * N, C), dtype=torch.float32, device=points.device)
            with _timed("group_points[C=%d,N=%d,M=%d,K=%d]" % (C, N, M, K),
                        B * (4 * C * N + 8 * M * K + 4 * C * M * K)):
                rc = _cabi.lib().s4g_group_points_ws_f32(_ptr(points), _ptr(index), B, C, N, M, K,
                                                         _ptr(out), cl.data_ptr(), cl.numel() * 4,
                                                         _stream())
        else:
            with _timed("group_points[C=%d,N=%d,M=%d,K=%d]" % (C, N, M, K),
                        B * (4 * C * N + 8 * M * K + 4 * C * M * K)):
                rc = _cabi.lib().s4g_group_points_f32(_ptr(points), _ptr(index), B, C, N, M, K,
                                                      _ptr(out), _stream())
    _cabi.check(rc, "group_points_forward")
    return out


def _group_points_backward(grad_output, index, num_points):
    if _is_f64(grad_output):
        grad_output, index = grad_output.contiguous(), _i64c(index, "index")
        if grad_output.dim() != 4 or index.dim() != 3:
            raise RuntimeError("grad_output must be 4-d and index 3-d")  # grouping_kernel.cu:118-119
        B, C, M, K = grad_output.shape
        if tuple(index.shape) != (B, M, K):
            raise RuntimeError("index shape does not match grad_output")  # :120-122
        gin = torch.empty((B, C, int(num_points)), dtype=torch.float64, device=grad_output.device)
        with torch.cuda.device(grad_output.device):
            rc = _cabi.lib().s4g_group_points_backward_f64(_ptr(grad_output), _ptr(index), B, C, int(num_points), M, K,
                                                           _ptr(gin), _stream())
        _cabi.check(rc, "group_points_backward (double)")
        return gin
    grad_output = _f32c(grad_output, "grad_output")
    index = _i64c(index, "index")
    if grad_output.dim() != 4 or index.dim() != 3:
        raise RuntimeError("grad_output must be 4-d and index 3-d")  # grouping_kernel.cu:118-119
    B, C, M, K = grad_output.shape
    if tuple(index.shape) != (B, M, K):
        raise RuntimeError("index shape does not match grad_output")  # :120-122
    gin = torch.empty((B, C, int(num_points)), dtype=torch.float32, device=grad_output.device)
    with torch.cuda.device(grad_output.device):
        ws, nbytes = _scatter_ws(grad_output.device, B, int(num_points), M * K, C) \
            if _BACKWARD_MODE == "deterministic" and gin.numel() > 0 else (None, 0)
        if ws is not None:
            rc = _cabi.lib().s4g_group_points_backward_det_f32(_ptr(grad_output), _ptr(index), B, C, int(num_points),
                                                               M, K, _ptr(gin), _ptr(ws), nbytes, _stream())
        else:
            rc = _cabi.lib().s4g_group_points_backward_f32(_ptr(grad_output), _ptr(index), B, C,
                                                           int(num_points), M, K, _ptr(gin), _stream())
    _cabi.check(rc, "group_points_backward")
    return gin


def _point_search(query_xyz, key_xyz, num_neighbours):
    if _is_f64(query_xyz, key_xyz):
        query_xyz, key_xyz = query_xyz.contiguous(), key_xyz.contiguous()
        B, N1, N2 = query_xyz.size(0), query_xyz.size(2), key_xyz.size(2)
        if key_xyz.size(0) != B or query_xyz.size(1) != 3 or key_xyz.size(1) != 3 or int(num_neighbours) != 3 or N2 < 3:
            raise RuntimeError("search_nn_distance: (B, 3, N1) queries, (B, 3, N2 >= 3) keys, 3 neighbours")
        index = torch.empty((B, N1, 3), dtype=torch.int64, device=query_xyz.device)
        dist = torch.empty((B, N1, 3), dtype=torch.float64, device=query_xyz.device)
        with torch.cuda.device(query_xyz.device):
            rc = _cabi.lib().s4g_three_nn_f64(_ptr(query_xyz), _ptr(key_xyz), B, N1, N2, _ptr(index), _ptr(dist),
                                              _DIST_FLAGS, _stream())
        _cabi.check(rc, "point_search (double)")
        return index, dist
    query_xyz = _f32c(query_xyz, "query_xyz")
    key_xyz = _f32c(key_xyz, "key_xyz")
    B = query_xyz.size(0)
    if key_xyz.size(0) != B:
        raise RuntimeError("key_xyz.size(0) does not equal to batch_size")  # interpolate_kernel.cu:101
    if query_xyz.size(1) != 3 or key_xyz.size(1) != 3:
        raise RuntimeError("xyz.size(1) does not equal to 3")  # :102-103
    if int(num_neighbours) != 3:
        raise RuntimeError("num_neighbours does not equal to K (3)")  # :105
    N1, N2 = query_xyz.size(2), key_xyz.size(2)
    if not N2 >= 3:
        raise RuntimeError("num_key is not greater than or equal to num_neighbours")  # :106
    index = torch.empty((B, N1, 3), dtype=torch.int64, device=query_xyz.device)
    dist = torch.empty((B, N1, 3), dtype=torch.float32, device=query_xyz.device)
    nbytes_alg = B * (12 * N2 + 12 * N1 + 24 * N1 + 12 * N1)
    use_grid = (2048 <= N2 <= 65536 and B * N1 > 0 and
                _cabi.knob("S4G_NN_MODE", "grid") != "scan")
    cell = -1.0   # the library derives the cell edge from the keys' extent on the device
    with torch.cuda.device(query_xyz.device):
        if use_grid:
            # keys binned into a cell grid, 27 cells per query, index-order scan for the
            # queries it cannot answer: same indices and distances as the scan, whatever the cell
            nbytes = _cabi.lib().s4g_three_nn_grid_workspace_bytes(B, N1, N2)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=query_xyz.device)
            with _timed("three_nn[N1=%d,N2=%d]" % (N1, N2), nbytes_alg):
                rc = _cabi.lib().s4g_three_nn_grid_f32(_ptr(query_xyz), _ptr(key_xyz), B, N1, N2, cell,
                                                       _ptr(index), _ptr(dist), ws.data_ptr(), nbytes,
                                                       _DIST_FLAGS, _stream())
        else:
            with _timed("three_nn[N1=%d,N2=%d]" % (N1, N2), nbytes_alg):
                rc = _cabi.lib().s4g_three_nn_f32(_ptr(query_xyz), _ptr(key_xyz), B, N1, N2,
                                                  _ptr(index), _ptr(dist), None, 0, _DIST_FLAGS,
                                                  _stream())
    _cabi.check(rc, "point_search")
    return index, dist


def _interpolate_forward(feature, index, weight):
    if _is_f64(feature, weight):
        feature, index, weight = feature.contiguous(), _i64c(index, "index"), weight.contiguous()
        B, C, N2 = feature.shape
        N1 = index.size(1)
        if tuple(index.shape) != (B, N1, 3) or tuple(weight.shape) != (B, N1, 3):
            raise RuntimeError("index / weight must be (batch_size, N, 3)")
        out = torch.empty((B, C, N1), dtype=torch.float64, device=feature.device)
        with torch.cuda.device(feature.device):
            rc = _cabi.lib().s4g_three_interpolate_f64(_ptr(feature), _ptr(index), _ptr(weight), B, C, N2, N1, _ptr(out),
                                                       _DIST_FLAGS, _stream())
        _cabi.check(rc, "interpolate_forward (double)")
        return out
    feature = _f32c(feature, "input")
    index = _i64c(index, "index")
    weight = _f32c(weight, "weight")
    B, C, N2 = feature.shape
    if index.size(0) != B or index.size(2) != 3:
        raise RuntimeError("index must be (batch_size, N, 3)")  # interpolate_kernel.cu:202-203
    N1 = index.size(1)
    if tuple(weight.shape) != (B, N1, 3):
        raise RuntimeError("weight must be (batch_size, N, 3)")  # :204-206
    out = torch.empty((B, C, N1), dtype=torch.float32, device=feature.device)
    with torch.cuda.device(feature.device):
        if C % 4 == 0 and C >= 16 and B * N1 > 0 and N1 >= N2:
            # channels-last copy of the sparse features: one 16-byte gather per channel quad
            ws = torch.empty((B * N2, C), dtype=torch.float32, device=feature.device)
            with _timed("three_interpolate[C=%d,N2=%d,N1=%d]" % (C, N2, N1),
                        B * (4 * C * N2 + 24 * N1 + 12 * N1 + 4 * C * N1)):
                rc = _cabi.lib().s4g_three_interpolate_ws_f32(_ptr(feature), _ptr(index), _ptr(weight),
                                                              B, C, N2, N1, _ptr(out), ws.data_ptr(),
                                                              ws.numel() * 4, _DIST_FLAGS, _stream())
        else:
            with _timed("three_interpolate[C=%d,N2=%d,N1=%d]" % (C, N2, N1),
                        B * (4 * C * N2 + 24 * N1 + 12 * N1 + 4 * C * N1)):
                rc = _cabi.lib().s4g_three_interpolate_f32(_ptr(feature), _ptr(index), _ptr(weight), B,
                                                           C, N2, N1, _ptr(out), _DIST_FLAGS,
                                                           _stream())
    _cabi.check(rc, "interpolate_forward")
    return out


def _interpolate_backward(grad_output, index, weight, num_inst):
    if _is_f64(grad_output, weight):
        grad_output, index, weight = grad_output.contiguous(), _i64c(index, "index"), weight.contiguous()
        if grad_output.dim() != 3:
            raise RuntimeError("grad_output must be (batch_size, channels, N)")
        B, C, N1 = grad_output.shape
        if tuple(index.shape) != (B, N1, 3) or tuple(weight.shape) != (B, N1, 3):
            raise RuntimeError("index / weight must be (batch_size, N, 3)")  # interpolate_kernel.cu:307-311
        gin = torch.empty((B, C, int(num_inst)), dtype=torch.float64, device=grad_output.device)
        with torch.cuda.device(grad_output.device):
            rc = _cabi.lib().s4g_three_interpolate_backward_f64(_ptr(grad_output), _ptr(index), _ptr(weight), B, C,
                                                                int(num_inst), N1, _ptr(gin), _stream())
        _cabi.check(rc, "interpolate_backward (double)")
        return gin
    grad_output = _f32c(grad_output, "grad_output")
    index = _i64c(index, "index")
    weight = _f32c(weight, "weight")
    B, C, N1 = grad_output.shape
    if tuple(index.shape) != (B, N1, 3) or tuple(weight.shape) != (B, N1, 3):
        raise RuntimeError("index / weight must be (batch_size, N, 3)")  # :307-311
    gin = torch.empty((B, C, int(num_inst)), dtype=torch.float32, device=grad_output.device)
    with torch.cuda.device(grad_output.device):
        ws, nbytes = _scatter_ws(grad_output.device, B, int(num_inst), 3 * N1, C, True) \
            if _BACKWARD_MODE == "deterministic" and gin.numel() > 0 else (None, 0)
        if ws is not None:
            rc = _cabi.lib().s4g_three_interpolate_backward_det_f32(_ptr(grad_output), _ptr(index), _ptr(weight), B, C,
                                                                    int(num_inst), N1, _ptr(gin), _ptr(ws), nbytes,
                                                                    _stream())
        else:
            rc = _cabi.lib().s4g_three_interpolate_backward_f32(_ptr(grad_output), _ptr(index),
                                                                _ptr(weight), B, C, int(num_inst), N1,
                                                                _ptr(gin), _stream())
    _cabi.check(rc, "interpolate_backward")
    return gin


def three_nn_weights_grid(query_xyz, key_xyz, cell, eps=1e-10):
    """Fast-path 3-NN: (index (B,N1,3) int32, interpolation weights (B,N1,3) fp32) through
    the cell-grid kernel (`s4g_three_nn_weights_grid_i32`); bit-identical to
    `search_nn_distance` + `interp_weights` for every input (unanswered queries fall back
    to the index-order scan inside the call).  `cell` = grid edge, e.g. the SA radius."""
    query_xyz = _f32c(query_xyz, "query_xyz")
    key_xyz = _f32c(key_xyz, "key_xyz")
    B, _, N1 = query_xyz.shape
    N2 = key_xyz.size(2)
    if not N2 >= 3:
        raise RuntimeError("num_key is not greater than or equal to num_neighbours")
    idx = torch.empty((B, N1, 3), dtype=torch.int32, device=query_xyz.device)
    w = torch.empty((B, N1, 3), dtype=torch.float32, device=query_xyz.device)
    with torch.cuda.device(query_xyz.device):
        nbytes = _cabi.lib().s4g_three_nn_grid_workspace_bytes(B, N1, N2)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=query_xyz.device)
        rc = _cabi.lib().s4g_three_nn_weights_grid_i32(_ptr(query_xyz), _ptr(key_xyz), B, N1, N2,
                                                       float(eps), float(cell), _ptr(idx), _ptr(w),
                                                       _ptr(ws), nbytes, _DIST_FLAGS, _stream())
    _cabi.check(rc, "three_nn_weights_grid")
    return idx, w


def three_nn_weights(query_xyz, key_xyz, eps=1e-10):
    """Fast-path 3-NN by index-order scan: (index (B,N1,3) int32, interpolation weights
    (B,N1,3) fp32), bit-identical to `search_nn_distance` + `interp_weights`
    (`s4g_three_nn_weights_i32`: split scan for small key sets, lane-per-query scan otherwise)."""
    query_xyz = _f32c(query_xyz, "query_xyz")
    key_xyz = _f32c(key_xyz, "key_xyz")
    B, _, N1 = query_xyz.shape
    N2 = key_xyz.size(2)
    if not N2 >= 3:
        raise RuntimeError("num_key is not greater than or equal to num_neighbours")
    idx = torch.empty((B, N1, 3), dtype=torch.int32, device=query_xyz.device)
    w = torch.empty((B, N1, 3), dtype=torch.float32, device=query_xyz.device)
    with torch.cuda.device(query_xyz.device):
        rc = _cabi.lib().s4g_three_nn_weights_i32(_ptr(query_xyz), _ptr(key_xyz), B, N1, N2, float(eps),
                                                  _ptr(idx), _ptr(w), None, 0, _DIST_FLAGS, _stream())
    _cabi.check(rc, "three_nn_weights")
    return idx, w


def interp_weights(distance, eps=1e-10):
    """Inverse-distance weights of FeatureInterpolator.forward (modules.py:118-120)
    in one launch instead of three elementwise ones."""
    if _is_f64(distance):      # modules.py:118-120 as written, in the tensor's own type
        inv = 1.0 / torch.clamp(distance, min=eps)
        return inv / inv.sum(dim=2, keepdim=True)
    distance = _f32c(distance, "distance")
    B, N1, _ = distance.shape
    w = torch.empty_like(distance)
    with torch.cuda.device(distance.device):
        rc = _cabi.lib().s4g_interp_weights_f32(_ptr(distance), B, N1, float(eps), _ptr(w),
                                                _stream())
    _cabi.check(rc, "interp_weights")
    return w


# ----------------------------------------------------------------------------
# public operator API (reference functions.py)
# ----------------------------------------------------------------------------
def gather_points(points, index):
    """Gather xyz of centroids according to indices (functions.py:10-25).

    points (B, C, N), index (B, M) -> (B, C, M).  Differentiable like the
    reference's torch.gather.
    """
    if points.requires_grad:
        index_expand = index.unsqueeze(1).expand(points.size(0), points.size(1), index.size(1))
        return points.gather(2, index_expand)
    if _is_f64(points):
        return _group_points_forward(points, _i64c(index, "index").unsqueeze(-1)).squeeze(-1)
    points = _f32c(points, "points")
    index = _i64c(index, "index")
    B, C, N = points.shape
    M = index.size(1)
    out = torch.empty((B, C, M), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        rc = _cabi.lib().s4g_gather_points_f32(_ptr(points), _ptr(index), B, C, N, M, _ptr(out),
                                               _stream())
    _cabi.check(rc, "gather_points")
    return out


class FarthestPointSample(torch.autograd.Function):
    """functions.py:28-47"""

    @staticmethod
    def forward(ctx, points, num_centroids):
        index = _farthest_point_sample(points, num_centroids)
        ctx.mark_non_differentiable(index)
        return index

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None


farthest_point_sample = FarthestPointSample.apply


class BallQuery(torch.autograd.Function):
    """functions.py:53-77"""

    @staticmethod
    def forward(ctx, points, centroids, radius, num_neighbours):
        index, count = _ball_query(points, centroids, radius, num_neighbours)
        ctx.mark_non_differentiable(index, count)
        return index, count

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None, None, None


ball_query = BallQuery.apply


class GroupPoints(torch.autograd.Function):
    """functions.py:83-106"""

    @staticmethod
    def forward(ctx, points, index):
        ctx.save_for_backward(index)
        ctx.num_points = points.size(2)
        return _group_points_forward(points, index)

    @staticmethod
    def backward(ctx, *grad_output):
        index = ctx.saved_tensors[0]
        grad_input = _group_points_backward(grad_output[0], index, ctx.num_points)
        return grad_input, None


group_points = GroupPoints.apply


class SearchNNDistance(torch.autograd.Function):
    """functions.py:112-132.  Returns (index, SQUARED distance), in that order."""

    @staticmethod
    def forward(ctx, query_xyz, key_xyz, num_neighbors):
        index, distance = _point_search(query_xyz, key_xyz, num_neighbors)
        ctx.mark_non_differentiable(index, distance)
        return index, distance

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None, None


search_nn_distance = SearchNNDistance.apply


class FeatureInterpolate(torch.autograd.Function):
    """functions.py:145-171"""

    @staticmethod
    def forward(ctx, feature, index, weight):
        _, _, num_inst = feature.size()
        ctx.save_for_backward(index, weight)
        ctx.num_inst = num_inst
        return _interpolate_forward(feature, index, weight)

    @staticmethod
    def backward(ctx, *grad_out):
        index, weight = ctx.saved_tensors
        grad_input = _interpolate_backward(grad_out[0], index, weight, ctx.num_inst)
        return grad_input, None, None


feature_interpolate = FeatureInterpolate.apply

class GatherKNN(torch.autograd.Function):
    """Drop-in for the reference's second native extension `dgcnn_ext`
    (`network_models/functions/gather_knn.py:10-24`, kernels
    `functions/csrc/gather_knn_kernel.cu:27-153`): feature (B,C,N), index (B,N,K)
    -> (B,C,N,K).  It is group_points with M == N, forward and backward, so it
    runs on the same two HIP kernels."""

    @staticmethod
    def forward(ctx, feature, index):
        ctx.save_for_backward(index)
        ctx.num_points = feature.size(2)
        return _group_points_forward(feature, index)

    @staticmethod
    def backward(ctx, grad_output):
        index = ctx.saved_tensors[0]
        return _group_points_backward(grad_output, index, ctx.num_points), None


gather_knn = GatherKNN.apply

# north-star spellings (erikwijmans lineage); same objects.
furthest_point_sample = farthest_point_sample


def three_nn(query_xyz, key_xyz):
    """Alias of search_nn_distance(q, k, 3); returns (index, squared distance)
    in the REFERENCE's order (functions.py:127-128), not (dist, idx)."""
    return search_nn_distance(query_xyz, key_xyz, 3)


three_interpolate = feature_interpolate

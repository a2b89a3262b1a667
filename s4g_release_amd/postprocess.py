"""Pose decode on the device -- the step right after `PointNet2.forward` in the
reference's callers (`grasp_proposal_test.py:83` -> `utils/file_logger_cls.py`,
`grasp_detector.py:137-185`); SURVEY.md section 8f row f1.

`decode_top_poses` turns the four head tensors into the K best grasp frames per
scene entirely on the GPU, so a serving loop ships K x 18 floats per scene
instead of 21 x N.  Collision filtering (row f2) is not part of it.
"""
import ctypes
import math
import threading
from collections import OrderedDict
from dataclasses import dataclass

import torch

from . import _cabi
from . import functions as _F

T_BINS = (0.08, 0.06, 0.04, 0.02)          # file_logger_cls.py:45, grasp_detector.py:177


def score_values(num_classes, convention="demo"):
    """`demo`: linspace(0,1,C+1)[:-1] (file_logger_cls.py:67); `detector`:
    linspace(0,1,C+1)[1:] (grasp_detector.py:147)."""
    v = torch.linspace(0, 1, num_classes + 1, dtype=torch.float64)
    return (v[:-1] if convention == "demo" else v[1:]).float()


_DEV_CONST = {}


def _dev_const(key, make, device):
    """Small constant tensors on the device, created once: a host -> device copy from pageable memory makes the HOST wait
    for everything queued in front of it on the stream (measured: a pipelined submit went from 0.6 to 10 ms of host time)."""
    k = (key, str(device))
    t = _DEV_CONST.get(k)
    if t is None:
        t = _DEV_CONST[k] = make().to(device)
    return t


_SMALL_LRU = OrderedDict()      # caller-supplied small matrices by value, at most _SMALL_LRU_MAX of them
_SMALL_LRU_MAX = 16
_SMALL_LOCK = threading.Lock()


def _small_on_device(x, dtype, device):
    """A caller's small matrix / vector (nested tuples, numpy, CPU tensor) on the device WITHOUT a per-call host -> device
    copy where the value repeats: a BOUNDED least-recently-used cache by value (16 entries -- `direction_matrix =
    camera2base[:3, :3] @ TRAIN2REAL` changes per capture on a moving camera; an unbounded cache would pin one device
    allocation per distinct pose for good).  A value holding NaN is never cached (NaN != NaN: it would miss on every
    call and leak).  A tensor that already lives on a device is passed through (`.to`): no host copy, no `.tolist()`
    synchronisation -- a serving loop that builds its matrices on the device pays nothing here."""
    if isinstance(x, torch.Tensor) and x.device.type != "cpu":
        return x.to(device=device, dtype=dtype)
    t = torch.as_tensor(x, dtype=torch.float64)
    vals = tuple(t.flatten().tolist())
    if any(v != v for v in vals):
        return t.to(dtype).to(device)
    key = (vals, tuple(t.shape), str(dtype), str(device))
    with _SMALL_LOCK:
        hit = _SMALL_LRU.get(key)
        if hit is not None:
            _SMALL_LRU.move_to_end(key)
            return hit
    d = t.to(dtype).to(device)
    with _SMALL_LOCK:
        _SMALL_LRU[key] = d
        while len(_SMALL_LRU) > _SMALL_LRU_MAX:
            _SMALL_LRU.popitem(last=False)
    return d


def expected_score(score_logits, convention="demo"):
    logits = _F._f32c(score_logits, "score")
    B, C, N = logits.shape
    vals = _dev_const(("score_values", C, convention), lambda: score_values(C, convention), logits.device)
    out = torch.empty((B, N), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = _cabi.lib().s4g_expected_score_f32(logits.data_ptr(), B, C, N, vals.data_ptr(),
                                                out.data_ptr(), _F._stream())
    _cabi.check(rc, "expected_score")
    return out


def _kept_points(predictions, scene_points):
    """Predictions of `FusedPointNet2(..., topk=K)` cover a scene's K best-scoring points only and carry their point
    numbers in "index": (those points' coordinates (B, 3, K), index) -- or (scene_points, None) for a full forward."""
    idx = predictions.get("index") if hasattr(predictions, "get") else None
    if idx is None:
        return scene_points, None
    return torch.gather(scene_points, 2, idx.unsqueeze(1).expand(-1, 3, -1)).contiguous(), idx


def decode_top_poses(predictions, scene_points, num_poses=50, convention="demo"):
    """-> (H (B,K,4,4) fp32, score (B,K) fp32, index (B,K) int64), best first
    (file_logger_cls.py:196-218: K = 50, argsort(-score)[:K], Gram-Schmidt).  Predictions over a scene's kept points
    (`FusedPointNet2(..., topk=)`) are accepted too: the returned index numbers the scene's points either way."""
    scene_points, kept = _kept_points(predictions, scene_points)
    if kept is not None:
        H, top, sel = decode_top_poses({k: v for k, v in predictions.items() if k != "index"}, scene_points,
                                       num_poses, convention)
        return H, top, torch.gather(kept, 1, sel)
    xyz = _F._f32c(scene_points, "scene_points")
    R = _F._f32c(predictions["frame_R"], "frame_R")
    t = _F._f32c(predictions["frame_t"], "frame_t")
    score = expected_score(predictions["score"], convention)
    B, _, N = xyz.shape
    K = min(int(num_poses), N)
    top, sel = torch.topk(score, K, dim=1, largest=True, sorted=True)
    sel = sel.contiguous()
    bins = _dev_const(("t_bins", t.shape[1]), lambda: torch.tensor(T_BINS[:t.shape[1]], dtype=torch.float32), xyz.device)
    H = torch.empty((B, K, 4, 4), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        rc = _cabi.lib().s4g_decode_poses_f32(xyz.data_ptr(), R.data_ptr(), t.data_ptr(),
                                              sel.data_ptr(), B, N, K, t.shape[1], bins.data_ptr(),
                                              H.data_ptr(), _F._stream())
    _cabi.check(rc, "decode_poses")
    return H, top, sel


REAL2TRAIN = ((0., 1., 0., 0.), (1., 0., 0., 0.), (0., 0., -1., 0.), (0., 0., 0., 1.))   # grasp_detector.py:26
TRAIN2REAL = REAL2TRAIN                               # :27 (the matrix is its own inverse)


def _detect_poses_as_written(predictions, scene_points, score_threshold, verticalness_threshold,
                             direction_matrix, vertical_direction, frame, max_poses):
    """`GraspDetector.post_processing` with the reference's indexing EXACTLY as written
    (grasp_detector.py:149-167), on the device.  Two quirks of those lines are reproduced:
      * :150-153 `index_high2low` (positions inside `high_score_index`, best score first) indexes
        the POINT axis of `frame_R`;
      * :154 `rotation.transpose(0, 1)` is a numpy call, i.e. the identity permutation, so the
        (9, n) array is reshaped row-major into n 3x3 blocks: block m, entry e is the flat element
        f = 9 m + e of the (9, n) array = frame_R[f // n, index_high2low[f % n]].
    Block m is then paired with point high_score_index[m] (:160-167); survivors of the verticalness
    test keep ASCENDING point order.  The expected score is formed like the reference's (fp32
    softmax, float64 weighted sum) so that thresholding and ordering see the same numbers."""
    xyz = _F._f32c(scene_points, "scene_points")
    B, _, N = xyz.shape
    dev = xyz.device
    K = min(int(max_poses), N)
    dm = torch.eye(3, dtype=torch.float64, device=dev) if direction_matrix is None else \
        torch.as_tensor(direction_matrix, dtype=torch.float64, device=dev)
    v = torch.as_tensor(vertical_direction, dtype=torch.float32, device=dev).double()
    vals = torch.linspace(0, 1, predictions["score"].shape[1] + 1, dtype=torch.float64, device=dev)[1:]
    bins = torch.tensor(T_BINS[:predictions["frame_t"].shape[1]], dtype=torch.float32, device=dev)
    fr = torch.as_tensor(frame, dtype=torch.float32, device=dev)
    H = torch.zeros((B, K, 4, 4), dtype=torch.float32, device=dev)
    top = torch.zeros((B, K), dtype=torch.float32, device=dev)
    sel = torch.full((B, K), -1, dtype=torch.int64, device=dev)
    count = torch.zeros((B,), dtype=torch.int64, device=dev)
    for b in range(B):       # ragged per scene (the reference itself only accepts B == 1, :49)
        prob = torch.softmax(predictions["score"][b].float(), dim=0)                  # :143
        score = (vals.view(-1, 1) * prob.double()).sum(dim=0)                          # :145-146
        high = torch.nonzero(score > score_threshold).flatten()                        # :149
        n = int(high.numel())
        if n == 0:
            continue
        h2l = torch.argsort(score[high], descending=True)                              # :150
        flat = predictions["frame_R"][b].float()[:, h2l].reshape(-1)                   # :153, (9 n,)
        rot = flat.view(n, 3, 3)                                                       # :154
        xdir = -(dm @ rot[:, :, 0].double().t())                                       # :155, (3, n)
        vertical = (xdir.t() * v.view(1, 3)).sum(dim=1)                                # :156
        good = torch.nonzero(vertical > verticalness_threshold).flatten()              # :157
        valid = high[good][:K]                                                         # :160
        good = good[:K]
        m = int(valid.numel())
        if m == 0:
            continue
        # the decode kernel works on (9, m) / (tc, m) / (3, m) columns: hand it the blocks as columns
        Rm = rot[good].reshape(m, 9).t().contiguous().unsqueeze(0)                     # :164
        tm = predictions["frame_t"][b].float()[:, valid].contiguous().unsqueeze(0)     # :165
        pm = xyz[b][:, valid].contiguous().unsqueeze(0)                                # :163
        ar = torch.arange(m, dtype=torch.int64, device=dev).unsqueeze(0)
        Hb = torch.empty((1, m, 4, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _cabi.lib().s4g_decode_poses_f32(pm.data_ptr(), Rm.data_ptr(), tm.data_ptr(),
                                                  ar.data_ptr(), 1, m, m, tm.shape[1], bins.data_ptr(),
                                                  Hb.data_ptr(), _F._stream())
        _cabi.check(rc, "decode_poses")
        H[b, :m] = torch.matmul(fr.view(1, 4, 4), Hb[0])                               # :180
        top[b, :m] = score[valid].float()                                              # :167
        sel[b, :m] = valid
        count[b] = m
    return H, top, sel, count


def detect_poses(predictions, scene_points, score_threshold=0.7, verticalness_threshold=0.2,
                 direction_matrix=None, vertical_direction=(0.0, 0.0, 1.0), frame=TRAIN2REAL,
                 max_poses=1024, reference_indexing=False):
    """`GraspDetector.post_processing` (grasp_detector.py:137-185) for a whole batch, on the device
    and without a host round trip: expected score with the detector's class values (:145-146),
    score threshold (:149), survivors in descending score order (:150-151), verticalness filter
    `(-direction_matrix @ R[:, 0]) . vertical_direction > threshold` (:155-157; direction_matrix is
    the caller's `camera2base[:3,:3] @ TRAIN2REAL[:3,:3]`, identity if None), translation decode
    and Gram-Schmidt (:124-135,177-179), result moved into the caller's frame (`frame @ H`, :180).

    Returns (H (B, max_poses, 4, 4) fp32, score (B, max_poses), index (B, max_poses) int64,
    count (B,) int64): per scene the first count[b] rows are the detections, best first; rows past
    the count are zero / -1.

    Default: every pose is built from its OWN point's rotation and translation, best score first --
    what the reference's comments describe.  `reference_indexing=True` reproduces what its lines
    :149-167 actually compute (a position list used as point indices and a numpy transpose that is
    a no-op; see `_detect_poses_as_written`): same poses, scores and order as the reference, pinned
    by tests/golden/post_detector.npz which the reference's own function generated."""
    scene_points, kept = _kept_points(predictions, scene_points)
    if kept is not None:
        # (exact whenever the kept points contain every candidate the filters would pass among the best max_poses:
        #  always when fewer than K points of a scene exceed the score threshold)
        if reference_indexing:
            raise ValueError("reference_indexing=True restates the reference's point-axis quirks: it needs the full forward")
        H, top, sel, count = detect_poses({k: v for k, v in predictions.items() if k != "index"}, scene_points,
                                          score_threshold, verticalness_threshold, direction_matrix=direction_matrix,
                                          vertical_direction=vertical_direction, frame=frame, max_poses=max_poses)
        return H, top, torch.where(sel >= 0, torch.gather(kept, 1, sel.clamp(min=0)), sel), count
    if reference_indexing:
        return _detect_poses_as_written(predictions, scene_points, score_threshold, verticalness_threshold,
                                        direction_matrix, vertical_direction, frame, max_poses)
    xyz = _F._f32c(scene_points, "scene_points")
    R = _F._f32c(predictions["frame_R"], "frame_R")
    t = _F._f32c(predictions["frame_t"], "frame_t")
    score = expected_score(predictions["score"], "detector")                    # (B, N)
    B, _, N = xyz.shape
    dev = xyz.device
    dm = _small_on_device(((1., 0., 0.), (0., 1., 0.), (0., 0., 1.)) if direction_matrix is None else direction_matrix,
                          torch.float32, dev)
    v = _small_on_device(vertical_direction, torch.float32, dev)
    w = -(dm.t() @ v)                                                           # (-A r0) . v == r0 . (-A^T v)
    r0 = R.view(B, 3, 3, N)[:, :, 0, :]                                         # first column of every R
    vertical = (r0 * w.view(1, 3, 1)).sum(dim=1)                                # (B, N)
    keep = (score > score_threshold) & (vertical > verticalness_threshold)
    key = torch.where(keep, score, torch.full_like(score, float("-inf")))
    K = min(int(max_poses), N)
    top, sel = torch.sort(key, dim=1, descending=True, stable=True)
    top, sel = top[:, :K], sel[:, :K].contiguous()
    count = keep.sum(dim=1).clamp(max=K)
    bins = _dev_const(("t_bins", t.shape[1]), lambda: torch.tensor(T_BINS[:t.shape[1]], dtype=torch.float32), dev)
    H = torch.empty((B, K, 4, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().s4g_decode_poses_f32(xyz.data_ptr(), R.data_ptr(), t.data_ptr(),
                                              sel.data_ptr(), B, N, K, t.shape[1], bins.data_ptr(),
                                              H.data_ptr(), _F._stream())
    _cabi.check(rc, "decode_poses")
    fr = _small_on_device(frame, torch.float32, dev)
    # frame @ H for every pose: a broadcast multiply + sum over the 4-long contraction (one elementwise kernel + one
    # reduction) -- `torch.matmul` dispatches 32 768 4x4 products to a batched library GEMM: 0.52 ms per call in the
    # detect loop's kernel trace (profiles/r06_detect_kernel_stats.md)
    H = (fr.view(1, 1, 4, 4, 1) * H.unsqueeze(2)).sum(dim=3)
    valid = torch.arange(K, device=dev).view(1, K) < count.view(B, 1)
    H = torch.where(valid.view(B, K, 1, 1), H, torch.zeros_like(H))
    top = torch.where(valid, top, torch.zeros_like(top))
    sel = torch.where(valid, sel, torch.full_like(sel, -1))
    return H, top, sel, count


def importance_sampling(score, count, num_selected, generator=None, uniforms=None):
    """grasp_detector.py:237-251 on the device: `num_selected` draws per scene from the first
    count[b] poses with probability proportional to exp(5 score) (sorted uniforms against the
    cumulative sum = systematic inverse-CDF sampling).  Returns indices (B, num_selected) into the
    pose list (ascending); scenes with count <= num_selected keep 0..count-1 (padded with -1)."""
    B, K = score.shape
    dev = score.device
    valid = torch.arange(K, device=dev).view(1, K) < count.view(B, 1)
    wgt = torch.where(valid, torch.exp(5.0 * score.double()), torch.zeros((), dtype=torch.float64, device=dev))
    cum = torch.cumsum(wgt, dim=1)
    if uniforms is not None:      # the draws passed in (tests; the reference calls np.random.rand(num_selected) unseeded)
        u = torch.as_tensor(uniforms, dtype=torch.float64).to(dev).reshape(-1, num_selected).expand(B, -1)
    else:
        u = torch.rand((B, num_selected), generator=generator, device=dev, dtype=torch.float64)
    target = torch.sort(u, dim=1)[0] * cum[:, -1:]
    pick = torch.searchsorted(cum, target, right=False).clamp(max=K - 1)
    ar = torch.arange(num_selected, device=dev).view(1, -1).expand(B, -1)
    few = count.view(B, 1) <= num_selected
    pick = torch.where(few, torch.where(ar < count.view(B, 1), ar, torch.full_like(ar, -1)), pick)
    return pick


@dataclass
class GripperConfig:
    """configs/gripper_config.py:10-21 and processing_config.py:25,37-40."""
    half_bottom_width: float = 0.057
    bottom_length: float = 0.16
    finger_width: float = 0.023
    half_hand_thickness: float = 0.012
    finger_length: float = 0.09
    back_collision_margin: float = 0.0
    back_collision_threshold: float = 10 * math.sqrt(8)
    finger_collision_threshold: float = 10

    @property
    def half_bottom_space(self):
        return self.half_bottom_width - self.finger_width


def se3_inverse(poses):
    """`torch_batch_transformation_inv` (utils/math_utils.py:26-40) for (..., 4, 4) fp32 poses:
    [R^T | -R^T t], the form `GraspDetector.detect` feeds the collision check (grasp_detector.py:219)."""
    T = poses.float()
    Rt = T[..., :3, :3].transpose(-1, -2)
    out = torch.zeros_like(T)
    out[..., :3, :3] = Rt
    out[..., :3, 3:] = torch.matmul(-Rt, T[..., :3, 3:])
    out[..., 3, 3] = 1.0
    return out.contiguous()


def view_non_collision(poses, scene_points, gripper=None, inverse="general", count=None):
    """Batched `CloudCollisionChecker.view_non_collision`
    (cloud_processor/view_collision_checker.py:37-65) for all poses of all scenes
    in one launch.  poses (B,K,4,4) gripper->global frames; returns
    (ok (B,K) bool, counts (B,K,2) int32).  inverse="general": the inverse is taken in float64 and
    rounded to fp32 like the demo's caller (file_logger_cls.py:223-224); inverse="se3": the fp32
    analytic SE(3) inverse the detector uses (grasp_detector.py:219, `se3_inverse`).
    count (B,) int64 on the device (optional): only the first count[b] rows of scene b are poses (the padded best-first
    lists of `detect_poses`); the other rows are not scanned, read ok = False and zero counts."""
    gripper = gripper or GripperConfig()
    xyz = _F._f32c(scene_points, "scene_points")
    B, _, N = xyz.shape
    K = poses.shape[1]
    if inverse not in ("general", "se3"):
        raise ValueError("inverse must be 'general' or 'se3'")
    # inverse="se3": the kernel forms [R^T | -R^T t] itself (one launch; a batched library GEMM of 3x3 blocks took 0.26 ms)
    g2l = poses.float().contiguous() if inverse == "se3" else torch.linalg.inv(poses.double()).float().contiguous()
    counts = torch.empty((B, K, 2), dtype=torch.int32, device=xyz.device)
    params = (ctypes.c_float * 6)(gripper.finger_length, gripper.bottom_length,
                                  gripper.half_hand_thickness, gripper.half_bottom_width,
                                  gripper.half_bottom_space, gripper.back_collision_margin)
    with torch.cuda.device(xyz.device):
        if count is None and inverse != "se3":
            rc = _cabi.lib().s4g_collision_counts_f32(xyz.data_ptr(), g2l.data_ptr(), B, N, K, params,
                                                      counts.data_ptr(), _F._stream())
        else:
            cnt = None if count is None else count.to(device=xyz.device, dtype=torch.int64).contiguous()
            rc = _cabi.lib().s4g_collision_counts_n_f32(xyz.data_ptr(), g2l.data_ptr(), B, N, K, params,
                                                        None if cnt is None else cnt.data_ptr(),
                                                        1 if inverse == "se3" else 0, counts.data_ptr(), _F._stream())
    _cabi.check(rc, "collision_counts")
    ok = (counts[..., 0] <= gripper.back_collision_threshold) & \
         (counts[..., 1] <= gripper.finger_collision_threshold)
    if count is not None:
        ok = ok & (torch.arange(K, device=xyz.device).view(1, K) < count.view(B, 1))
    return ok, counts

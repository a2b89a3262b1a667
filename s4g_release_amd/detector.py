"""`GraspDetector.detect` composed on the device: clouds in, the selected grasp poses out
(reference `grasp_proposal/grasp_detector.py:187-254`; SURVEY.md section 8f rows f1-f3 composed).

Reference flow (one scene, host numpy between every stage, a host sync per stage and a Python loop with one launch
sequence + sync PER POSE in the collision check, `:216-234`):
    _pre_processing (:94-105)  ->  model (:206-209)  ->  post_processing (:137-185)  ->
    view_non_collision per pose (:213-234)  ->  importance sampling (:237-251)
Here every stage is a launch (or a handful) on the caller's stream over a BATCH of scenes, nothing reads a device value
on the host before the caller asks for the poses, and the whole call is capturable as one HIP graph
(`GraspDetector.graph`):
    subsample + REAL2TRAIN (one gather)  ->  FusedPointNet2(topk=K): score head on every point, pose heads on the K
    best-scoring points  ->  postprocess.detect_poses (thresholds, best-first order, decode, Gram-Schmidt, frame)  ->
    postprocess.view_non_collision for all K poses of all scenes in ONE launch against the full input cloud  ->
    stable compaction of the survivors  ->  postprocess.importance_sampling.

Pre-processing modes (`preprocess=`):
  "shipped"   what the reference EXECUTES: `CloudPreProcessor.voxelize` / `remove_outliers` call open3d and drop the
              returned clouds (cloud_processor.py:34,40 -- SURVEY.md Appendix D), so its network input is the (masked)
              raw cloud, REAL2TRAIN-transformed and randomly subsampled to NUM_INPUT points.  Sync-free: the subsample
              is a seeded permutation of a host-known size.
  "intended"  what the config constants describe: voxel down-sample (5 mm) and radius-outlier removal (32 points within
              2 cm) APPLIED (`preprocess.pre_processing`).  Their output sizes are data dependent, so this mode reads
              two counts per scene on the host (single-scene passes, `preprocess.py`); parity unpinned (open3d absent).

Order of the detections: each pose is built from its OWN point's rotation / translation, best expected score first --
`postprocess.detect_poses`' default; the reference's index mix-up (`:149-167`) is reproduced by
`detect_poses(reference_indexing=True)` on a full forward and pinned there, not here.
"""
import threading
from collections import OrderedDict

import numpy as np
import torch

from . import functions as _F
from . import postprocess as _post
from . import preprocess as _pre
from .fused import FusedPointNet2

TRAIN2REAL = _post.TRAIN2REAL


class Detections(tuple):
    """(poses (B, S, 4, 4) fp32 in the caller's frame, scores (B, S) fp32, count (B,) int64): scene b's first count[b]
    rows are its selected grasps (rows past the count are zero).  Also `.candidates` = (H, score, index, count) of
    every detection that survived the collision check, best first (what the importance sampling drew from)."""

    def __new__(cls, poses, scores, count, candidates=None):
        self = super().__new__(cls, (poses, scores, count))
        self.candidates = candidates
        return self


class GraspDetector:
    """Device counterpart of the reference class of the same name for `MODEL.TYPE = "PN2_CLS"`."""

    def __init__(self, net, precision=None, topk=2048, num_input=25600, camera2base=None,
                 vertical_direction=(0.0, 0.0, 1.0), seed=0, preprocess="shipped", sample_mode="random", gripper=None):
        """net: a `model.PointNet2` / reference `PointNet2` instance, or a ready `FusedPointNet2`.
        topk: points per scene whose pose heads are evaluated (the K best expected scores).  The result equals the
        full-forward detection whenever fewer than K points of a scene pass the score threshold (always, for a
        threshold of 0.7 on a trained network: the reference then keeps a few hundred).
        camera2base: (4, 4) camera -> robot base (the reference's `realworld.camera2base`, :155); identity if None.
        seed: of the subsample (scene b of a batch uses seed + b; every call may pass its own).  The reference draws
        unseeded (`np.random.choice`, :86-89); here the same cloud with the same seed gives the same detections."""
        self.run = net if isinstance(net, FusedPointNet2) else FusedPointNet2(net, precision=precision)
        if preprocess not in ("shipped", "intended"):
            raise ValueError("preprocess must be 'shipped' or 'intended'")
        self.topk, self.num_input, self.seed = int(topk), int(num_input), int(seed)
        self.preprocess, self.sample_mode = preprocess, sample_mode
        c2b = np.eye(4) if camera2base is None else np.asarray(camera2base, dtype=np.float64)
        t2r = np.asarray(TRAIN2REAL, dtype=np.float64)
        self.direction_matrix = tuple(map(tuple, (c2b[:3, :3] @ t2r[:3, :3]).tolist()))      # :155
        self.vertical_direction = tuple(float(v) for v in vertical_direction)                 # :80
        self.gripper = gripper or _post.GripperConfig()
        self._idx_cache = OrderedDict()
        self._lock = threading.Lock()
        self.stage_events = None

    # ---- stage 1: _pre_processing (:94-105)
    def _subsample_index(self, n, seeds, device):
        """(B, num_input) int64 on the device: scene b's seeded permutation of its n points (`sample_single_cloud`,
        :82-92).  Built on the host from a HOST-known n, cached per (n, seed): no device value is read."""
        key = (int(n), tuple(seeds), str(device))
        with self._lock:
            hit = self._idx_cache.get(key)
            if hit is not None:
                self._idx_cache.move_to_end(key)
                return hit
        idx = np.stack([_pre.sample_indices(int(n), self.num_input, s) for s in seeds]).astype(np.int64)
        t = torch.from_numpy(idx).pin_memory().to(device, non_blocking=True)
        with self._lock:
            self._idx_cache[key] = t
            while len(self._idx_cache) > 8:
                self._idx_cache.popitem(last=False)
        return t

    def pre_processing(self, cloud, seed=None):
        """cloud (B, 3, n) fp32 on the device, REAL frame -> scene_points (B, 3, num_input), TRAIN frame."""
        seed = self.seed if seed is None else int(seed)
        B, _, n = cloud.shape
        if self.preprocess == "intended":
            pts = [_pre.pre_processing(cloud[b], self.num_input, seed + b, self.sample_mode)[0] for b in range(B)]
            return torch.stack(pts).contiguous()
        if n <= 0:
            raise RuntimeError("cannot sample from an empty cloud")
        if self.sample_mode == "fps" and n > self.num_input:
            idx = _F.farthest_point_sample(cloud, self.num_input)
        else:
            idx = self._subsample_index(n, [seed + b for b in range(B)], cloud.device)
        g = _F.gather_points(cloud, idx)                  # (B, 3, num_input): the operator the samplers use
        # REAL2TRAIN (:26) is a signed permutation: (x, y, z) -> (y, x, -z), exact in fp32
        return torch.stack([g[:, 1], g[:, 0], -g[:, 2]], dim=1)

    # ---- the whole call
    def _mark(self, name):
        if self.stage_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.stage_events.append((name, ev))

    def submit(self, cloud, num_selected=5, score_threshold=0.7, verticalness_threshold=0.2,
               collision_check=True, seed=None, uniforms=None, generator=None, collision_cloud=None):
        """Enqueue pre-processing and the forward pass of one batch and return a handle without waiting; `.result()`
        enqueues the decode / collision check / sampling behind it.  Submitting batch i + 1 before collecting batch i
        lets its FPS chain run underneath batch i's contractions (`FusedPointNet2.submit`)."""
        cloud = _F._f32c(cloud, "cloud")
        if cloud.dim() != 3 or cloud.size(1) != 3:
            raise RuntimeError("cloud must be (B, 3, n)")
        collision_cloud = cloud if collision_cloud is None else _F._f32c(collision_cloud, "collision_cloud")
        self._mark("start")
        pts = self.pre_processing(cloud, seed)
        self._mark("pre_processing")
        h = self.run.submit({"scene_points": pts}, topk=self.topk)
        return _Pending(self, h, pts, collision_cloud, num_selected, score_threshold, verticalness_threshold,
                        collision_check, uniforms, generator)

    def detect_device(self, cloud, num_selected=5, score_threshold=0.7, verticalness_threshold=0.2,
                      collision_check=True, seed=None, uniforms=None, generator=None, collision_cloud=None):
        """cloud (B, 3, n) fp32 on the device, REAL frame: what the network sees (the masked `target_cloud` of :199-203);
        collision_cloud (B, 3, n_all): what the collision check sees (the whole `cloud_array`, :220-222; default: cloud)
        -> `Detections`.  Nothing here waits for the device."""
        return self.submit(cloud, num_selected, score_threshold, verticalness_threshold, collision_check, seed, uniforms,
                           generator, collision_cloud).result()

    def _finish(self, pred, pts, collision_cloud, num_selected, score_threshold, verticalness_threshold, collision_check,
                uniforms, generator):
        self._mark("prediction")
        H, score, index, count = _post.detect_poses(
            pred, pts, score_threshold, verticalness_threshold, direction_matrix=self.direction_matrix,
            vertical_direction=self.vertical_direction, frame=TRAIN2REAL, max_poses=self.topk)
        self._mark("post_processing")
        B, K = score.shape
        dev = score.device
        if collision_check:
            # :213-234 -- against the WHOLE input cloud (`cloud_array`, not the subsample), analytic SE(3) inverse
            ok, _ = _post.view_non_collision(H, collision_cloud, self.gripper, inverse="se3", count=count)
            order = torch.sort((~ok).to(torch.uint8), dim=1, stable=True)[1]       # survivors first, order kept
            H = torch.gather(H, 1, order.view(B, K, 1, 1).expand(-1, -1, 4, 4))
            score = torch.gather(score, 1, order)
            index = torch.gather(index, 1, order)
            count = ok.sum(dim=1)
            live = torch.arange(K, device=dev).view(1, K) < count.view(B, 1)
            H = torch.where(live.view(B, K, 1, 1), H, torch.zeros_like(H))
            score = torch.where(live, score, torch.zeros_like(score))
            index = torch.where(live, index, torch.full_like(index, -1))
            self._mark("collision_check")
        pick = _post.importance_sampling(score, count, num_selected, generator=generator, uniforms=uniforms)   # :237-251
        safe = pick.clamp(min=0)
        have = pick >= 0
        poses = torch.gather(H, 1, safe.view(B, -1, 1, 1).expand(-1, -1, 4, 4))
        poses = torch.where(have.view(B, -1, 1, 1), poses, torch.zeros_like(poses))
        scores = torch.where(have, torch.gather(score, 1, safe), torch.zeros_like(score[:, :1]).expand_as(pick))
        self._mark("importance_sampling")
        return Detections(poses, scores, have.sum(dim=1), candidates=(H, score, index, count))

    def detect(self, cloud_array, cloud_mask=None, num_selected=5, score_threshold=0.7, verticalness_threshold=0.2,
               collision_check=True, seed=None, uniforms=None):
        """The reference's signature (:187-188): cloud_array (n, 3) or (3, n) numpy / tensor (or a (B, 3, n) batch),
        optional boolean cloud_mask (n,) -> (poses, scores) of the selected grasps; for a single scene they are
        trimmed to the detections found, like the reference's return value (this trim reads the count: the one
        host synchronisation of the call)."""
        cloud = torch.as_tensor(cloud_array, dtype=torch.float32)
        single = cloud.dim() == 2
        if single:
            assert cloud.shape[0] == 3 or cloud.shape[1] == 3, \
                "input should have shape (n, 3) or (3, n), but given {}".format(tuple(cloud.shape))     # :191-192
            if cloud.shape[1] == 3:
                cloud = cloud.t()                                                                         # :193-194
            cloud = cloud.unsqueeze(0)
        dev = torch.device("cuda", torch.cuda.current_device())
        cloud = cloud.to(dev).contiguous()
        target = cloud
        if isinstance(cloud_mask, (np.ndarray, torch.Tensor)):                                            # :196-199
            keep = np.nonzero(np.asarray(torch.as_tensor(cloud_mask).cpu(), dtype=bool))[0]               # host-known mask
            target = cloud.index_select(2, torch.from_numpy(keep).to(dev)).contiguous()
        out = self.detect_device(target, num_selected, score_threshold, verticalness_threshold, collision_check,
                                 seed, uniforms, collision_cloud=cloud)
        if not single:
            return out
        n = int(out[2][0])
        return out[0][0, :n], out[1][0, :n]

    def stage_ms(self):
        """Milliseconds per stage of the last `detect_device` call when `stage_events` was a list (synchronises)."""
        torch.cuda.synchronize()
        ev = self.stage_events or []
        return {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(ev[:-1], ev[1:])}

    def graph(self, example_cloud, **kw):
        """Record one `detect_device` call of `example_cloud`'s shape as a HIP graph: see `GraphedDetect`."""
        return GraphedDetect(self, example_cloud, kw)


class _Pending:
    """An in-flight `GraspDetector.submit`."""

    def __init__(self, det, handle, pts, *rest):
        self.det, self.handle, self.pts, self.rest = det, handle, pts, rest

    def result(self):
        return self.det._finish(self.handle.result(), self.pts, *self.rest)


class GraphedDetect:
    """`GraspDetector.detect_device` for a fixed cloud shape and fixed thresholds as ONE HIP graph: pre-processing,
    the ~45 launches of the forward, decode, collision check and sampling replayed by a single host call (the
    one-scene serving latency).  The returned tensors are the graph's static outputs (valid until the next replay)."""

    def __init__(self, det, example, kw):
        cloud = _F._f32c(example, "cloud")
        self.static_in = cloud.clone()
        dev = cloud.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up off the capture: constants, allocator pools, index cache
            for _ in range(2):
                det.detect_device(self.static_in, **kw)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        was, ev = _F.OpTimer.enabled, det.stage_events
        _F.OpTimer.enabled, det.stage_events = False, None      # event pairs with timing cannot be captured
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self.static_out = det.detect_device(self.static_in, **kw)
        finally:
            _F.OpTimer.enabled, det.stage_events = was, ev

    def __call__(self, cloud):
        if tuple(cloud.shape) != tuple(self.static_in.shape):
            raise RuntimeError("graph was recorded for clouds of shape %s" % (tuple(self.static_in.shape),))
        if cloud.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(cloud, non_blocking=True)
        self.graph.replay()
        return self.static_out

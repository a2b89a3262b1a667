"""Multi-GPU data path: scenes shard embarrassingly, outputs are all-gathered.

The reference's only multi-device code is a never-active `nn.DataParallel`
(`grasp_proposal_test.py:52-53`, batch size asserted to 1).  Here: one process
per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo"
for the CPU functional tests), rank g of G owns scenes [g*B/G, (g+1)*B/G); the
forward pass has no cross-scene dependency (BatchNorm runs on running stats),
so the only collective is ONE `all_gather_into_tensor` per batch on the packed
per-point head outputs (B_local, 21, N) -- 34 MB/rank at 16 scenes.
"""
import os

import torch
import torch.distributed as dist

HEADS = ("score", "frame_R", "frame_t", "movable_logits")


RCCL_MAX_CHANNELS = "8"


def bound_rccl_channels():
    """Bound the CUs RCCL's collectives can occupy beside the contraction kernels (call before the process group
    exists): NCCL_MAX_NCHANNELS defaults to 8 here unless the caller's environment already sets it.  A channel is
    one workgroup; the contraction launches run one or two workgroups per CU with the LDS full, so a channel that
    lands on a CU holds that slot for the collective's duration (the same mechanism through which the FPS
    workgroups cost the step 1.6 x their CU-time share, profiles/r04_geometry_cost.md).  The per-batch all-gather
    (34 MB per rank out, 238 MB in at 8 ranks) needs ~0.3 ms of the seven xGMI links' time per 7.6 ms step; at 8
    channels it is still several times shorter than a step and overlapped on its own stream."""
    os.environ.setdefault("NCCL_MAX_NCHANNELS", RCCL_MAX_CHANNELS)
    return os.environ["NCCL_MAX_NCHANNELS"]


def init_from_env(backend=None, bound_channels=False):
    """Initialise the default process group from torchrun's environment.
    Returns (rank, world, local_rank); a no-op world of 1 without WORLD_SIZE.
    bound_channels: call `bound_rccl_channels()` first (an inference job whose only collective is the per-batch
    all-gather; NOT the default: the bound is process-wide and would also throttle a training job's all-reduce)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            if bound_channels:
                bound_rccl_channels()
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, **kw)
    return rank, world, local_rank


def shard_range(total, rank, world):
    """Contiguous block partition; the first `total % world` ranks get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_outputs(pred):
    """dict of (B, C_h, N) -> one contiguous (B, sum C_h, N) tensor + channel splits.  The fast path's
    `fused.PackedPred` already IS that tensor (the heads launch wrote the four outputs as its channel
    slices): it is returned as it stands, no copy; any other dict is concatenated."""
    if hasattr(pred, "get") and pred.get("index") is not None:
        # FusedPointNet2(..., topk=K): the (B, 21, K) tensor covers a scene's K kept points only and "index" says which --
        # gathered without it the per-point channels would no longer say which scene points they belong to
        raise ValueError("predictions over kept points (topk=) carry an 'index' entry that the head gather would drop: "
                         "gather decoded poses instead (OutputGather('poses')), or gather pred['index'] alongside")
    chans = [pred[k].shape[1] for k in HEADS]
    packed = getattr(pred, "packed", None)
    if (packed is not None and packed.is_contiguous() and packed.shape[1] == sum(chans) and
            all(pred[k].data_ptr() == packed.data_ptr() + packed.element_size() * packed.shape[2] * c0
                for k, c0 in zip(HEADS, [sum(chans[:i]) for i in range(len(chans))]))):
        return packed, chans
    return torch.cat([pred[k] for k in HEADS], dim=1).contiguous(), chans


def unpack_outputs(packed, chans):
    out, c0 = {}, 0
    for k, c in zip(HEADS, chans):
        out[k] = packed[:, c0:c0 + c]
        c0 += c
    return out


def all_gather_outputs(pred, group=None):
    """All-gather the head outputs of equally sized shards.

    Every rank passes its local dict (B_local, C, N); every rank receives the
    dict for all world*B_local scenes in rank order.  One collective."""
    packed, chans = pack_outputs(pred)
    if not dist.is_initialized():
        return unpack_outputs(packed, chans)
    world = dist.get_world_size(group)   # a 1-rank group still takes the collective
    gathered = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]),
                           dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    return unpack_outputs(gathered, chans)


def sharded_forward(runner, scene_points, group=None):
    """Run `runner` on this rank's block of `scene_points` (B_total, 3, N) and
    return the all-gathered outputs for the whole batch (B_total must divide
    evenly over the ranks so the gather is a single fixed-size collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    total = scene_points.shape[0]
    scenes_per_rank(total, world)        # raises for an uneven split
    lo, hi = shard_range(total, rank, world)
    pred = runner({"scene_points": scene_points[lo:hi]})
    return all_gather_outputs(pred, group)


def all_gather_poses(H, score, index, group=None):
    """All-gather decoded grasp frames instead of per-point heads (row f1):
    per scene K x (16 + 1 + 1) values -- 3.6 KB at K = 50 against 2.15 MB of
    21-channel outputs.  One collective on a packed (B_local, K, 18) fp32 tensor
    (the int64 point index rides along as exactly representable fp32 < 2^24)."""
    B, K = score.shape
    packed = torch.cat([H.reshape(B, K, 16), score.unsqueeze(-1), index.to(H.dtype).unsqueeze(-1)],
                       dim=2).contiguous()
    if dist.is_initialized():            # a 1-rank group still takes the collective
        world = dist.get_world_size(group)
        out = torch.empty((world * B, K, 18), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(out, packed, group=group)
        packed = out
    return (packed[..., :16].reshape(-1, K, 4, 4), packed[..., 16], packed[..., 17].long())


class OutputGather:
    """The per-batch collective of the multi-GPU path, issued on its OWN stream.

    mode "heads": one `all_gather_into_tensor` of the packed (B_local, 21, N) head outputs (34 MB per
    rank at 16 scenes of 25 600 points); mode "poses": the K best grasp frames per scene are decoded
    on the device first (`decode`, e.g. `postprocess.decode_top_poses`) and one collective moves the
    packed (B_local, K, 18) tensor instead -- 3.6 KB per scene at K = 50.

    On a HIP device the collective is enqueued on a side stream that waits for the producer's
    current stream through an event; the caller's stream then waits for the gather's event, so the
    next batch's contractions (other streams) overlap it by construction and not by accident of the
    collective library's internal stream.  `last_stream` names where the last gather ran (bench.py
    records it).  On CPU tensors (gloo functional tests) there are no streams and the call is the
    plain collective."""

    def __init__(self, mode="heads", decode=None, group=None, device=None):
        if mode not in ("heads", "poses"):
            raise ValueError("mode must be 'heads' or 'poses'")
        if mode == "poses" and decode is None:
            raise ValueError("mode 'poses' needs a decode(pred, scene_points) -> (H, score, index)")
        self.mode, self.decode, self.group = mode, decode, group
        self.stream = torch.cuda.Stream(device=device) if device is not None and device.type == "cuda" else None
        self.last_stream = "caller"
        self.payload_bytes = 0

    def local_payload(self, pred, scene_points=None):
        """This rank's contribution as a list of tensors, in the order of `as_list(gathered)`."""
        if self.mode == "poses":
            return list(self.decode(pred, scene_points))
        return [pred[k] for k in HEADS]

    @staticmethod
    def as_list(gathered):
        return [gathered[k] for k in HEADS] if isinstance(gathered, dict) else list(gathered)

    def _collect(self, pred, scene_points):
        if self.mode == "poses":
            H, score, index = self.decode(pred, scene_points)
            self.payload_bytes = H.shape[0] * H.shape[1] * 18 * 4
            return all_gather_poses(H, score, index, self.group)
        self.payload_bytes = sum(pred[k].numel() * pred[k].element_size() for k in HEADS)
        return all_gather_outputs(pred, self.group)

    def __call__(self, pred, scene_points=None):
        if self.stream is None:
            return self._collect(pred, scene_points)
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_event(cur.record_event())
        with torch.cuda.stream(self.stream):
            out = self._collect(pred, scene_points)
            done = self.stream.record_event()
        for t in list(pred.values()) + ([scene_points] if scene_points is not None else []):
            t.record_stream(self.stream)
        cur.wait_event(done)
        for t in (out.values() if isinstance(out, dict) else out):
            t.record_stream(cur)
        self.last_stream = "side stream %#x" % self.stream.cuda_stream
        return out


def scenes_per_rank(global_batch, world):
    """Scenes each rank runs when a GLOBAL batch is given: it must divide evenly (the gather is one
    fixed-size collective); bench.py's --global-batch goes through here."""
    if global_batch <= 0 or world <= 0:
        raise ValueError("global batch and world size must be positive")
    if global_batch % world != 0:
        raise ValueError("batch of %d scenes does not divide over %d ranks" % (global_batch, world))
    return global_batch // world


def shard_report(scene_ids, global_batch, device_name="", group=None):
    """What the first real multi-GPU run needs to verify itself (BASELINE.json configs[3]: 128 scenes
    over 8 ranks): every rank contributes (rank, its scene ids, its device) through ONE
    `all_gather_object`; every rank gets the world's table back in rank order together with the
    communicator size the collective library reports, after checking that the ranks' scene ranges
    tile [0, global_batch) exactly once.  A single process returns its own row without a collective."""
    mine = {"rank": dist.get_rank(group) if dist.is_initialized() else 0,
            "scenes": [int(scene_ids[0]), int(scene_ids[-1]) + 1] if len(scene_ids) else [0, 0],
            "n_scenes": len(scene_ids), "device": device_name}
    if list(scene_ids) != list(range(mine["scenes"][0], mine["scenes"][1])):
        raise ValueError("a rank's scenes must be one contiguous block")
    if dist.is_initialized():
        comm = dist.get_world_size(group)
        rows = [None] * comm
        dist.all_gather_object(rows, mine, group=group)
        backend = dist.get_backend(group)
    else:
        comm, rows, backend = 1, [mine], None
    if any(r is None for r in rows):
        raise RuntimeError("shard_report: all_gather_object returned %d of %d rows"
                           % (sum(r is not None for r in rows), comm))
    rows = sorted(rows, key=lambda r: r["rank"])
    if [r["rank"] for r in rows] != list(range(comm)):
        raise RuntimeError("shard_report: ranks %s of a communicator of %d" % ([r["rank"] for r in rows], comm))
    edge = 0
    for r in rows:
        if r["scenes"][0] != edge:
            raise RuntimeError("shard_report: rank %d starts at scene %d, expected %d" % (r["rank"], r["scenes"][0], edge))
        edge = r["scenes"][1]
    if edge != global_batch:
        raise RuntimeError("shard_report: the ranks cover %d scenes of a global batch of %d" % (edge, global_batch))
    # `world`: the launcher's count (torchrun's WORLD_SIZE; 1 without a launcher); `communicator_size`: what
    # the collective library's communicator reports; `rows_gathered`: the rows the collective actually
    # returned.  Three sources, so their agreement (asserted by bench.py's tests) says something.
    return {"world": int(os.environ.get("WORLD_SIZE", "1")) if group is None else comm,
            "communicator_size": comm, "rows_gathered": len(rows), "backend": backend,
            "global_batch": int(global_batch), "per_rank": rows}


def _bits_checksum(tensors):
    """Order-free exact checksum of a list of tensors: the int64 sum of their 32-bit patterns (int64 tensors:
    of their values).  Equal data -> equal checksum whatever kernel or blocking summed it."""
    total = 0
    for t in tensors:
        t = t.contiguous()
        v = t if t.dtype == torch.int64 else t.view(torch.int32).to(torch.int64)
        total += int(v.sum().item())
    return total & 0xFFFFFFFFFFFFFFFF


def gather_check(gather, pred, scene_points=None, group=None):
    """Verify ONE gathered batch on every rank (bench.py runs it before the timed region; world-2 / -8 gloo tests
    run it in the container): (1) this rank's block of the gathered tensors equals what the rank computed, bit for
    bit; (2) block r carries the checksum rank r announced for its own payload (`all_gather_object`), for every r --
    i.e. the collective put every rank's data where `unpack` expects it.  Raises RuntimeError on a mismatch,
    returns a summary for the bench line."""
    local = gather.local_payload(pred, scene_points)
    out = gather.as_list(gather(pred, scene_points))
    if group is None:
        group = getattr(gather, "group", None)      # the sub-group the gather itself runs on
    if dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    per = local[0].shape[0]
    for lt, gt in zip(local, out):
        if gt.shape[0] != world * per:
            raise RuntimeError("gather_check: gathered %d rows for %d ranks x %d" % (gt.shape[0], world, per))
        if not torch.equal(gt[rank * per:(rank + 1) * per], lt):
            raise RuntimeError("gather_check: rank %d's own block changed in the all-gather" % rank)
    mine = _bits_checksum(local)
    sums = [mine]
    if dist.is_initialized():
        sums = [None] * world
        dist.all_gather_object(sums, mine, group=group)
    for r in range(world):
        got = _bits_checksum([gt[r * per:(r + 1) * per] for gt in out])
        if got != sums[r]:
            raise RuntimeError("gather_check: block %d of the gathered batch does not carry rank %d's data "
                               "(checksum %x, announced %x)" % (r, r, got, sums[r]))
    return {"blocks_verified": world, "own_block_bit_identical": True, "scenes_per_block": int(per),
            "payload": gather.mode, "checksum_rank0": "%016x" % sums[0]}

"""world_size-2 functional tests of the multi-GPU path on CPU (gloo): scene
sharding and the single all-gather of the packed head outputs.  The forward
itself is replaced by a deterministic stand-in (the HIP kernels need a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from s4g_release_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_runner(batch):
    x = batch["scene_points"]                      # (B,3,N)
    base = x[:, 0:1] * 2.0 + x[:, 1:2] - x[:, 2:3]   # (B,1,N), purely elementwise
    return {"score": base.repeat(1, 3, 1) + 1.0, "frame_R": base.repeat(1, 9, 1) * 2.0,
            "frame_t": base.repeat(1, 4, 1) - 3.0, "movable_logits": base.repeat(1, 5, 1) * 0.5}


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(total, 3, 50, generator=g)
    out = sdist.sharded_forward(_fake_runner, pts)
    ref = _fake_runner({"scene_points": pts})
    ok = all(torch.equal(out[k], ref[k]) for k in sdist.HEADS)
    shapes = {k: tuple(v.shape) for k, v in out.items()}
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, shapes))


def test_shard_range_partitions():
    for total in (1, 7, 16, 128):
        for world in (1, 2, 3, 8):
            spans = [sdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    pred = _fake_runner({"scene_points": torch.randn(3, 3, 11)})
    packed, chans = sdist.pack_outputs(pred)
    assert packed.shape == (3, 21, 11) and chans == [3, 9, 4, 5]
    back = sdist.unpack_outputs(packed, chans)
    assert all(torch.equal(back[k], pred[k]) for k in pred)
    single = sdist.all_gather_outputs(pred)      # world of 1: identity
    assert all(torch.equal(single[k], pred[k]) for k in pred)


def _run_world(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results)


@pytest.mark.parametrize("world,total", [(2, 6), (4, 8)])
def test_sharded_forward_all_gather_gloo(world, total):
    for rank, ok, shapes in _run_world(_worker, world, total):
        assert ok, rank
        assert shapes["score"] == (total, 3, 50) and shapes["movable_logits"] == (total, 5, 50)


def _fake_decode(pred, xyz, K=5):
    """Shapes and dtypes of postprocess.decode_top_poses: (B,K,4,4) fp32, (B,K) fp32, (B,K) int64 --
    a deterministic function of the scene so that every rank can check every other rank's rows."""
    B, _, N = xyz.shape
    score, index = torch.topk(pred["score"][:, 0], K, dim=1)
    H = torch.eye(4).repeat(B, K, 1, 1)
    H[:, :, :3, 3] = torch.gather(xyz, 2, index.unsqueeze(1).expand(B, 3, K)).transpose(1, 2)
    return H, score, index


def _gather_worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sdist.init_from_env(backend="gloo")
    per = 3
    g = torch.Generator().manual_seed(1)
    pts = torch.randn(world * per, 3, 40, generator=g)
    lo, hi = sdist.shard_range(world * per, rank, world)
    mine = pts[lo:hi]
    gather = sdist.OutputGather(mode, decode=_fake_decode if mode == "poses" else None, device=mine.device)
    out = gather(_fake_runner({"scene_points": mine}), mine)
    if mode == "poses":
        H, s, i = out
        rH, rs, ri = _fake_decode(_fake_runner({"scene_points": pts}), pts)
        ok = torch.equal(H, rH) and torch.equal(s, rs) and torch.equal(i, ri) and i.dtype == torch.int64
        nbytes = gather.payload_bytes == per * 5 * 18 * 4
    else:
        ref = _fake_runner({"scene_points": pts})
        ok = all(torch.equal(out[k], ref[k]) for k in sdist.HEADS)
        nbytes = gather.payload_bytes == per * 21 * 40 * 4
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok and nbytes), gather.last_stream))


@pytest.mark.parametrize("mode", ["heads", "poses"])
def test_output_gather_modes_gloo_world2(mode):
    """bench.py's collective object: packed heads, or decoded top-K poses (the payload reduction of
    SURVEY 8e / 8f1) with a real decode-shaped payload; on CPU tensors it runs on the caller."""
    for rank, ok, where in _run_world(_gather_worker, 2, mode):
        assert ok, rank
        assert where == "caller"


def _configs3_worker(rank, world, port, global_batch, q):
    """BASELINE.json configs[3] in miniature: `--global-batch 128` over 8 ranks, bench.py's own scene
    assignment (rank g runs scenes [g B, (g + 1) B)), the self-verifying shard table, one all-gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sdist.init_from_env(backend="gloo")
    B = sdist.scenes_per_rank(global_batch, world)
    scene_ids = [rank * B + i for i in range(B)]          # bench.py: scene_ids
    rep = sdist.shard_report(scene_ids, global_batch, device_name="cpu:%d" % rank)
    g = torch.Generator().manual_seed(3)
    pts = torch.randn(global_batch, 3, 8, generator=g)
    out = sdist.OutputGather("heads", device=pts.device)(_fake_runner({"scene_points": pts[scene_ids]}), None)
    ref = _fake_runner({"scene_points": pts})
    ok = all(torch.equal(out[k], ref[k]) for k in sdist.HEADS)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, rep))


def test_configs3_global_batch_128_over_8_ranks_gloo():
    for rank, ok, rep in _run_world(_configs3_worker, 8, 128):
        assert ok, rank
        assert rep["world"] == 8 and rep["communicator_size"] == 8 and rep["backend"] == "gloo"
        assert rep["global_batch"] == 128
        assert [r["scenes"] for r in rep["per_rank"]] == [[16 * g, 16 * g + 16] for g in range(8)]
        assert [r["device"] for r in rep["per_rank"]] == ["cpu:%d" % g for g in range(8)]


def test_shard_report_single_process_and_bad_ranges():
    rep = sdist.shard_report(list(range(16)), 16, "cuda:0")
    assert rep["world"] == 1 and rep["per_rank"][0]["scenes"] == [0, 16] and rep["backend"] is None
    with pytest.raises(RuntimeError):
        sdist.shard_report(list(range(16)), 32)            # the ranks do not cover the global batch
    with pytest.raises(ValueError):
        sdist.shard_report([0, 2, 3], 3)                   # not one contiguous block


def test_output_gather_argument_errors():
    with pytest.raises(ValueError):
        sdist.OutputGather("poses")              # no decode function
    with pytest.raises(ValueError):
        sdist.OutputGather("everything")
    assert sdist.scenes_per_rank(128, 8) == 16
    with pytest.raises(ValueError):
        sdist.scenes_per_rank(6, 4)


def test_bench_rejects_an_uneven_global_batch_before_touching_a_gpu():
    """`bench.py --global-batch` through the script's own argument path: 6 scenes over 4 ranks must
    stop with exit code 2 and a message (no GPU is needed to get there)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--global-batch", "6"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 2 and "does not divide" in out.stderr and out.stdout.strip() == ""


def test_all_gather_poses_single_process_roundtrip():
    H = torch.randn(3, 7, 4, 4)
    score = torch.rand(3, 7)
    index = torch.randint(0, 25600, (3, 7))
    h2, s2, i2 = sdist.all_gather_poses(H, score, index)
    assert torch.equal(h2, H) and torch.equal(s2, score) and torch.equal(i2, index)


def test_uneven_batch_is_rejected():
    with pytest.raises(ValueError):
        class _G:   # fake a 4-rank world without initialising a process group
            pass
        import unittest.mock as um
        with um.patch.object(dist, "is_initialized", return_value=True), \
                um.patch.object(dist, "get_world_size", return_value=4), \
                um.patch.object(dist, "get_rank", return_value=0):
            sdist.sharded_forward(_fake_runner, torch.randn(6, 3, 5))


def test_pack_outputs_hands_a_packed_pred_over_without_a_copy():
    """The fast path's heads launch writes the four outputs as channel slices of one (B, 21, N) tensor
    (`fused.PackedPred`): `pack_outputs` must return that tensor itself; any other dict is concatenated."""
    from s4g_release_amd.fused import PackedPred
    packed = torch.randn(3, 21, 11)
    pred = PackedPred(zip(sdist.HEADS, packed.split([3, 9, 4, 5], dim=1)), packed=packed)
    out, chans = sdist.pack_outputs(pred)
    assert out.data_ptr() == packed.data_ptr() and chans == [3, 9, 4, 5]
    plain = {k: v.clone() for k, v in pred.items()}
    out2, _ = sdist.pack_outputs(plain)
    assert out2.data_ptr() != packed.data_ptr() and torch.equal(out2, packed)
    # a PackedPred whose slices were replaced is not trusted
    pred["score"] = pred["score"].clone()
    out3, _ = sdist.pack_outputs(pred)
    assert out3.data_ptr() != packed.data_ptr() and torch.equal(out3, packed)
    # predictions over kept points (topk=) carry "index": the head gather would drop it -> refused, not silently lossy
    kept = PackedPred(zip(sdist.HEADS, packed.split([3, 9, 4, 5], dim=1)), packed=packed)
    kept["index"] = torch.arange(11).expand(3, 11)
    with pytest.raises(ValueError):
        sdist.pack_outputs(kept)
    with pytest.raises(ValueError):
        sdist.all_gather_outputs(kept)


def test_gather_check_passes_a_faithful_gather_and_refuses_a_corrupted_one():
    pred = _fake_runner({"scene_points": torch.randn(4, 3, 9)})
    g = sdist.OutputGather("heads")
    rep = sdist.gather_check(g, pred)
    assert rep["blocks_verified"] == 1 and rep["own_block_bit_identical"] and rep["scenes_per_block"] == 4

    class Corrupt(sdist.OutputGather):
        def __call__(self, pred, scene_points=None):
            out = super().__call__(pred, scene_points)
            out = {k: v.clone() for k, v in out.items()}
            out["frame_t"][2, 1, 3] += 1.0
            return out
    with pytest.raises(RuntimeError, match="own block changed"):
        sdist.gather_check(Corrupt("heads"), pred)


def _gather_check_worker(rank, world, port, swap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(100 + rank)
    pts = torch.randn(2, 3, 17, generator=g)
    pred = _fake_runner({"scene_points": pts})

    class Swapped(sdist.OutputGather):       # a collective that delivers the OTHER ranks' blocks rotated
        def __call__(self, pred, scene_points=None):
            out = super().__call__(pred, scene_points)
            per = pred["score"].shape[0]
            res = {}
            for k, v in out.items():
                blocks = list(v.split(per, dim=0))
                me = dist.get_rank()
                others = [i for i in range(len(blocks)) if i != me]
                rot = others[1:] + others[:1]
                new = list(blocks)
                for dst, src in zip(others, rot):
                    new[dst] = blocks[src]
                res[k] = torch.cat(new, dim=0)
            return res
    try:
        rep = sdist.gather_check((Swapped if swap else sdist.OutputGather)("heads"), pred)
        res = ("ok", rep["blocks_verified"])
    except RuntimeError as e:
        res = ("refused", str(e)[:120])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res))


def test_gather_check_world3_catches_blocks_delivered_in_the_wrong_place():
    assert [r[1] for r in _run_world(_gather_check_worker, 3, False)] == [("ok", 3)] * 3
    res = _run_world(_gather_check_worker, 3, True)
    assert all(r[1][0] == "refused" and "does not carry rank" in r[1][1] for r in res), res

"""bench.py's N > 1 control flow, END TO END, in the container: `python bench.py --gpus W`
self-launches its real `torch.distributed.run` child with W ranks on gloo (S4G_BENCH_BACKEND=gloo: CPU tensors,
`tests/bench_stub.StubRunner` in place of the network) and must run shard_report -> gather check -> fenced timed
region -> all_reduce(MAX) -> ranks != 0 destroy and leave -> rank 0 prints EXACTLY one JSON line with n_gpus = W,
a communicator of W, W contiguous scene ranges and every rank's gathered block verified.  The reference's only
multi-device site is `inference/grasp_proposal/grasp_proposal_test.py:52-53`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "S4G_BENCH_FORCE_DIST")}
    env.update(S4G_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def _one_line(proc):
    assert proc.returncode == 0, proc.stderr[-4000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]          # rank 0's line and nothing else on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_runs_end_to_end_at_world_n_on_gloo(world):
    # (the pose payload's decode is a HIP kernel: `--gather poses` is covered on gloo by tests/test_dist.py with
    #  decode-shaped tensors and on RCCL by tests/test_bench_gpu.py)
    steps, warm, batch, points = 3, 1, 16, 256
    argv = ["--gpus", str(world), "--steps", str(steps), "--warmup", str(warm), "--batch", str(batch),
            "--points", str(points)]
    line = _one_line(_bench(argv))
    assert "test_mode" in line and "NOT a measurement" in line["test_mode"]
    assert line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warm
    assert line["scaling"] == "weak" and line["unit"] == "scenes/sec" and line["higher_is_better"] is True
    assert line["config"]["global_batch"] == world * batch and line["config"]["scenes_per_gpu"] == batch
    # value = all ranks' scenes / the slowest rank's time
    assert abs(line["value"] - world * batch * steps / (line["ms_per_step"] * steps / 1e3)) < 0.01 * line["value"]
    d = line["distributed"]
    assert d["world"] == world and d["communicator_size"] == world and d["rows_gathered"] == world
    assert d["backend"] == "gloo" and d["global_batch"] == world * batch
    assert [r["rank"] for r in d["per_rank"]] == list(range(world))
    assert [r["scenes"] for r in d["per_rank"]] == [[r * batch, (r + 1) * batch] for r in range(world)]
    assert len({r["device"] for r in d["per_rank"]}) == world          # one device per rank
    g = d["gather_check"]
    assert g["blocks_verified"] == world and g["own_block_bit_identical"] and g["scenes_per_block"] == batch
    c = line["collective"]
    assert c["op"].startswith("all_gather_into_tensor") and c["payload"] == "heads"
    assert c["payload_bytes_per_rank_per_step"] == batch * 21 * points * 4
    assert line["step_ms"]["n"] == steps - 1
    # rank 0's single-GPU legs do not run at N > 1 (the other ranks have left)
    assert line["latency"] is None and line["configs4"] is None and line["cpu_baseline"] is None
    assert line["roofline"]["frac"] is None        # nothing of the network ran: no kernel to price


def test_bench_configs3_layout_global_batch_128_over_8_ranks():
    """BASELINE.json configs[3]: 128 scenes over 8 ranks through --global-batch."""
    line = _one_line(_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--global-batch", "128",
                             "--points", "128"]))
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 128 and line["config"]["scenes_per_gpu"] == 16
    assert [r["scenes"] for r in line["distributed"]["per_rank"]] == [[16 * r, 16 * r + 16] for r in range(8)]
    assert line["distributed"]["gather_check"]["blocks_verified"] == 8


def test_bench_under_the_drivers_torchrun_line():
    """The driver's own launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` (no self-launch involved)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(S4G_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                           "--gpus", "4", "--steps", "2", "--warmup", "1", "--points", "128"],
                          env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-4000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["distributed"]["communicator_size"] == 4


def test_a_rank_that_dies_fails_the_run_instead_of_hanging():
    """A rank that raises before the timed region must take the job down with a non-zero exit code within the
    timeout -- not leave the others in a barrier (S4G_BENCH_STUB_FAIL_RANK: test hook of tests/bench_stub only)."""
    proc = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "128"],
                  extra_env={"S4G_BENCH_STUB_FAIL_RANK": "1"}, timeout=300)
    assert proc.returncode != 0
    assert not [l for l in proc.stdout.splitlines() if l.strip().startswith("{")]


def test_without_the_test_backend_bench_needs_a_gpu():
    """No GPU and no test backend: bench.py fails loudly, it never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("S4G_BENCH_BACKEND", "WORLD_SIZE", "RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                          env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode != 0 and "needs a GPU" in proc.stderr and proc.stdout.strip() == ""

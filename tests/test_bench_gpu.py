"""bench.py contract: ONE JSON line with the keys the driver reads (GPU box only)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "scenes/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32 (f16x2 split)"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    ng = r["without_geometry"]       # the same launches with the next batches' geometry answered from a cache
    assert ng["frac"] > 0.9 * r["frac"] and ng["ms_per_step"] > 0 and ng["steps"] == 32
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    # calibrated weights (the default network): outputs that depend on the input, the device within 1e-4 of scale of
    # float64 arithmetic and of the CPU forward by the triangle
    assert c["kind"] == "port" and c["output_spread_min_std_over_max"] > 0.05
    assert c["max_err_over_scale_gpu_vs_float64"] < 1e-4
    assert c["max_err_over_scale_gpu_vs_cpu"] < 1e-4 + c["max_err_over_scale_cpu_vs_float64"]
    assert d["config"]["weights"] == "calibrated" and d["weights_leg"]["weights"] == "randomized"
    det = d["detect"]                      # the reference's production entry composed on the device (detector.GraspDetector)
    assert det["unit"] == "scenes/sec" and det["value"] > 0 and det["candidates_per_scene_mean"] > 0
    assert list(det["stage_ms_b1"]) == ["pre_processing", "prediction", "post_processing", "collision_check",
                                        "importance_sampling"]
    assert 0 < det["latency_ms_b1_graph"] < 50 and 0 < det["latency_ms_b1"] < 50
    p = d["roofline_ball_query_group_points"]
    assert p["bound"] == "hbm" and p["unit"] == "GB/s" and 0 < p["frac"] < 1
    assert ("traffic_source" in r) != ("traffic_note" in r)      # measured on this build, or null + why
    s = d["step_ms"]
    assert s["p10"] <= s["median"] <= s["p90"] and s["n"] == 2
    lat = d["latency"]
    assert lat["latency_ms_one_batch"] > 0 and lat["latency_ms_b1"] > 0 and lat["scenes_per_sec_b1"] > 0
    assert 0 < lat["latency_ms_b1_graph"] < lat["latency_ms_b1"] * 1.2       # the pass as one HIP graph (measured: -5 %)
    assert 0 < lat["latency_ms_one_batch_graph"] < lat["latency_ms_one_batch"] * 1.2
    assert d["io"]["h2d_bytes"] == 16 * 3 * 25600 * 4 and d["io"]["d2h_bytes"] == 16 * 21 * 25600 * 4
    assert "besides the one being collected" in d["config"]["workload"] and d["config"]["in_flight"] == 2
    # BASELINE.json configs[4] rides on the same line: bf16, 32 x 51 200 points, its own roofline
    c4 = d["configs4"]
    assert c4["dtype"] == "bf16" and c4["unit"] == "scenes/sec" and c4["value"] > 0 and c4["steps"] == 30
    assert "51 200" in c4["workload"] and c4["roofline"]["bound"] == "mfma" and 0 < c4["roofline"]["frac"] < 1
    assert c4["roofline"]["mfma_products_per_mac"] == 1
    assert d["collective"]["op"] is None
    # what swapping only the extension buys, the other arithmetic modes, a batch with tie-heavy scenes
    m = d["modules_path"]
    assert m["unit"] == "scenes/sec" and 0 < m["value"] < d["value"]
    legs = d["precision_legs"]
    assert set(legs) == {"fp32", "bf16x3"} and all(0 < v["value"] < d["value"] * 1.05 for v in legs.values())
    assert legs["fp32"]["roofline_peak_TFLOPs"] == 157.3
    x = d["mixed_batch"]
    # the lattice scenes fail the prefix proof: the level-2 sampler (a 1 023-step chain, ~0.9 ms) really ran
    assert x["value"] > 0 and max(x["deeper_fps_launches_ms"].values()) > 0.3
    sh = d["distributed"]
    assert sh["world"] == 1 and sh["communicator_size"] == 1 and sh["per_rank"][0]["scenes"] == [0, 16]


def test_bench_under_torchrun_takes_the_rccl_path():
    """The driver's N > 1 launch line with one rank: RCCL init, the per-batch all-gather,
    barrier and max-reduce all run (S4G_BENCH_FORCE_DIST=1), and the line stays one."""
    env = dict(os.environ, S4G_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
                          "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    c = d["collective"]
    assert c["payload"] == "heads" and c["payload_bytes_per_rank_per_step"] == 16 * 21 * 25600 * 4
    assert c["stream"].startswith("side stream")
    sh = d["distributed"]     # the self-verifying shard table went through the communicator
    assert sh["backend"] == "nccl" and sh["communicator_size"] == 1 and sh["per_rank"][0]["scenes"] == [0, 16]
    # one gathered batch was verified through RCCL before the timed region (own block bit-identical, checksums)
    assert sh["gather_check"]["blocks_verified"] == 1 and sh["gather_check"]["own_block_bit_identical"]
    assert d["configs4"]["value"] > 0 and d["mixed_batch"]["value"] > 0     # rank-0-only legs: no collective inside


def test_bench_gather_poses_over_rccl():
    """`--gather poses`: the K best frames per scene are decoded on the device and ONE all-gather of
    (B, K, 18) fp32 replaces the 21-channel one -- through RCCL with one rank."""
    env = dict(os.environ, S4G_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
                          "29519", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--gather", "poses"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["collective"]
    assert c["payload"] == "poses" and c["payload_bytes_per_rank_per_step"] == 16 * 50 * 18 * 4
    assert "decoded grasp frames" in d["config"]["parallelism"] and d["value"] > 0
    assert "all-gather" in d["config"]["parallelism"]

#!/usr/bin/env python3
"""Headline benchmark: scenes/sec of the S4G forward pass on 25 600-point clouds.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one forward pass over one batch of `--batch` (default 16) synthetic
`tabletop-v1` scenes per GPU, clouds already resident in HBM, four head outputs
left in HBM; with N > 1 every rank runs its own scenes (weak scaling, no
data-path collective) and the per-point outputs are all-gathered over RCCL.
The timed region is bracketed by barrier + synchronize on both sides and the
slowest rank's time is used.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
launches itself: the parent starts `python -m torch.distributed.run ... bench.py`
as a child process BEFORE importing torch or touching a GPU, relays rank 0's
JSON line and returns the child's exit code (reference's only multi-device site:
`grasp_proposal_test.py:52-53`).  A WORLD_SIZE that disagrees with --gpus is an error.

Extra objects on the line:
  roofline      the dominant kernel of a step: the MFMA shared-MLP contraction (all its
                launches), HIP-event durations from the timed region, flops the matrix
                cores execute against the dense peak, PMC traffic from profiles/ (only
                when it was measured on the same sources and arguments, else null)
  roofline_ball_query_group_points
                the HBM-bound operator pair the north star names, at SA1 size, through
                the operator API on the same batch (+ the fused single-pass entry point)
  step_ms       per-step completion intervals: median / p10 / p90 / pipeline fill
  latency       one batch alone (no pipelining) and the one-scene (B = 1) figures
  io            H2D of the clouds / D2H of the outputs, measured apart (never in `value`)
  kernels       every native launch: mean ms, algorithmic bytes / flops, GB/s / TFLOP/s
  configs4      BASELINE.json configs[4] (bf16 path, 32 x 51 200-point clouds): scenes/s, ms/step and
                its own MFMA roofline object, 5 warm + 10 timed steps after the headline region
  modules_path  the reference-shaped modules over the HIP operators on the same scenes (what swapping
                only the extension buys), scenes/s
  precision_legs  the fast path in true-fp32 MFMA and exact 3 x bf16 arithmetic on the same batch
  detect        the reference's production entry GraspDetector.detect composed on the device (detector.py): raw clouds in,
                selected grasp poses out; scenes/s pipelined at the step's batch, per-stage ms, one-scene latency (eager / graph)
  weights_leg   the same step on the other network (--weights: `calibrated` is the default and the headline; `randomized`
                is what rounds 1-5 timed)
  mixed_batch   2 of the 16 scenes tie-heavy: the conditional level-2 / level-3 FPS samplers run timed
  kept_points   clouds in, 50 grasp frames per scene out with the pose heads evaluated on the 2 048 best-scoring points
                per scene only (FusedPointNet2(..., topk=)), next to the full forward + decode; NOT the headline
  collective    the per-batch all-gather: payload (--gather heads | poses), bytes, the stream it ran on
  distributed   world, communicator size, every rank's scene range and device (self-verifying)
  cpu_baseline  the CPU oracle forward (oracle/pn2_forward.py) on ONE scene,
                rank 0 at N == 1 only -- a reported baseline, not the target

TEST MODE (tests/test_bench_world.py only): S4G_BENCH_BACKEND=gloo runs this file's whole control flow --
self-launch, process group, shard table, gather check, fenced timed region, max over ranks, ranks != 0
leaving, rank 0's line -- on CPU with `tests/bench_stub.StubRunner` in place of the network (it computes
nothing of the network: a fill pattern with the fast path's submit() / result() shape).  The line then
carries `"test_mode"` and its `value` is not a measurement.  Without that variable a GPU is required.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3      # fp32-in MFMA = fp32 vector peak
BF16_MFMA_PEAK_TF = 2500.0     # dense bf16 / fp16 MFMA (no sparsity)
# calibrated on this board (profiles/r05_mfma_ceiling.md, tools/micro/mfma_ceiling.hip): the product kernels' inner loop
# (A fragments from LDS, W fragments through a register ring from L2, 2 row blocks per wave) on RANDOM fp16 operands
# sustains 1 248 TFLOP/s at the clock the power governor allows (1.81 GHz, 1.29 kW); registers-only: 1 681; constant
# operands: 2 436
F16_MFMA_ATTAINABLE_TF = 1248.0
# ... and the single-plane bf16 chains' loop (8 MFMAs per step and wave on four row blocks, configs[4]): 1 357 TFLOP/s
BF16_MFMA_ATTAINABLE_TF = 1357.0
GFLOP_PER_SCENE = 203.48       # SURVEY.md Appendix B (BN folded, 2*MAC), N = 25 600
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~0.5 s (long enough for an external GPU-busy sampler to see it,
    # and for the one pipeline fill inside it to stop mattering)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="scenes per GPU per step")
    ap.add_argument("--points", type=int, default=25600)
    ap.add_argument("--impl", default="auto", choices=["auto", "fused", "modules"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the latency / io / operator-pair probes after the timed region")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches kept in flight besides the one being collected")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="collect each batch before submitting the next")
    ap.add_argument("--variant", default="tabletop-v1")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total scenes per step over all GPUs instead of --batch per GPU; must divide "
                         "evenly over the ranks (exit code 2 otherwise)")
    ap.add_argument("--gather", default="heads", choices=["heads", "poses"],
                    help="payload of the per-batch all-gather (N > 1): the 21 per-point head channels, or "
                         "the --num-poses best decoded grasp frames per scene (3.6 KB instead of 2.15 MB)")
    ap.add_argument("--num-poses", type=int, default=50)
    ap.add_argument("--timer-every", type=int, default=int(os.environ.get("S4G_BENCH_TIMER_EVERY", "4")),
                    help="HIP event pairs around the native launches of every k-th forward pass of the timed "
                         "region (default 4: the pairs are barrier packets in the queues they time -- with a "
                         "pair around EVERY launch the step measured 9.04 ms against 8.85 ms at k = 4 and "
                         "8.84 ms with none, profiles/r03_timer_sampling.md)")
    ap.add_argument("--no-configs4", action="store_true",
                    help="skip the configs[4] (bf16, 51 200 points, 32 scenes) leg after the timed region")
    ap.add_argument("--weights", default="calibrated", choices=["calibrated", "randomized"],
                    help="the network the step is timed on: 'calibrated' (default) = the golden run's network "
                         "(tests/golden/pn2_calib_full.npz: seeded convolutions, BatchNorm statistics calibrated through "
                         "the reference's own modules -- activations that carry signal at every point, as a trained "
                         "network's do); 'randomized' = rounds 1-5's network (randomize_bn_: every activation a "
                         "per-channel constant after the first layers)")
    ap.add_argument("--precision", default=None, choices=["f16x2", "bf16x3", "fp32", "bf16"],
                    help="contraction arithmetic of the fast path (default f16x2 = fp32-class); "
                         "'bf16' is the reduced-precision roofline configuration (configs[4]), "
                         "not the headline")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_command(args, argv):
    """The child command of a self-launched multi-GPU run (one process per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
            str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
            os.path.join(ROOT, "bench.py")] + list(argv)


def maybe_spawn(args, argv, environ=None, run=subprocess.run, out=None, err=None):
    """`python bench.py --gpus N` (N > 1) without torchrun: start the N ranks as a CHILD process
    (never an exec; nothing in this process has touched the GPU yet), relay rank 0's JSON line.
    Returns the exit code to leave with, or None when this process is itself a rank."""
    environ = os.environ if environ is None else environ
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != args.gpus:
            print("bench.py: WORLD_SIZE=%d disagrees with --gpus %d" % (world, args.gpus), file=err)
            return 2
        return None
    if args.gpus <= 1:
        return None
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = run(spawn_command(args, argv), env=env, cwd=ROOT, stdout=subprocess.PIPE,
               stderr=subprocess.PIPE, text=True)
    lines = [l for l in (proc.stdout or "").splitlines() if l.strip().startswith("{")]
    rest = [l for l in (proc.stdout or "").splitlines() if not l.strip().startswith("{")]
    if rest:
        print("\n".join(rest), file=err)
    if proc.stderr:
        print(proc.stderr[-8000:], file=err)
    if proc.returncode == 0 and len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=err)
        return 1
    for l in lines[-1:]:
        print(l, file=out)
    return proc.returncode


def source_stamp():
    """Hash of the sources the measured kernels are built from; PMC traffic figures in
    profiles/ are attached only when they were collected on the same sources."""
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "s4g_release_amd")
    files = sorted(os.path.join(pkg, "csrc", f) for f in os.listdir(os.path.join(pkg, "csrc"))
                   if f.endswith((".hip", ".h")))
    files += [os.path.join(pkg, "fused.py"), os.path.join(pkg, "functions.py")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(key):
    """(bytes, source, None) of `key` in profiles/r04_traffic.json when its stamp matches this
    tree, else (None, None, reason)."""
    try:
        with open(TRAFFIC_FILE) as f:
            tr = json.load(f).get(key)
    except (OSError, ValueError):
        return None, None, "no PMC summary for this build in profiles/"
    if not tr:
        return None, None, "no PMC pass was collected for %s" % key
    if tr.get("source_stamp") != source_stamp():
        return None, None, ("PMC pass in %s was collected on other sources (stamp %s)"
                            % (os.path.basename(TRAFFIC_FILE), tr.get("source_stamp")))
    return tr["traffic_bytes"], tr["source"], None


class _HipRank:
    """Device plumbing of one rank: a HIP device, RCCL, HIP events."""
    backend, test_mode = "nccl", None

    def __init__(self, torch, local_rank):
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        self.torch = torch
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.name = torch.cuda.get_device_name(self.dev) + " cuda:%d" % local_rank

    def sync(self):
        self.torch.cuda.synchronize()

    def event(self):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def init_kw(self):
        return {"backend": "nccl", "device_id": self.dev}


class _HostEvent:
    def __init__(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


class _HostRank:
    """S4G_BENCH_BACKEND=gloo (tests only): the same control flow on CPU tensors, host clock for events."""
    backend = "gloo"
    test_mode = ("S4G_BENCH_BACKEND=gloo: tests/bench_stub.StubRunner on CPU instead of the network -- "
                 "control-flow test of bench.py, `value` is NOT a measurement")

    def __init__(self, torch, local_rank):
        self.dev = torch.device("cpu")
        self.name = "host-stub:%d" % local_rank

    def sync(self):
        pass

    def event(self):
        return _HostEvent()

    def init_kw(self):
        return {"backend": "gloo"}


def percentile(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    pos = (len(xs) - 1) * q
    lo = int(pos)
    hi = min(lo + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (pos - lo)


def dense_roofline(summary, steps_timed, precision, world_note=""):
    """(`roofline` object of the MFMA contraction launches, `kernels` table) from an OpTimer summary
    whose launches were timed in `steps_timed` forward passes."""
    kernels = {}
    gemm_ms = gemm_flops = 0.0
    gemm_launches = 0
    for name, (n, ms, nbytes, flops) in sorted(summary.items()):
        if flops > 0:
            kernels[name] = {"launches": n, "ms": round(ms, 5), "GFLOP": round(flops / 1e9, 3),
                             "TFLOPs": round(flops / ms / 1e9, 2) if ms > 0 else None}
            gemm_ms += ms * n / steps_timed
            gemm_flops += flops * n / steps_timed
            gemm_launches += n
        else:
            kernels[name] = {"launches": n, "ms": round(ms, 5), "bytes": int(nbytes),
                             "GBps": round(nbytes / ms / 1e6, 2) if ms > 0 else None}
    if gemm_ms <= 0:
        return None, kernels
    dense_tf = gemm_flops / gemm_ms / 1e9
    products = {"bf16": 1, "f16x2": 3, "bf16x3": 6}.get(precision)
    if products is None:
        return {"kernel": "mlp_gemm_kernel (v_mfma_f32_32x32x2_f32), all launches of one step",
                "bound": "mfma", "achieved": round(dense_tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(dense_tf / FP32_MFMA_PEAK_TF, 4), "GFLOP_per_step": round(gemm_flops / 1e9, 1),
                "ms_per_step": round(gemm_ms, 3)}, kernels
    attainable = BF16_MFMA_ATTAINABLE_TF if precision == "bf16" else F16_MFMA_ATTAINABLE_TF
    kname = {"bf16": "mlp_chain_kernel<PL=1> (fused layer chains, one bf16 plane) + mlp_heads_kernel<PL=1> + "
                     "mlp_gemm_f16x2_kernel<PL=1> (v_mfma_f32_32x32x16_bf16, ONE product per MAC: REDUCED "
                     "PRECISION, the configs[4] roofline configuration)",
             "f16x2": "mlp_chain_kernel<PL=2> (fused layer chains) + mlp_heads_kernel<PL=2> + mlp_gemm_f16x2_kernel "
                      "(v_mfma_f32_32x32x16_f16, 3 products per fp32-equivalent product)",
             "bf16x3": "mlp_gemm_bf16x3_kernel (v_mfma_f32_32x32x16_bf16, 6 products per "
                       "fp32-equivalent product)"}[precision]
    return {"kernel": kname + ", all %d contraction launches of one step; flops = those executed after "
                              "moving the linear first SA / FP layers in front of the grouping / "
                              "interpolation" % (gemm_launches // max(steps_timed, 1)),
            "bound": "mfma", "achieved": round(products * dense_tf, 1), "peak": BF16_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": round(products * dense_tf / BF16_MFMA_PEAK_TF, 4),
            "mfma_products_per_mac": products, "fp32_equivalent_TFLOPs": round(dense_tf, 2),
            "fp32_equivalent_vs_fp32_mfma_peak": round(dense_tf / FP32_MFMA_PEAK_TF, 4),
            "GFLOP_per_step_fp32_equivalent": round(gemm_flops / 1e9, 1), "ms_per_step": round(gemm_ms, 3),
            "timed_passes": steps_timed,
            "attainable": {"TFLOPs": attainable, "of_peak": round(attainable / BF16_MFMA_PEAK_TF, 4),
                           "source": "profiles/r05_mfma_ceiling.md: this precision's inner loop (A fragments from LDS, W "
                                     "fragments from L2) on random operands under hwmon sampling, 1.29-1.34 kW / 1.8-1.95 GHz; "
                                     "f16x2 registers only 1 681 TF, constant operands 2 436"},
            "frac_of_attainable": round(products * dense_tf / attainable, 4),
            "peak_note": "peak is the nominal 2.4 GHz figure; measured (profiles/r03_power_clock.md: hwmon power / "
                         "clock sensors while each kernel runs back to back) these kernels draw 1.24-1.40 kW of the "
                         "1.40 kW board cap and are clocked at 1.8-2.2 GHz"}, kernels


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    rc = maybe_spawn(args, argv)          # before torch / HIP are imported
    if rc is not None:
        sys.exit(rc)

    import numpy as np
    import torch
    import torch.distributed as dist
    from s4g_release_amd import _cabi, dist as sdist, functions as F, synth
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.global_batch is not None:
        try:
            args.batch = sdist.scenes_per_rank(args.global_batch, world)
        except ValueError as e:
            print("bench.py: --global-batch: %s" % e, file=sys.stderr)
            sys.exit(2)
    stub = os.environ.get("S4G_BENCH_BACKEND") == "gloo"      # tests only: see the docstring
    hw = (_HostRank if stub else _HipRank)(torch, local_rank)
    dev = hw.dev
    if not stub:
        _cabi.lib()   # no HIP library -> fail loudly, never a fallback
    # S4G_BENCH_FORCE_DIST=1: take the RCCL path (init, all-gather, barrier, max-reduce)
    # even with one rank, so the multi-GPU code can be exercised on a 1-GPU box
    use_dist = world > 1 or (os.environ.get("S4G_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's all-gather runs as channel workgroups beside contraction kernels that fill every CU with one or two
        # workgroups: each channel can hold a CU's slot for the collective's duration.  8 channels move the 238 MB a rank
        # receives per step in well under a step and cap that cost at 8 of 256 CUs; a value set by
        # the caller wins.
        sdist.bound_rccl_channels()
        dist.init_process_group(**hw.init_kw())

    cfg = S4GConfig()
    impl = args.impl
    fused_cls = None
    if stub:
        from tests.bench_stub import StubRunner
        net, runner, impl = None, StubRunner(), "stub"
        args.no_extras = args.no_cpu_baseline = True
    else:
        def make_net(kind):
            if kind == "calibrated":
                from tests import golden_util as GU      # data fixture: BatchNorm tensors by value, sha-checked
                return GU.calib_full_model().to(dev).eval()
            torch.manual_seed(20260101)
            n_ = build_pointnet2_cls(cfg)
            randomize_bn_(n_, 20260102)
            return n_.to(dev).eval()
        net = make_net(args.weights)
        if impl in ("auto", "fused"):
            try:
                from s4g_release_amd.fused import FusedPointNet2 as fused_cls
            except ImportError:
                if impl == "fused":
                    raise
        if fused_cls is not None:
            runner = fused_cls(net, precision=args.precision)
            impl = "fused"
        else:
            runner = net
            impl = "modules"
    precision = getattr(runner, "precision", "library")
    if impl not in ("fused", "stub"):
        args.timer_every = 1      # only the fast path announces its passes to the timers

    B = args.batch
    scene_ids = [rank * B + i for i in range(B)]
    pts_host = torch.from_numpy(synth.make_batch(scene_ids, args.points, variant=args.variant))
    pts = pts_host.to(dev)
    batch = {"scene_points": pts}
    heads = ("score", "frame_R", "frame_t", "movable_logits")

    pipelined = impl in ("fused", "stub") and not args.no_pipeline

    # the per-batch collective: packed head outputs (21 channels per point) or, with --gather poses,
    # the K best decoded grasp frames per scene, on its own stream
    gather = None
    if use_dist:
        decode = None
        if args.gather == "poses":
            from s4g_release_amd import postprocess as PP
            decode = lambda pred, xyz: PP.decode_top_poses(pred, xyz, args.num_poses, "detector")
        gather = sdist.OutputGather(args.gather, decode=decode, device=dev)

    solo = [False]    # set once the other ranks have left: nothing after that may enter a collective

    def run_steps(n, run=None, data=None, pipe=None, marks=None, gathered=True):
        """n forward passes over the batch.  Pipelined mode keeps up to `--in-flight`
        batches submitted besides the one being collected: batch i+1's FPS chain
        starts on the geometry stream before batch i's outputs are collected; every
        batch is complete when the trailing fence returns.  `marks` collects one HIP
        event per completed step (recorded on the collecting stream).  gathered=False: no
        collective (the probes after the timed region run on rank 0 alone)."""
        run = runner if run is None else run
        data = batch if data is None else data
        pipe = pipelined if pipe is None else pipe
        assert not (solo[0] and gathered and gather is not None), "all-gather inside a rank-0-only leg"
        fin = (lambda pred: gather(pred, data["scene_points"])) if (gathered and gather is not None) \
            else (lambda pred: pred)

        def mark():
            if marks is not None:
                marks.append(hw.event())
        if n <= 0:
            return
        with torch.no_grad():
            if not pipe:
                for _ in range(n):
                    fin(run(data))
                    mark()
                return
            pending = []
            for _ in range(n):
                pending.append(run.submit(data))
                if len(pending) > args.in_flight:
                    fin(pending.pop(0).result())
                    mark()
            while pending:
                fin(pending.pop(0).result())
                mark()

    def _fence(collective=True):
        """collective=False: the rank-0-only probes after the headline region -- the other ranks have left,
        a barrier there would wait for peers that never come."""
        hw.sync()
        if use_dist and collective:
            assert not solo[0], "collective fence inside a rank-0-only leg"
            dist.barrier()
            hw.sync()

    def timed_region(steps, warmup, collective=True, timers=True, **kw):
        """`warmup` untimed + EXACTLY `steps` timed passes between fences -> (seconds, step_ms, summary,
        passes whose launches carried event pairs).  collective=False (rank-0-only legs): the fences only
        synchronise the device and nothing in the region may enter a collective."""
        assert collective or kw.get("gathered") is False
        fence = lambda: _fence(collective)
        run_steps(warmup, **kw)
        F.OpTimer.reset(enabled=timers, every=args.timer_every)
        fence()
        marks = []
        ev0 = hw.event()
        t0 = time.perf_counter()
        run_steps(steps, marks=marks, **kw)
        fence()
        elapsed = time.perf_counter() - t0
        F.OpTimer.enabled = False
        stamps = [ev0.elapsed_time(m) for m in marks]
        deltas = [b - a for a, b in zip([0.0] + stamps[:-1], stamps)]
        steady = deltas[1:] if len(deltas) > 1 else deltas
        step_ms = {"median": round(percentile(steady, 0.5), 3), "p10": round(percentile(steady, 0.1), 3),
                   "p90": round(percentile(steady, 0.9), 3), "first_step_incl_pipeline_fill": round(deltas[0], 3),
                   "pipeline_fill": round(max(0.0, deltas[0] - percentile(steady, 0.5)), 3) if len(deltas) > 1 else None,
                   "n": len(steady), "source": "HIP events after each collected batch"}
        timed_passes = (steps + args.timer_every - 1) // args.timer_every
        return elapsed, step_ms, F.OpTimer.summary(), timed_passes

    # who runs what (one all_gather_object, outside the timed region): world, communicator size as the
    # collective library reports it, every rank's scene range and device -- checked to tile the global batch
    shards = sdist.shard_report(scene_ids, world * B, hw.name)
    if gather is not None:
        # one gathered batch checked on EVERY rank before anything is timed: block r of the gathered tensor must
        # carry rank r's checksum (exchanged through all_gather_object) and this rank's block must equal what it
        # computed, bit for bit -- the first real N > 1 run verifies its data path, not only its shard table
        with torch.no_grad():
            shards["gather_check"] = sdist.gather_check(gather, runner(batch), pts)
        hw.sync()

    elapsed, step_ms, summary, timed_passes = timed_region(args.steps, args.warmup)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    solo[0] = True

    scenes = world * B * args.steps
    value = scenes / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    roofline_dense, kernels = dense_roofline(summary, timed_passes, precision)

    N, M, K = args.points, cfg.num_centroids[0], cfg.num_neighbours[0]
    roofline = latency = io = configs4 = modules_path = precision_legs = mixed_batch = kept_points = weights_leg = detect_leg = None
    # (the single-GPU probes below run at N = 1 only: at N > 1 the other ranks have left, rank 0 prints its line
    #  and tears the communicator down without making the job wait for figures the N = 1 line already carries)
    if not args.no_extras and world == 1:
        # ---- one batch alone / one scene: the figures the pipeline hides
        def timed_forward(data, reps):
            ts = []
            with torch.no_grad():
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    runner(data)
                    torch.cuda.synchronize()
                    ts.append(1e3 * (time.perf_counter() - t1))
            return percentile(ts, 0.5)
        def timed_forward_with(fn, data, reps):
            ts = []
            for _ in range(2 + reps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn(data)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t1))
            return percentile(ts[2:], 0.5)
        one = {"scene_points": pts[:1].contiguous()}
        latency = {"latency_ms_one_batch": round(timed_forward(batch, 5), 3), "batch": B,
                   "latency_ms_b1": round(timed_forward(one, 5), 3)}
        if pipelined:
            run_steps(3, data=one, gathered=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(30, data=one, gathered=False)
            torch.cuda.synchronize()
            latency["scenes_per_sec_b1"] = round(30 / (time.perf_counter() - t1), 2)
        latency["scenes_per_sec_one_batch_at_a_time"] = round(1e3 * B / latency["latency_ms_one_batch"], 2)
        if impl == "fused":
            # the same two figures with the pass recorded as ONE HIP graph (FusedPointNet2.graph: the ~45 launches of
            # a pass replayed by one host call; outputs bit-identical, tests/test_fused_gpu.py)
            g1, g16 = runner.graph(one), runner.graph(batch)
            latency["latency_ms_b1_graph"] = round(timed_forward_with(g1, one, 10), 3)
            latency["latency_ms_one_batch_graph"] = round(timed_forward_with(g16, batch, 5), 3)
            del g1, g16

        # ---- host <-> device transfers of one step, apart from `value`
        with torch.no_grad():
            pred = runner(batch)
        pin = pts_host.pin_memory()
        d2h_src = [pred.packed] if getattr(pred, "packed", None) is not None else [pred[k] for k in heads]
        outs_host = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in d2h_src]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        h2d = d2h = 0.0
        for rep in range(4):
            torch.cuda.synchronize()
            e[0].record()
            pts.copy_(pin, non_blocking=True)
            e[1].record()
            for t, o in zip(d2h_src, outs_host):
                o.copy_(t, non_blocking=True)
            e[2].record()
            torch.cuda.synchronize()
            if rep:
                h2d += e[0].elapsed_time(e[1]) / 3
                d2h += e[1].elapsed_time(e[2]) / 3
        io = {"h2d_ms_per_step": round(h2d, 4), "h2d_bytes": int(pts.numel() * 4),
              "d2h_ms_per_step": round(d2h, 4), "d2h_bytes": int(sum(o.numel() for o in outs_host) * 4),
              "note": "pinned host buffers, one step's clouds in / the packed (B, 21, N) head outputs out; not part of `value`"}

        # ---- the contraction launches WITHOUT the next batches' geometry beside them: the same pipelined loop with
        # the coordinate-only operators answered from a cache (tools/geo_cost.py's trick; the tensors are the same, the
        # operators do not run) -- what the MFMA kernels reach when no FPS workgroup holds 16 CUs' register files
        if impl == "fused" and roofline_dense is not None:
            geo_ops = ("_fps_gather", "_fps_prefix_check", "_ball_query", "_group_rel_xyz_unique", "_group_rel_xyz",
                       "_three_nn")
            geo_orig = {n: getattr(runner, n) for n in geo_ops}
            geo_cache = {}

            def _cached(name):
                def f(*a, **kw):
                    key = (name,) + tuple(tuple(t.shape) if isinstance(t, torch.Tensor) else t for t in a) + \
                        tuple(sorted((k, v is not None) for k, v in kw.items()))
                    if key not in geo_cache:
                        geo_cache[key] = geo_orig[name](*a, **kw)
                    return geo_cache[key]
                return f
            try:
                for n in geo_ops:
                    setattr(runner, n, _cached(n))
                ng_el, _, ng_sum, ng_tp = timed_region(32, 6, collective=False, gathered=False)
            finally:
                for n in geo_ops:
                    setattr(runner, n, geo_orig[n])
            ng_roof, _ = dense_roofline(ng_sum, ng_tp, precision)
            roofline_dense["without_geometry"] = {
                "frac": ng_roof["frac"], "achieved": ng_roof["achieved"], "ms_per_step": ng_roof["ms_per_step"],
                "step_ms": round(1e3 * ng_el / 32, 3), "steps": 32,
                "note": "same loop, the next batches' FPS / ball queries / 3-NN answered from a cache: the contraction "
                        "launches with the whole chip to themselves (tools/geo_cost.py, profiles/r04_geometry_cost.md)"}

        # ---- north-star roofline: the operator pair ball_query + group_points(xyz) at SA1
        # size on this step's batch, through the public operator API (int64 indices),
        # HIP events around each launch on the launch stream.
        with torch.no_grad():
            ctr = F.gather_points(pts, F.farthest_point_sample(pts, M))
            for rep in range(2 + 10):
                if rep == 2:
                    torch.cuda.synchronize()
                    F.OpTimer.reset(enabled=True)
                gidx, _ = F.ball_query(pts, ctr, cfg.radius[0], K)
                F.group_points(pts, gidx)
                F.query_and_group(pts, ctr, cfg.radius[0], K)
            torch.cuda.synchronize()
            F.OpTimer.enabled = False
        probe = F.OpTimer.summary()
        bq = probe["ball_query[N=%d,M=%d,K=%d]" % (N, M, K)]
        gp = probe["group_points[C=3,N=%d,M=%d,K=%d]" % (N, M, K)]
        nbytes = bq[2] + gp[2]
        ms = bq[1] + gp[1]
        roofline = {"kernel": "ball_query + group_points(xyz) at SA1 size (N=%d, M=%d, K=%d), "
                              "operator API" % (N, M, K), "bound": "hbm",
                    "achieved": round(nbytes / ms / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": None,
                    "bytes_per_launch_pair": int(nbytes), "ms_ball_query": round(bq[1], 5),
                    "ms_group_points": round(gp[1], 5), "scenes_per_launch": B,
                    "note": "algorithmic bytes = B*(12N+12M+8MK+8M) + B*(4CN+8MK+4CMK), C=3"}
        tb, src, why = load_traffic("ball_query+group_points[N=%d,M=%d,K=%d,B=%d]" % (N, M, K, B))
        roofline["traffic"] = tb
        roofline["traffic_source" if tb is not None else "traffic_note"] = src if tb is not None else why
        fq = probe.get("query_group[N=%d,M=%d,K=%d]" % (N, M, K))
        if fq:   # the same pair as ONE pass (s4g_query_group_f32), same algorithmic bytes
            roofline["fused_pair_ms"] = round(fq[1], 5)
            roofline["fused_pair_achieved"] = round(fq[2] / fq[1] / 1e6, 2)
            roofline["fused_pair_frac"] = round(fq[2] / fq[1] / 1e6 / HBM_PEAK_GBS, 4)

        # ---- BASELINE.json configs[4] on the same line: bf16 grouped-MLP path, 51 200-point clouds,
        # 32 scenes per step, achieved vs the bf16 MFMA peak (rank 0 alone, no collective)
        if impl == "fused" and not args.no_configs4 and not (args.points == 51200 and precision == "bf16"):
            c4_runner = fused_cls(net, precision="bf16")
            c4_pts = torch.from_numpy(synth.make_batch(list(range(32)), 51200, variant=args.variant)).to(dev)
            c4_batch = {"scene_points": c4_pts}
            c4_steps, c4_warm = 30, 5      # (30: the pipeline's fill and drain are one step of the region)
            c4_el, c4_step, c4_sum, c4_tp = timed_region(c4_steps, c4_warm, collective=False, run=c4_runner,
                                                         data=c4_batch, gathered=False)
            c4_roof, _ = dense_roofline(c4_sum, c4_tp, "bf16")
            tb, src, why = load_traffic("contractions[step,B=32,N=51200,precision=bf16]")
            c4_roof["traffic"] = tb
            c4_roof["traffic_source" if tb is not None else "traffic_note"] = src if tb is not None else why
            configs4 = {"workload": "BASELINE.json configs[4]: bf16 grouped-MLP MFMA path, 32 scenes x 51 200-pt "
                                    "tabletop-v1 clouds per step, single-plane bf16 contraction (fp32 accumulate), "
                                    "indices bit-exact, pipelined like the headline run",
                        "value": round(32 * c4_steps / c4_el, 2), "unit": "scenes/sec", "dtype": "bf16",
                        "ms_per_step": round(1e3 * c4_el / c4_steps, 3), "steps": c4_steps, "warmup": c4_warm,
                        "step_ms_median": c4_step["median"], "roofline": c4_roof}
            del c4_runner, c4_pts, c4_batch

        # ---- what a maintainer who swaps ONLY the extension gets (INTEGRATION.md levels 1-2): the
        # reference-shaped modules (modules.py / nn_utils.py: torch conv + BN + ReLU over materialised
        # (B, C, M, K) tensors) on the HIP operators, same 16 scenes, one batch at a time
        if impl == "fused":
            m_steps, m_warm = 5, 2
            m_el, m_step, _, _ = timed_region(m_steps, m_warm, collective=False, timers=False, run=net,
                                              data=batch, pipe=False, gathered=False)
            modules_path = {"workload": "reference-shaped modules (QueryGrouper / PointNetSAModule / "
                                        "PointnetFPModule + torch conv-BN-ReLU) over the HIP operator API, "
                                        "same %d scenes, fp32 library convolutions, one batch at a time" % B,
                            "value": round(B * m_steps / m_el, 2), "unit": "scenes/sec",
                            "ms_per_step": round(1e3 * m_el / m_steps, 3), "steps": m_steps, "warmup": m_warm}
            # ---- the other arithmetic modes of the fast path on the same batch: true fp32 MFMA
            # (v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain) and the exact 3 x bf16 split
            precision_legs = {}
            for prec in ("fp32", "bf16x3"):
                if prec == precision:
                    continue
                p_runner = fused_cls(net, precision=prec)
                p_steps, p_warm = 6, 3
                p_el, p_step, p_sum, p_tp = timed_region(p_steps, p_warm, collective=False, run=p_runner,
                                                         data=batch, gathered=False)
                p_roof, _ = dense_roofline(p_sum, p_tp, prec)
                precision_legs[prec] = {"value": round(B * p_steps / p_el, 2), "unit": "scenes/sec",
                                        "ms_per_step": round(1e3 * p_el / p_steps, 3), "steps": p_steps,
                                        "warmup": p_warm, "roofline_frac": p_roof["frac"] if p_roof else None,
                                        "roofline_peak_TFLOPs": p_roof["peak"] if p_roof else None}
                del p_runner
            # ---- the same step on the OTHER network: rounds 1-5 timed `randomized` (randomize_bn_: sigma^2 in
            # [0.5, 1.5] against real variances of 1e-4 .. 1e-2 -- every activation panel a per-channel constant after the
            # first layers); the headline is now timed on `calibrated` weights, whose activations toggle the MFMA operand
            # lanes the way a trained network's do (MFMA throughput under the power cap depends on operand toggle rate:
            # profiles/r05_mfma_ceiling.md).  Same 16 scenes, same pipelining, 20 steps.
            other = "randomized" if args.weights == "calibrated" else "calibrated"
            o_runner = fused_cls(make_net(other), precision=precision)
            o_steps, o_warm = 20, 5
            o_el, o_step, o_sum, o_tp = timed_region(o_steps, o_warm, collective=False, run=o_runner, data=batch,
                                                     gathered=False)
            o_roof, _ = dense_roofline(o_sum, o_tp, precision)
            weights_leg = {"weights": other, "value": round(B * o_steps / o_el, 2), "unit": "scenes/sec",
                           "ms_per_step": round(1e3 * o_el / o_steps, 3), "step_ms_median": o_step["median"],
                           "steps": o_steps, "warmup": o_warm,
                           "contraction_ms_per_step": o_roof["ms_per_step"] if o_roof else None,
                           "roofline_frac": o_roof["frac"] if o_roof else None,
                           "headline_over_this": round(value / (B * o_steps / o_el), 4)}
            del o_runner
            # ---- clouds in, grasp frames out, with the heads evaluated where the decode reads them: the
            # score head on every point, the rotation / translation / movable heads on the 2 048 best-scoring points per
            # scene (FusedPointNet2(..., topk=): the same top-50 frames as the full forward + decode,
            # tests/test_sparse_heads_gpu.py), then the device-side top-50 decode -- pipelined like the headline run.
            # NOT the headline: the headline leaves all 21 channels of every point in HBM.
            from s4g_release_amd import postprocess as PP

            class _KeptPoints:
                def __init__(self, r, k, poses):
                    self.r, self.k, self.poses = r, k, poses

                class _H:
                    def __init__(self, h, data, poses):
                        self.h, self.data, self.poses = h, data, poses

                    def result(self):
                        return PP.decode_top_poses(self.h.result(), self.data["scene_points"], self.poses)

                def submit(self, data):
                    return self._H(self.r.submit(data, topk=self.k), data, self.poses)

                def __call__(self, data):
                    return self.submit(data).result()
            k_steps, k_warm, k_keep = 20, 5, 2048
            kp_el, kp_step, kp_sum, kp_tp = timed_region(k_steps, k_warm, collective=False, run=_KeptPoints(runner, k_keep, 50),
                                                         data=batch, gathered=False)
            kp_roof, _ = dense_roofline(kp_sum, kp_tp, precision)
            with torch.no_grad():
                f_el, _, _, _ = timed_region(k_steps, k_warm, collective=False, timers=False,
                                             run=type("_FullDecode", (), {
                                                 "submit": lambda self, d: _KeptPoints._H(runner.submit(d), d, 50),
                                                 "__call__": lambda self, d: PP.decode_top_poses(runner(d), d["scene_points"], 50)})(),
                                             data=batch, gathered=False)
            kept_points = {"workload": "clouds in, the 50 best grasp frames per scene out: score head on every point, pose heads "
                                       "on the %d best-scoring points per scene, device-side decode; same %d scenes, pipelined"
                                       % (k_keep, B),
                           "value": round(B * k_steps / kp_el, 2), "unit": "scenes/sec",
                           "ms_per_step": round(1e3 * kp_el / k_steps, 3), "step_ms_median": kp_step["median"],
                           "steps": k_steps, "warmup": k_warm, "kept_points_per_scene": k_keep, "poses_per_scene": 50,
                           "contraction_ms_per_step": kp_roof["ms_per_step"] if kp_roof else None,
                           "full_forward_plus_decode": {"value": round(B * k_steps / f_el, 2),
                                                        "ms_per_step": round(1e3 * f_el / k_steps, 3)},
                           "note": "not the headline metric: the headline forward leaves all 21 channels of every point"}
            # ---- the reference's production entry, `GraspDetector.detect` (grasp_detector.py:187-254), composed on the
            # device (s4g_release_amd/detector.py): raw 48 902-point clouds in (the size of the reference's sample scene),
            # selected grasp poses out -- subsample + REAL2TRAIN, forward with the pose heads on the 2 048 best-scoring
            # points, thresholds / decode / Gram-Schmidt, ONE batched collision launch against the whole raw cloud (the
            # reference: a Python loop with a launch sequence and a host sync per pose, :216-234), survivor compaction,
            # importance sampling; no host synchronisation inside.  NOT the headline.
            from s4g_release_amd.detector import GraspDetector
            n_raw = 48902
            t_raw = synth.make_batch(scene_ids, n_raw, variant=args.variant)                 # TRAIN frame
            raw = torch.from_numpy(np.ascontiguousarray(np.stack([t_raw[:, 1], t_raw[:, 0], -t_raw[:, 2]], axis=1))).to(dev)
            det = GraspDetector(runner, topk=2048)
            with torch.no_grad():
                probe = PP.expected_score(runner({"scene_points": det.pre_processing(raw)})["score"].contiguous(), "detector")
                d_thr = float(torch.quantile(probe.flatten().float()[:: 7], 0.98))      # ~500 candidates per scene
            d_kw = dict(num_selected=5, score_threshold=d_thr, verticalness_threshold=-2.0, collision_check=True)

            class _Detect:
                def submit(self, data):
                    return det.submit(data["cloud"], **d_kw)

                def __call__(self, data):
                    return self.submit(data).result()
            d_steps, d_warm = 20, 5
            with torch.no_grad():
                d_el, d_step, _, _ = timed_region(d_steps, d_warm, collective=False, timers=False, run=_Detect(),
                                                  data={"cloud": raw, "scene_points": raw}, gathered=False)
                det.stage_events = []
                out16 = det.detect_device(raw, **d_kw)
                stage16 = det.stage_ms()
                det.stage_events = []
                det.detect_device(raw[:1].contiguous(), **d_kw)
                stage1 = det.stage_ms()
                det.stage_events = None
                one_raw = raw[:1].contiguous()
                lat1 = timed_forward_with(lambda c: det.detect_device(c, **d_kw), one_raw, 10)
                g_det = det.graph(one_raw, **d_kw)
                lat1_graph = timed_forward_with(g_det, one_raw, 20)
                del g_det
            detect_leg = {"workload": "GraspDetector.detect on the device: %d raw %d-point %s clouds per step -> 5 selected grasp "
                                      "poses per scene (pre-processing as the reference executes it, forward with topk=2048, "
                                      "decode, batched collision check against the raw cloud, importance sampling), pipelined"
                                      % (B, n_raw, args.variant),
                          "value": round(B * d_steps / d_el, 2), "unit": "scenes/sec",
                          "ms_per_step": round(1e3 * d_el / d_steps, 3), "step_ms_median": d_step["median"],
                          "steps": d_steps, "warmup": d_warm, "score_threshold": round(d_thr, 4),
                          "candidates_per_scene_mean": round(float(out16.candidates[3].float().mean()), 1),
                          "stage_ms_one_batch_of_%d" % B: {k: round(v, 3) for k, v in stage16.items()},
                          "stage_ms_b1": {k: round(v, 3) for k, v in stage1.items()},
                          "latency_ms_b1": round(lat1, 3), "latency_ms_b1_graph": round(lat1_graph, 3),
                          "reference_stages": "grasp_detector.py logs the same stages per scene (:203 pre-processing, :209 "
                                              "prediction, :231 collision check, :251 importance sampling); its collision "
                                              "check is one launch sequence + host sync PER POSE",
                          "note": "not the headline metric; stage times of an unpipelined call (HIP events between stages)"}
            del det, raw, out16
            # ---- a batch that is NOT all "proven": 2 of the 16 scenes are `lattice` clouds (coordinates
            # snapped to a 3.9 mm lattice: exact distance ties), so the FPS prefix check refuses them and
            # the level-2 / level-3 samplers run inside the timed region (the headline's scenes all pass)
            n_tie = min(2, B)
            mix = synth.make_batch(scene_ids, args.points, variant=args.variant)
            mix[:n_tie] = synth.make_batch(scene_ids[:n_tie], args.points, variant="lattice")
            mix_batch = {"scene_points": torch.from_numpy(mix).to(dev)}
            x_steps, x_warm = 10, 3
            x_el, x_step, x_sum, _ = timed_region(x_steps, x_warm, collective=False, run=runner, data=mix_batch,
                                                  gathered=False)
            lv = [k for k in x_sum if k.startswith("fps[") and not k.startswith("fps[N=%d," % args.points)]
            mixed_batch = {"workload": "%d `lattice` + %d `%s` scenes per step: the tie-heavy scenes fail the FPS "
                                       "prefix proof, their level-2 / level-3 samplers run in the timed region"
                                       % (n_tie, B - n_tie, args.variant),
                           "value": round(B * x_steps / x_el, 2), "unit": "scenes/sec",
                           "ms_per_step": round(1e3 * x_el / x_steps, 3), "step_ms_median": x_step["median"],
                           "steps": x_steps, "warmup": x_warm,
                           "deeper_fps_launches_ms": {k: round(x_sum[k][1], 4) for k in sorted(lv)}}
            del mix_batch

    tb, src, why = load_traffic("contractions[step,B=%d,N=%d,precision=%s]" % (B, N, precision))
    if stub:
        roofline_dense = {"kernel": None, "bound": "mfma", "achieved": None, "peak": BF16_MFMA_PEAK_TF,
                          "unit": "TFLOP/s", "frac": None, "note": "test mode: no kernel ran"}
    elif roofline_dense is None:
        dense_tf = GFLOP_PER_SCENE * value / world / 1e3
        roofline_dense = {"kernel": "whole forward (library GEMMs), dense flops / wall time",
                          "bound": "mfma", "achieved": round(dense_tf, 2),
                          "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                          "frac": round(dense_tf / FP32_MFMA_PEAK_TF, 4)}
    roofline_dense["traffic"] = tb
    if tb is not None:
        roofline_dense["traffic_source"] = src
        roofline_dense["traffic_unit"] = "bytes per step (all contraction launches)"
    else:
        roofline_dense["traffic_note"] = why

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as _oracle
        from oracle import pn2_forward
        ncores = os.cpu_count() or 1
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        one = pts[:1].cpu().numpy()
        # torch's CPU convolutions do not scale to hundreds of threads on these short layers (256
        # threads measured 13x SLOWER than 8): time the scene at a few team sizes, report the best
        best = None
        for nt in sorted({min(ncores, 16), min(ncores, 64)}):
            torch.set_num_threads(nt)        # torch's own conv / BN thread pool
            used = _oracle.set_threads(nt)   # the C operators' OpenMP team (OMP_NUM_THREADS is only read
                                             # when the OpenMP runtime starts)
            t1 = time.perf_counter()
            ref = pn2_forward.forward(sd, one, cfg.num_centroids, cfg.radius, cfg.num_neighbours)
            dt = time.perf_counter() - t1
            if best is None or dt < best[0]:
                best = (dt, used)
        cpu_s, nt = best
        # the sample proper: up to six more scenes of the step's batch at the better team size, bounded to ~10 s
        torch.set_num_threads(nt)
        _oracle.set_threads(nt)
        n_cpu, t_cpu = 0, 0.0
        host_pts = pts.cpu().numpy()
        for i in range(min(B, 6)):
            t1 = time.perf_counter()
            ref_i = pn2_forward.forward(sd, host_pts[i:i + 1], cfg.num_centroids, cfg.radius, cfg.num_neighbours)
            t_cpu += time.perf_counter() - t1
            n_cpu += 1
            if i == 0:
                ref = ref_i
            if t_cpu > 10.0:
                break
        with torch.no_grad():
            got = runner({"scene_points": pts[:1]})
        err = max(float(np.max(np.abs(got[k].cpu().numpy() - ref[k]))) for k in heads)
        err_rel = max(float(np.max(np.abs(got[k].cpu().numpy() - ref[k]))) / max(1.0, float(np.abs(ref[k]).max()))
                      for k in heads)
        from tests.ref64 import forward64      # float64 arithmetic on the same weights and indices (test infrastructure)
        ref64 = forward64(sd, host_pts[:1], cfg.num_centroids, cfg.radius, cfg.num_neighbours)
        sc = {k: max(1.0, float(np.abs(ref64[k]).max())) for k in heads}
        err_gpu64 = max(float(np.max(np.abs(got[k].cpu().numpy().astype(np.float64) - ref64[k]))) / sc[k] for k in heads)
        err_cpu64 = max(float(np.max(np.abs(ref[k].astype(np.float64) - ref64[k]))) / sc[k] for k in heads)
        spread = min(float((ref[k].std(axis=2) / np.maximum(np.abs(ref[k]).max(axis=2), 1e-30)).min()) for k in heads)
        cpu_baseline = {"value": round(n_cpu / t_cpu, 4), "unit": "scenes/sec",
                        "cores": nt, "kind": "port",
                        "sample": "%d scenes (scenes %d..%d) of the same workload, one at a time: oracle C operators "
                                  "(OpenMP over centroids / queries, FPS scan over a team of <= 8) + "
                                  "torch CPU conv/BN, the better of %s threads (scene %d timed at both) on a %d-core host"
                                  % (n_cpu, scene_ids[0], scene_ids[0] + n_cpu - 1,
                                     sorted({min(ncores, 16), min(ncores, 64)}), scene_ids[0], ncores),
                        "seconds": round(t_cpu, 2), "scenes": n_cpu,
                        "max_abs_err_gpu_vs_cpu": err, "max_err_over_scale_gpu_vs_cpu": err_rel,
                        "max_err_over_scale_gpu_vs_float64": err_gpu64, "max_err_over_scale_cpu_vs_float64": err_cpu64,
                        "output_spread_min_std_over_max": round(spread, 4),
                        "err_note": "scene %d, %s weights, all four heads over all points; scale = max(1, max|ref|) per "
                                    "head; float64 = tests/ref64.py (the same composition in float64 on the same indices): "
                                    "the device and torch's CPU fp32 kernels are each that far from exact arithmetic; the spread says the outputs depend on the input (a per-channel constant "
                                    "would read 0)" % (scene_ids[0], args.weights)}

    arith = {"f16x2": "fp32-class: contraction as scaled 2xfp16 split, 3 MFMA products, fp32 accumulate",
             "bf16x3": "fp32-class: contraction as exact 3xbf16 split, fp32 accumulate",
             "bf16": "REDUCED PRECISION: plain bf16 contraction, fp32 accumulate",
             "fp32": "fp32 MFMA"}.get(precision, "library GEMM")
    pipe_label = (", pipelined: up to %d batches submitted besides the one being collected" % args.in_flight
                  if pipelined else ", one batch at a time")
    if gather is None:
        collective = {"op": None, "note": "single process: no collective"}
    else:
        collective = {"op": "all_gather_into_tensor (%s)" % ("RCCL" if hw.backend == "nccl" else hw.backend),
                      "payload": args.gather,
                      "payload_bytes_per_rank_per_step": int(gather.payload_bytes),
                      "stream": gather.last_stream,
                      "rccl_env": {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS") if k in os.environ},
                      "note": "issued on a side stream behind an event of the collecting stream; the next batch's "
                              "contractions run on their own stream meanwhile"}
    payload = ("all-gather of 21 ch/point" if args.gather == "heads" else
               "all-gather of the %d best decoded grasp frames per scene" % args.num_poses)
    line = {
        **({"test_mode": hw.test_mode} if hw.test_mode else {}),
        "metric": "scenes/sec (25.6k-pt clouds) end-to-end grasp inference",
        "value": round(value, 3), "unit": "scenes/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "f16x2": "f32 (f16x2 split)", "bf16x3": "f32 (bf16x3 split)"}.get(precision, "f32"),
        "data": "synthetic" if not stub else "synthetic clouds, stub outputs (test mode)",
        "config": {"workload": "S4G PN2_CLS forward (3 SA + 3 FP + 4 heads), %d scenes/GPU/step, "
                               "%d-pt %s clouds, %s, impl=%s%s" % (B, args.points, args.variant, arith,
                                                                   impl, pipe_label),
                   "weights": args.weights if not stub else None,
                   "scenes_per_gpu": B, "num_points": args.points, "global_batch": world * B,
                   "in_flight": args.in_flight if pipelined else 0,
                   "kernel_timers": "HIP event pairs around every native launch of every %s forward pass of the "
                                    "timed region" % ("" if args.timer_every == 1 else "%d-th" % args.timer_every),
                   "parallelism": "scenes sharded over %d GPU(s), %s" % (world, payload)},
        # `roofline`: the dominant kernel of the step (the MFMA contraction, >90 % of GPU time);
        # `roofline_ball_query_group_points`: the HBM-bound operator pair the north star names.
        "roofline": roofline_dense, "roofline_ball_query_group_points": roofline,
        "configs4": configs4, "modules_path": modules_path, "precision_legs": precision_legs,
        "weights_leg": weights_leg, "detect": detect_leg, "mixed_batch": mixed_batch, "kept_points": kept_points, "collective": collective, "distributed": shards,
        "step_ms": step_ms, "latency": latency, "io": io, "kernels": kernels,
        "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Measured board power and shader clock while ONE kernel of the default step runs back to back.

DESIGN.md argues that the big f16x2 contraction kernels are clocked down to the power budget
(GRBM_GUI_ACTIVE / wall time: 1.6-2.1 GHz instead of 2.4).  That was an inference from cycle
counters; this tool measures it: every native launch of one forward is recorded (descriptor
copies, buffers kept alive), then each selected launch is replayed in a loop for `--seconds`
while a sampler thread reads the driver's power / clock sensors.

Sensor sources, first one that answers:
  1. hwmon sysfs  /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average|power1_input,freq1_input}
  2. `amd-smi metric -p -c --json`
  3. `rocm-smi --showpower --showclocks --json`
Output: one markdown table on stdout (+ the raw samples' summary as JSON with --json).

    python tools/power_probe.py [--seconds 2.0] [--batch 16] [--points 25600] [--precision f16x2]
"""
import argparse
import ctypes
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sensors:
    def __init__(self, pci_bus=None):
        """pci_bus: bus number of the GPU the process uses (a host may expose every GPU's sensors
        in sysfs while the container sees one device)."""
        self.kind = None
        self.power_path = self.freq_path = self.cap_path = None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        if pci_bus is not None:
            def bus_of(hw):
                real = os.path.realpath(os.path.dirname(os.path.dirname(hw)))   # .../0000:bb:dd.f
                try:
                    return int(os.path.basename(real).split(":")[1], 16)
                except (IndexError, ValueError):
                    return None
            match = [hw for hw in cands if bus_of(hw) == pci_bus]
            cands = match or cands
        for hw in cands:
            for name in ("power1_average", "power1_input"):
                if _read(os.path.join(hw, name)) not in (None, ""):
                    self.power_path = os.path.join(hw, name)
                    break
            if self.power_path:
                f = os.path.join(hw, "freq1_input")
                self.freq_path = f if _read(f) not in (None, "") else None
                c = os.path.join(hw, "power1_cap")
                self.cap_path = c if _read(c) not in (None, "") else None
                self.kind = "hwmon:" + self.power_path
                break
        if self.kind is None:
            for cmd, kind in ((["amd-smi", "metric", "-p", "-c", "--json"], "amd-smi"),
                              (["rocm-smi", "--showpower", "--showclocks", "--json"], "rocm-smi")):
                try:
                    out = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                    if out.returncode == 0 and out.stdout.strip():
                        self.kind, self.cmd = kind, cmd
                        break
                except (OSError, subprocess.TimeoutExpired):
                    continue

    def describe(self):
        d = {"source": self.kind}
        if self.kind and self.kind.startswith("hwmon"):
            d["pci"] = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(os.path.dirname(self.power_path)))))
        if self.cap_path:
            d["power_cap_W"] = int(_read(self.cap_path)) / 1e6
        return d

    def sample(self):
        """(watts or None, sclk MHz or None)"""
        if self.kind and self.kind.startswith("hwmon"):
            p = _read(self.power_path)
            f = _read(self.freq_path) if self.freq_path else None
            return (int(p) / 1e6 if p else None, int(f) / 1e6 if f else None)
        if self.kind in ("amd-smi", "rocm-smi"):
            try:
                out = subprocess.run(self.cmd, capture_output=True, text=True, timeout=20).stdout
                return _parse_smi(out)
            except (OSError, subprocess.TimeoutExpired):
                return (None, None)
        return (None, None)


def _walk(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _walk(v, path + "/" + str(k))
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _walk(v, path + "/%d" % i)
    else:
        yield path, o


def _num(v):
    if isinstance(v, (int, float)):
        return float(v)
    try:
        return float(str(v).split()[0].strip("()MHzWw"))
    except (ValueError, IndexError):
        return None


def _parse_smi(text):
    try:
        j = json.loads(text)
    except ValueError:
        return (None, None)
    watts = mhz = None
    for path, v in _walk(j):
        pl = path.lower()
        if watts is None and ("socket_power" in pl or "average graphics package power" in pl or
                              "current socket graphics package power" in pl) and "unit" not in pl:
            watts = _num(v)
        if mhz is None and ("gfx_0/clk" in pl or "sclk clock speed" in pl) and "unit" not in pl:
            mhz = _num(v)
    return (watts, mhz)


class Sampler(threading.Thread):
    def __init__(self, sensors, period):
        super().__init__(daemon=True)
        self.s, self.period = sensors, period
        self.samples = []
        self._halt = threading.Event()

    def run(self):
        while not self._halt.is_set():
            self.samples.append(self.s.sample())
            time.sleep(self.period)

    def stop(self):
        self._halt.set()
        self.join()
        w = [a for a, _ in self.samples if a is not None]
        f = [b for _, b in self.samples if b is not None]
        mean = lambda xs: sum(xs) / len(xs) if xs else None
        return {"n": len(self.samples), "W_mean": mean(w), "W_max": max(w) if w else None,
                "MHz_mean": mean(f), "MHz_min": min(f) if f else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=25600)
    ap.add_argument("--precision", default="f16x2")
    ap.add_argument("--json", default=None)
    ap.add_argument("--weights", default="calibrated", choices=["calibrated", "randomized"],
                    help="calibrated (default): the golden run's network, activations that carry signal (bench.py's default); "
                         "randomized: rounds 1-5's per-channel-constant network")
    args = ap.parse_args()

    import torch
    from s4g_release_amd import _cabi, functions as F, synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_

    dev = torch.device("cuda:0")
    sensors = Sensors(getattr(torch.cuda.get_device_properties(0), "pci_bus_id", None))
    if args.weights == "calibrated":
        from tests import golden_util as GU
        net = GU.shipped_net(dev)
    else:
        torch.manual_seed(20260101)
        net = build_pointnet2_cls(S4GConfig())
        randomize_bn_(net, 20260102)
        net = net.to(dev).eval()
    runner = FusedPointNet2(net, precision=args.precision)
    pts = torch.from_numpy(synth.make_batch(list(range(args.batch)), args.points)).to(dev)
    batch = {"scene_points": pts}
    with torch.no_grad():
        for _ in range(2):
            runner(batch)
    torch.cuda.synchronize()

    # ---- record every contraction launch of one forward (descriptor copies; tensors kept alive
    # by holding the handle, and nothing is allocated afterwards)
    lib = _cabi.lib()
    recorded = []
    real_gemm, real_heads = lib.s4g_mlp_gemm_f32, lib.s4g_heads_chain_f32
    names = []
    real_timed_enter = F._timed.__enter__

    def timed_enter(self):
        names.append((self.name, self.flops))
        return real_timed_enter(self)
    F._timed.__enter__ = timed_enter

    class Rec:
        def __init__(self, fn, kind):
            self.fn, self.kind = fn, kind

        def __call__(self, dref, stream):
            d = type(dref._obj).from_buffer_copy(dref._obj)
            recorded.append((names[-1][0], names[-1][1], self.fn, d))
            return self.fn(dref, stream)

    lib.s4g_mlp_gemm_f32 = Rec(real_gemm, "gemm")
    lib.s4g_heads_chain_f32 = Rec(real_heads, "heads")
    with torch.no_grad():
        handle = runner.submit(batch)
        handle.result()
    torch.cuda.synchronize()
    lib.s4g_mlp_gemm_f32, lib.s4g_heads_chain_f32 = real_gemm, real_heads
    F._timed.__enter__ = real_timed_enter
    # (a launch with a data-dependent row count -- the distinct-row SA level -- reports its flops lazily)
    recorded = [(n, f() if callable(f) else f, fn, d) for n, f, fn, d in recorded]

    st = torch.cuda.current_stream().cuda_stream
    rows = []

    def measure(label, body, flops_per_iter):
        body()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # calibrate the iteration count for ~args.seconds
        e0.record()
        for _ in range(3):
            body()
        e1.record()
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 3
        n = max(5, int(args.seconds * 1e3 / max(per, 1e-3)))
        smp = Sampler(sensors, 0.01 if sensors.kind and sensors.kind.startswith("hwmon") else 0.2)
        e0.record()
        smp.start()
        for _ in range(n):
            body()
        e1.record()
        torch.cuda.synchronize()
        st_ = smp.stop()
        ms = e0.elapsed_time(e1) / n
        tf = 3.0 * flops_per_iter / ms / 1e9 if args.precision == "f16x2" else flops_per_iter / ms / 1e9
        rows.append(dict(kernel=label, ms=ms, executed_TFLOPs=tf, iters=n, **st_))

    # idle baseline
    smp = Sampler(sensors, 0.01 if sensors.kind and sensors.kind.startswith("hwmon") else 0.2)
    smp.start()
    time.sleep(1.0)
    rows.append(dict(kernel="(idle)", ms=None, executed_TFLOPs=None, iters=0, **smp.stop()))

    big = [r for r in recorded if r[1] >= 5e10]
    for name, flops, fn, d in big:
        measure(name, lambda fn=fn, d=d: fn(ctypes.byref(d), st), flops)
    small = [r for r in recorded if r[1] < 5e10]
    if small:
        def all_small():
            for _, _, fn, d in small:
                fn(ctypes.byref(d), st)
        measure("the %d small launches together" % len(small), all_small, sum(r[1] for r in small))

    def whole():
        with torch.no_grad():
            runner(batch)
    measure("whole forward, one batch at a time", whole, sum(r[1] for r in recorded))

    info = sensors.describe()
    print("sensor: %s" % json.dumps(info))
    print("| kernel | ms | executed TFLOP/s | W mean | W max | sclk MHz mean | sclk MHz min | samples |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    f = lambda v, p=1: "-" if v is None else ("%." + str(p) + "f") % v
    for r in rows:
        print("| %s | %s | %s | %s | %s | %s | %s | %d |" % (
            r["kernel"], f(r["ms"], 3), f(r["executed_TFLOPs"]), f(r["W_mean"]), f(r["W_max"]),
            f(r["MHz_mean"], 0), f(r["MHz_min"], 0), r["n"]))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"sensor": info, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()

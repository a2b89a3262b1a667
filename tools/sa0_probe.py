#!/usr/bin/env python3
"""Round-6 bounded attempt on the first SA level's chain launch (sa0.1+sa0.2, 1.07 ms of the 7.7 ms step): what do the
EMPTY tiles of the distinct-row form cost?  The launch is sized for the plain layout (B M K / 64 tiles); in the distinct-
row layout ~30 % of them find no rows and return at once.  A device-built tile list would remove exactly those.
Captures the launch's descriptor from a real forward and replays it (a) as it is, (b) with every scene's row count set
to 0 (all 81 920 tiles empty: the pure dispatch cost), (c) with the plain 64-row layout.  python tools/sa0_probe.py"""
import copy
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["S4G_TEST_KNOBS"] = "1"


def main():
    from s4g_release_amd import _cabi, synth
    from s4g_release_amd.fused import FusedPointNet2
    from tests import golden_util as GU
    dev = torch.device("cuda:0")
    net = GU.shipped_net(dev)
    pts = torch.from_numpy(synth.make_batch(list(range(16)), 25600)).to(dev)
    run = FusedPointNet2(net)
    keep = {}
    orig = run._gemm

    def spy(name, layer, P, loader, epi, **kw):
        if name == "sa0.1":
            keep["kw"] = dict(kw)
            real = _cabi.lib().s4g_mlp_gemm_f32

            class L:
                def __getattr__(self, n):
                    return getattr(_cabi.lib(), n)
            # copy the descriptor at the moment of the launch
            import s4g_release_amd.fused as Fz
            old = Fz._cabi.lib

            def lib():
                class P_:
                    def __getattr__(s, n):
                        if n == "s4g_mlp_gemm_f32":
                            def f(dref, st):
                                c = type(dref._obj)()
                                ctypes.memmove(ctypes.byref(c), dref, ctypes.sizeof(c))
                                keep["desc"] = c
                                return real(dref, st)
                            return f
                        return getattr(old(), n)
                return P_()
            Fz._cabi.lib = lib
            try:
                orig(name, layer, P, loader, epi, **kw)
            finally:
                Fz._cabi.lib = old
        else:
            orig(name, layer, P, loader, epi, **kw)
    run._gemm = spy
    with torch.no_grad():
        h = run.submit({"scene_points": pts})
        h.result()
    torch.cuda.synchronize()
    d = keep["desc"]
    rows = keep["kw"]["seg_rows"]
    used = rows.clone()
    rps = d.rows_per_scene
    tiles = d.P // 64
    nonempty = int(((used + 63) // 64).sum())
    print("P %d tiles %d rows_per_scene %d; rows used per scene: min %d mean %.0f max %d -> non-empty tiles %d (%.1f %%)"
          % (d.P, tiles, rps, int(used.min()), float(used.float().mean()), int(used.max()), nonempty, 100.0 * nonempty / tiles))
    st = torch.cuda.current_stream().cuda_stream

    def time(desc, reps=30):
        for _ in range(3):
            _cabi.check(_cabi.lib().s4g_mlp_gemm_f32(ctypes.byref(desc), st), "replay")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            _cabi.lib().s4g_mlp_gemm_f32(ctypes.byref(desc), st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_real = time(d)
    rows.zero_()
    t_empty = time(d)
    rows.copy_(used)
    half = used.clone()
    print("as launched          %.4f ms" % t_real)
    print("every tile empty     %.4f ms  (%d workgroups that read one word and return)" % (t_empty, tiles))
    print("=> the %d empty tiles of the real launch cost at most %.4f ms (%.1f %% of it) if their dispatch does not hide "
          "behind the working tiles at all" % (tiles - nonempty, t_empty * (tiles - nonempty) / tiles,
                                               100.0 * t_empty * (tiles - nonempty) / tiles / t_real))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Stand-alone timing of the shared-MLP contraction (plain loader, store epilogue).
Usage: python tools/bench_gemm.py [P K N] [--prec 0|1|2|3] [--reps 10]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import _cabi  # noqa: E402
from s4g_release_amd.fused import fragment_order, split_bf16x3, split_f16x2  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dims", nargs="*", type=int, default=[409600, 256, 1024])
    ap.add_argument("--prec", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dbg", type=int, default=0)
    a = ap.parse_args()
    P, K, N = a.dims
    dev = torch.device("cuda:0")
    A = torch.randn(P, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(P, N, device=dev)
    k16 = (K + 15) // 16 * 16
    w16 = W.new_zeros(N, k16)
    w16[:, :K] = W
    w3 = split_bf16x3(w16)
    d = _cabi.GemmDesc()
    d.loader, d.epilogue, d.groups, d.relu = 0, 0, 1, 1
    d.P, d.Cin, d.Kpad, d.Cout = P, K, K, N
    d.W, d.bias, d.A, d.lda, d.out, d.ldc = W.data_ptr(), b.data_ptr(), A.data_ptr(), K, out.data_ptr(), N
    d.precision, d.Kpad16, d.W_bf16x3 = a.prec, k16, w3.data_ptr()
    wh2, winv = split_f16x2(w16)
    amax_in = torch.zeros(64, device=dev)
    amax_in[0] = A.abs().max()
    amax_out = torch.zeros(64, device=dev)
    d.W_f16x2, d.w_inv_scale = wh2.data_ptr(), winv.data_ptr()
    d.a_amax, d.out_amax = amax_in.data_ptr(), amax_out.data_ptr()
    if N % 32 == 0:
        wfrag = fragment_order(wh2.view(2, 1, N, k16))
        d.W_f16x2_frag = wfrag.data_ptr()
    wh2, winv = split_f16x2(w16)
    amax_in = torch.zeros(64, device=dev)
    amax_in[0] = A.abs().max()
    amax_out = torch.zeros(64, device=dev)
    d.W_f16x2, d.w_inv_scale = wh2.data_ptr(), winv.data_ptr()
    d.a_amax, d.out_amax = amax_in.data_ptr(), amax_out.data_ptr()
    lib = _cabi.lib()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _cabi.check(lib.s4g_mlp_gemm_f32(ctypes.byref(d), st), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        lib.s4g_mlp_gemm_f32(ctypes.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("dbg=%d " % a.dbg, end="")
    print("P=%d K=%d N=%d prec=%d: %.3f ms  %.1f TFLOP/s (fp32-equivalent)" % (
        P, K, N, a.prec, ms, 2.0 * P * K * N / ms / 1e9))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""HBM streaming reference points on the box (torch fill / copy of 1 GiB): what a pure
write and a read+write stream reach, to put the gather kernels' GB/s into perspective."""
import torch
dev = torch.device("cuda:0")
n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timeit(lambda: a.fill_(1.0)); print("fill 1 GiB   %.3f ms  %.0f GB/s written" % (ms, 4 * n / ms / 1e6))
ms = timeit(lambda: b.copy_(a));   print("copy 1 GiB   %.3f ms  %.0f GB/s read + written" % (ms, 8 * n / ms / 1e6))

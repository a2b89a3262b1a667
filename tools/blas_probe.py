import torch, time
dev = torch.device("cuda:0")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (M, K, N) in ((409600, 256, 2048), (409600, 512, 1024), (5242880, 128, 256), (1048576, 256, 512), (409600, 256, 1024)):
    for dt, k3 in ((torch.float16, 3), (torch.float16, 1), (torch.float32, 1)):
        a = torch.randn(M, K * k3, device=dev, dtype=dt)
        w = torch.randn(N, K * k3, device=dev, dtype=dt)
        ms = t(lambda: torch.matmul(a, w.t()))
        print("M=%d K=%d(x%d) N=%d %s: %.3f ms  %.0f TFLOP/s executed, %.0f fp32-equivalent" % (
            M, K, k3, N, str(dt).split('.')[-1], ms, 2.0 * M * K * k3 * N / ms / 1e9, 2.0 * M * K * N / ms / 1e9))
        del a, w

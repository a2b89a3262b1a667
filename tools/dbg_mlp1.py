import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_fused_gpu import _run, _w3, _h2, _h2_second
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, M, K, C = 1, 8, 64, 128
P = B * M * K
rel = (torch.rand(M, 3, generator=g) - 0.5) * 0.4
rel4 = torch.zeros(P, 4)
rel4[:, :3] = rel.repeat_interleave(K, 0)
rel4 = rel4.to(dev)
w1 = torch.randn(C, 4, generator=g)
case = sys.argv[1] if len(sys.argv) > 1 else "full"
if case == "nobias": w1[:, 3] = 0
if case == "biasonly": w1[:, :3] = 0
if case == "xonly": w1[:, 1:] = 0
w1 = w1.to(dev)
print("case", case)
W = torch.eye(C).to(dev); b = torch.zeros(C, device=dev)
W2 = torch.eye(C).to(dev); b2 = torch.zeros(C, device=dev)
k16, w3 = _w3(W)
bound = float((w1[:, :3].abs().sum(1) * 0.2 + w1[:, 3].abs()).max())
h2 = _h2(W, floor=bound)
frag2, inv2 = _h2_second(W2)
ref = (rel.double().to(dev) @ w1[:, :3].double().t() + w1[:, 3].double()).clamp_min(0)
for mode in ("1",):
    os.environ["S4G_MLP1_MFMA"] = mode
    out = torch.full((B * M, C), float("nan"), device=dev)
    _run(dict(loader=3, epilogue=1, groups=1, relu=1, P=P, Cin=C, Kpad=C, Cout=C, W=W, bias=b, rel_xyz4=rel4,
              N=999, M=M, K=K, mlp1_w=w1, out=out, ldc=C, precision=3, Kpad16=k16, W_bf16x3=w3,
              W2_f16x2_frag=frag2, w2_inv_scale=inv2, bias2=b2, Cout2=C, relu2=1, **h2), dev)
    err = (out.double() - ref).abs()
    print("mode", mode, "max err", err.max().item(), "ref max", ref.max().item())
    print(" per-channel-block max err:", [err[:, i:i + 32].max().item() for i in range(0, C, 32)])
    print(" per-row max err:", err.max(1)[0].tolist())
    if mode == "1":
        i = err.argmax().item(); r, c = divmod(i, C)
        print(" worst", r, c, out[r, c].item(), ref[r, c].item(), "rel", rel[r].tolist(), "w", w1[c].tolist())

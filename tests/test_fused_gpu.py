"""GPU tests of the inference fast path (fp32-MFMA shared-MLP contraction with
fused group / interpolate / max / head epilogues).  Unit tests compare each
loader x epilogue against a plain fp32 torch restatement of the same layer;
model tests compare against the golden fixtures captured from the reference's
Python network (1e-4 abs; indices bit-exact)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests import golden_util as GU
from tests.conftest import with_variants

pytestmark = pytest.mark.gpu
TOL = 1e-4
# random-architecture tests: 6 seeds in the suite; S4G_FUZZ_SEEDS=n widens them into a sweep (run once per round)
FUZZ_SEEDS = int(os.environ.get("S4G_FUZZ_SEEDS", "6"))


@pytest.fixture(params=with_variants(["chain", "tiled"], ["resident"]))
def gemm_variant(request, monkeypatch):
    """Unit tests run per single-layer kernel: default dispatch (plain layers whose widths allow it on the
    chain kernel's first-layer machinery), the tiled kernel for every shape (S4G_GEMM_SINGLE_CHAIN=0) and --
    measurement builds only -- the resident-A kernel forced wherever it applies (S4G_GEMM_RESIDENT=1)."""
    if request.param == "resident":
        monkeypatch.setenv("S4G_GEMM_SINGLE_CHAIN", "0")
        monkeypatch.setenv("S4G_GEMM_RESIDENT", "1")
    elif request.param == "tiled":
        monkeypatch.setenv("S4G_GEMM_SINGLE_CHAIN", "0")
        monkeypatch.setenv("S4G_GEMM_RESIDENT", "0")
    else:
        monkeypatch.setenv("S4G_GEMM_SINGLE_CHAIN", "1")     # every shape the chain form supports, not only where it wins
    return request.param


def _run(desc_kwargs, dev):
    from s4g_release_amd import _cabi
    d = _cabi.GemmDesc()
    keep = []
    for k, v in desc_kwargs.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, k, v)
    rc = _cabi.lib().s4g_mlp_gemm_f32(ctypes.byref(d), torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "gemm")
    torch.cuda.synchronize()


def _w3(w):
    """(Kpad16, bf16x3 planes) for a (.., Cout, K) fp32 weight."""
    from s4g_release_amd.fused import split_bf16x3
    k = w.shape[-1]
    kp = (k + 15) // 16 * 16
    w16 = w.new_zeros(w.shape[:-1] + (kp,))
    w16[..., :k] = w
    return kp, split_bf16x3(w16)


PRECISIONS = [0, 1, 3]   # S4G_GEMM_FP32, S4G_GEMM_BF16X3, S4G_GEMM_F16X2


def _h2(w, *tensors, floor=0.0):
    """Descriptor fields of the f16x2 mode: scaled fp16 planes of W, per-channel
    inverse scales, one 64-slot amax row per input tensor (the slot position is
    arbitrary) and a zeroed out_amax row."""
    from s4g_release_amd.fused import fragment_order, split_f16x2
    k = w.shape[-1]
    kp = (k + 15) // 16 * 16
    w16 = w.new_zeros(w.shape[:-1] + (kp,))
    w16[..., :k] = w
    planes, inv = split_f16x2(w16)
    kw = dict(W_f16x2=planes, w_inv_scale=inv, a_amax_floor=float(floor),
              out_amax=torch.zeros(64, device=w.device))
    if w.shape[-2] % 32 == 0:     # enables the resident-A kernel where the shape qualifies
        p4 = planes if planes.dim() == 4 else planes.unsqueeze(1)
        kw["W_f16x2_frag"] = fragment_order(p4)
    for name, t in zip(("a_amax", "a_amax2"), [t for t in tensors if t is not None]):
        row = torch.zeros(64, device=w.device)
        row[17] = t.abs().max()
        kw[name] = row
    return kw


def _check_out_amax(kw, out):
    """out_amax is an upper bound of max|out| (rows past P contribute |bias|)."""
    got = kw["out_amax"].max().item()
    assert got >= out.abs().max().item()


def _padk(w):
    k = w.shape[-1]
    kp = (k + 7) // 8 * 8
    out = w.new_zeros(w.shape[:-1] + (kp,))
    out[..., :k] = w
    return out.contiguous()


@pytest.mark.parametrize("P,Cin,Cout,relu", [(128, 32, 128, True), (1000, 128, 256, True),
                                             (77, 260, 21, False), (4096, 1536, 1024, True),
                                             (300, 8, 130, True),
                                             # resident-A kernel shapes (K % 64 == 0, Cout % 128 == 0)
                                             (777, 256, 512, True), (300, 128, 128, False),
                                             (4100, 64, 256, True), (64, 256, 2048, True)])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_plain_store(dev, P, Cin, Cout, relu, prec, gemm_variant):
    g = torch.Generator(device="cpu").manual_seed(P + Cin)
    A = torch.randn(P, Cin, generator=g).to(dev)
    W = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    out = torch.full((P, Cout), float("nan"), device=dev)
    Wp = _padk(W)
    k16, w3 = _w3(W)
    h2 = _h2(W, A)
    _run(dict(loader=0, epilogue=0, groups=1, relu=int(relu), P=P, Cin=Cin, Kpad=Wp.shape[1],
              Cout=Cout, W=Wp, bias=b, A=A, lda=Cin, out=out, ldc=Cout, precision=prec,
              Kpad16=k16, W_bf16x3=w3, **h2), dev)
    ref = A.double() @ W.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    assert torch.isfinite(out).all()
    if prec == 3:
        _check_out_amax(h2, out)
    assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("P,Cin,split,Cout", [(1000, 256, 256, 768), (4096, 512, 512, 1536), (333, 256, 512, 768)])
@pytest.mark.parametrize("prec", [2, 3])
def test_gemm_two_output_tensors(dev, P, Cin, split, Cout, prec, gemm_variant):
    """Two layers that read the same input as ONE launch (descriptor fields out2 / ldc2 / split_n / out_amax2):
    every output channel bit-identical to the two separate launches of the same kernel, and the per-scene
    maxima of each tensor in its own row."""
    from s4g_release_amd.fused import fragment_order, split_bf16x3
    if gemm_variant == "resident":
        pytest.skip("the resident-A measurement kernel has one output tensor")
    g = torch.Generator(device="cpu").manual_seed(P + Cout)
    A = torch.randn(P, Cin, generator=g).to(dev)
    W = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    W[split:] *= 7.0                      # the halves have different per-channel weight scales
    zero = torch.zeros(Cout, device=dev)

    def launch(w, outs, amaxs, split_n=0):
        k16, w3 = _w3(w)
        h2 = _h2(w, A)
        h2["out_amax"] = amaxs[0]
        kw = dict(loader=0, epilogue=0, groups=1, relu=0, P=P, Cin=Cin, Kpad=Cin, Cout=w.shape[0], W=_padk(w),
                  bias=zero, A=A, lda=Cin, out=outs[0], ldc=outs[0].shape[1], precision=prec, Kpad16=k16,
                  W_bf16x3=w3, rows_per_scene=0, **h2)
        if prec == 2:
            kw["W_f16x2_frag"] = fragment_order(split_bf16x3(w)[:1].unsqueeze(1))[:, :, :, 0].contiguous()
        if split_n:
            kw.update(out2=outs[1], ldc2=outs[1].shape[1], split_n=split_n, out_amax2=amaxs[1])
        _run(kw, dev)

    both = [torch.full((P, split), float("nan"), device=dev), torch.full((P, Cout - split), float("nan"), device=dev)]
    am_both = [torch.zeros(64, device=dev), torch.zeros(64, device=dev)]
    launch(W, both, am_both, split_n=split)
    for half, (lo, hi) in enumerate(((0, split), (split, Cout))):
        alone, am = torch.full((P, hi - lo), float("nan"), device=dev), torch.zeros(64, device=dev)
        launch(W[lo:hi].contiguous(), [alone], [am])
        assert torch.equal(both[half], alone), half
        ref = A.double() @ W[lo:hi].double().t()
        tol = (2e-2 if prec == 2 else 1e-5) * max(1.0, ref.abs().max().item())
        assert (both[half].double() - ref).abs().max().item() < tol
        if prec == 3:
            assert am_both[half].max().item() >= both[half].abs().max().item() > 0
            assert am_both[half].max().item() == am.max().item()


def test_gemm_two_output_tensors_rejects_other_shapes(dev):
    from s4g_release_amd import _cabi
    A = torch.randn(256, 256, device=dev)
    W = torch.randn(768, 256, device=dev) / 16
    k16, w3 = _w3(W)
    out, out2 = torch.empty(256, 256, device=dev), torch.empty(256, 512, device=dev)
    base = dict(loader=0, epilogue=0, groups=1, relu=0, P=256, Cin=256, Kpad=256, Cout=768, W=W,
                bias=torch.zeros(768, device=dev), A=A, lda=256, out=out, ldc=256, precision=3, Kpad16=k16,
                W_bf16x3=w3, out2=out2, ldc2=512, out_amax2=torch.zeros(64, device=dev), **_h2(W, A))
    for bad in (dict(split_n=0), dict(split_n=128), dict(split_n=768), dict(precision=0), dict(precision=1),
                dict(split_n=256, epilogue=1, K=64), dict(split_n=256, c_coff=4)):
        with pytest.raises(RuntimeError):
            _run(dict(base, **bad), dev)
    _run(dict(base, split_n=256), dev)


@pytest.mark.parametrize("Cin,Cout", [(64, 48), (64, 128), (128, 256)])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_grouped_column_slices(dev, prec, Cin, Cout, gemm_variant):
    g = torch.Generator(device="cpu").manual_seed(3)
    P, G = 700, 4
    A = torch.randn(P, G * Cin, generator=g).to(dev)
    W = (torch.randn(G, Cout, Cin, generator=g) / 8).to(dev)
    b = torch.randn(G, Cout, generator=g).to(dev)
    out = torch.full((P, G * Cout), float("nan"), device=dev)
    k16, w3 = _w3(W)
    _run(dict(loader=0, epilogue=0, groups=G, relu=1, P=P, Cin=Cin, Kpad=Cin, Cout=Cout, W=W,
              bias=b, w_gstride=Cout * Cin, b_gstride=Cout, A=A, lda=G * Cin, a_gcol=Cin, out=out,
              ldc=G * Cout, c_gcol=Cout, precision=prec, Kpad16=k16, W_bf16x3=w3, **_h2(W, A)), dev)
    for i in range(G):
        ref = (A[:, i * Cin:(i + 1) * Cin].double() @ W[i].double().t() + b[i].double()).clamp_min(0)
        assert (out[:, i * Cout:(i + 1) * Cout].double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("K", [16, 32, 64])
@pytest.mark.parametrize("Cf", [0, 24, 64, 88])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_gather_max(dev, K, Cf, prec):
    g = torch.Generator(device="cpu").manual_seed(K + Cf)
    B, N, M, Cout = 2, 500, 37, 96
    xyz = torch.randn(B, 3, N, generator=g).to(dev)
    cidx = torch.randint(0, N, (B, M), generator=g)
    ctr = torch.stack([xyz[b][:, cidx[b]] for b in range(B)]).contiguous()
    gidx = torch.randint(0, N, (B, M, K), generator=g).int().to(dev)
    feat = torch.randn(B * N, Cf, generator=g).to(dev) if Cf else None
    Cin = Cf + 3
    W = (torch.randn(Cout, Cin, generator=g) / 2).to(dev)      # K order [feat, xyz]
    b = torch.randn(Cout, generator=g).to(dev)
    Wp = _padk(W)
    P = B * M * K
    out = torch.full((B * M, Cout), float("nan"), device=dev)
    k16, w3 = _w3(W)
    # random neighbour lists are not inside any ball: bound the xyz columns by the cloud's extent
    h2 = _h2(W, feat, floor=2 * xyz.abs().max().item())
    _run(dict(loader=1, epilogue=1, groups=1, relu=1, P=P, Cin=Cin, Kpad=Wp.shape[1], Cout=Cout,
              W=Wp, bias=b, gidx=gidx, feat=feat, xyz=xyz, ctr=ctr, Cf=Cf, N=N, M=M, K=K, out=out,
              ldc=Cout, precision=prec, Kpad16=k16, W_bf16x3=w3, **h2), dev)
    if prec == 3:
        _check_out_amax(h2, out)
    rows = []
    for bi in range(B):
        gi = gidx[bi].long()                                        # (M,K)
        rel = xyz[bi][:, gi] - ctr[bi][:, :, None]                  # (3,M,K)
        r = rel.permute(1, 2, 0)
        if Cf:
            f = feat.view(B, N, Cf)[bi][gi]                         # (M,K,Cf)
            r = torch.cat([f, r], dim=2)
        rows.append(r)
    A = torch.stack(rows).double()                                  # (B,M,K,Cin)
    ref = (A @ W.double().t() + b.double()).clamp_min(0).max(dim=2)[0].view(B * M, Cout)
    assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("Cin,Cout,K", [(128, 256, 64), (256, 512, 64), (64, 128, 64), (96, 160, 32)])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_plain_max(dev, Cin, Cout, K, prec, gemm_variant):
    """Last SA layer: max over the K neighbour rows of each centroid, then bias + ReLU."""
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout)
    groups_n = 37
    P = groups_n * K
    A = torch.randn(P, Cin, generator=g).to(dev)
    W = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    out = torch.full((groups_n, Cout), float("nan"), device=dev)
    Wp = _padk(W)
    k16, w3 = _w3(W)
    h2 = _h2(W, A)
    _run(dict(loader=0, epilogue=1, groups=1, relu=1, P=P, Cin=Cin, Kpad=Wp.shape[1], Cout=Cout,
              W=Wp, bias=b, A=A, lda=Cin, K=K, out=out, ldc=Cout, precision=prec, Kpad16=k16,
              W_bf16x3=w3, **h2), dev)
    ref = (A.double() @ W.double().t() + b.double()).clamp_min(0).view(groups_n, K, Cout).max(dim=1)[0]
    assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    if prec == 3:
        _check_out_amax(h2, out)


@pytest.mark.parametrize("C1,Cout,epi", [(128, 128, 0), (128, 256, 1), (32, 64, 0)])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_gather_mlp1(dev, C1, Cout, epi, prec, gemm_variant):
    """First (xyz-only) SA layer evaluated inside the second layer's loader."""
    g = torch.Generator(device="cpu").manual_seed(C1 + Cout + epi)
    B, N, M, K = 2, 400, 21, 64
    xyz = (torch.rand(B, 3, N, generator=g) * 0.2).to(dev)
    cidx = torch.randint(0, N, (B, M), generator=g)
    ctr = torch.stack([xyz[b][:, cidx[b]] for b in range(B)]).contiguous()
    gidx = torch.randint(0, N, (B, M, K), generator=g).int().to(dev)
    w1 = torch.randn(C1, 4, generator=g).to(dev)                   # wx, wy, wz, bias
    W = (torch.randn(Cout, C1, generator=g) / C1 ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    P = B * M * K
    rows = B * M if epi == 1 else P
    out = torch.full((rows, Cout), float("nan"), device=dev)
    Wp = _padk(W)
    k16, w3 = _w3(W)
    bound = float((w1[:, :3].abs().sum(1) * 0.4 + w1[:, 3].abs()).max())
    h2 = _h2(W, floor=bound)
    _run(dict(loader=3, epilogue=epi, groups=1, relu=1, P=P, Cin=C1, Kpad=Wp.shape[1], Cout=Cout,
              W=Wp, bias=b, gidx=gidx, xyz=xyz, ctr=ctr, N=N, M=M, K=K, mlp1_w=w1, out=out,
              ldc=Cout, precision=prec, Kpad16=k16, W_bf16x3=w3, **h2), dev)
    rel = torch.stack([xyz[bi][:, gidx[bi].long()] - ctr[bi][:, :, None] for bi in range(B)])  # (B,3,M,K)
    rel = rel.permute(0, 2, 3, 1).reshape(P, 3).double()
    A = (rel @ w1[:, :3].double().t() + w1[:, 3].double()).clamp_min(0)
    ref = (A @ W.double().t() + b.double()).clamp_min(0)
    if epi == 1:
        ref = ref.view(B * M, K, Cout).max(dim=1)[0]
    assert (out.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


def test_group_rel_xyz_and_pregathered_mlp1_loader(dev):
    """s4g_group_rel_xyz_i32 = group_points(xyz, idx) - centroid as (P, 4) records, and the MLP1
    loader reading them (rel_xyz4) gives bit-identical results to following gidx itself."""
    from s4g_release_amd import _cabi
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N, M, K, C1, Cout = 3, 700, 37, 64, 128, 256
    xyz = (torch.rand(B, 3, N, generator=g) * 0.3).to(dev)
    cidx = torch.randint(0, N, (B, M), generator=g)
    ctr = torch.stack([xyz[b][:, cidx[b]] for b in range(B)]).contiguous()
    gidx = torch.randint(0, N, (B, M, K), generator=g).int().to(dev)
    P = B * M * K
    rel4 = torch.full((P, 4), float("nan"), device=dev)
    rc = _cabi.lib().s4g_group_rel_xyz_i32(xyz.data_ptr(), ctr.data_ptr(), gidx.data_ptr(), B, N, M, K,
                                           rel4.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "group_rel_xyz")
    ref = torch.stack([xyz[bi][:, gidx[bi].long()] - ctr[bi][:, :, None] for bi in range(B)])   # (B,3,M,K)
    assert torch.equal(rel4[:, :3], ref.permute(0, 2, 3, 1).reshape(P, 3))
    assert (rel4[:, 3] == 0).all()
    w1 = torch.randn(C1, 4, generator=g).to(dev)
    W = (torch.randn(Cout, C1, generator=g) / C1 ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    Wp = _padk(W)
    k16, w3 = _w3(W)
    h2 = _h2(W, floor=float((w1[:, :3].abs().sum(1) * 0.6 + w1[:, 3].abs()).max()))
    outs = []
    for extra in (dict(gidx=gidx, xyz=xyz, ctr=ctr), dict(rel_xyz4=rel4)):
        out = torch.full((B * M, Cout), float("nan"), device=dev)
        _run(dict(loader=3, epilogue=1, groups=1, relu=1, P=P, Cin=C1, Kpad=Wp.shape[1], Cout=Cout,
                  W=Wp, bias=b, N=N, M=M, K=K, mlp1_w=w1, out=out, ldc=Cout, precision=3, Kpad16=k16,
                  W_bf16x3=w3, **h2, **extra), dev)
        outs.append(out)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_model_with_and_without_pregathered_rows(dev, monkeypatch):
    """FusedPointNet2 hands the first SA level its rows pre-gathered (S4G_REL_XYZ, default on):
    same outputs bit for bit as the loader following the indices itself."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    net = GU.shipped_net(dev)
    pts = torch.from_numpy(synth.make_batch([5, 6], 25600)).to(dev)
    monkeypatch.setenv("S4G_SA_UNIQUE", "0")     # (the distinct-row form has its own tests: test_sa_unique_gpu.py)
    mfma = FusedPointNet2(net)({"scene_points": pts})          # round 5: the 3 -> 128 layer on the matrix cores
    monkeypatch.setenv("S4G_MLP1_MFMA", "0")     # the first layer on the vector ALU, as the index-following loader does
    a = FusedPointNet2(net)({"scene_points": pts})
    monkeypatch.setenv("S4G_REL_XYZ", "0")
    b = FusedPointNet2(net)({"scene_points": pts})
    for k in a:
        assert torch.equal(a[k], b[k]), k
        err = (mfma[k] - a[k]).abs().max().item()              # the same layer as an f16x2 product: fp32-class, not bitwise
        # (calibrated weights: two fp32-class forms of the network differ by ~3e-5 of scale, each ~5e-5 from float64)
        assert err < TOL * max(1.0, a[k].abs().max().item()), (k, err)


def _h2_second(W2):
    """W2_f16x2_frag / w2_inv_scale of the layer fused behind (groups leading)."""
    from s4g_release_amd.fused import fragment_order, split_f16x2
    planes, inv = split_f16x2(W2.contiguous())
    p4 = planes if planes.dim() == 4 else planes.unsqueeze(1)
    return fragment_order(p4), inv


@pytest.mark.parametrize("C,Cout2,epi,groups,P", [
    (128, 256, 1, 1, 37 * 64),     # SA0 shape: plain loader, max over 64 rows
    (128, 128, 1, 1, 3 * 64),      # odd number of centroids: half-empty last workgroup
    (256, 512, 1, 1, 21 * 64),     # SA1 shape, 64 positions per workgroup
    (256, 256, 0, 1, 1000),        # FP2 shape, store epilogue, ragged last tile
    (128, 384, 0, 1, 777),         # 128-wide pair, store, three strips
    (256, 128, 0, 4, 500),         # head layers 2 + 3: four groups, partial strip (waves 2, 3 sit out)
    (256, 64, 0, 2, 130),          # quarter strip
    (512, 1024, 1, 1, 9 * 64),     # SA2 shape: eight-wave workgroup, 512-wide panel
    (512, 512, 0, 1, 333),         # 512-wide pair, store, ragged
])
def test_gemm_fused_layer_pair(dev, C, Cout2, epi, groups, P):
    """Two layers, one launch (intermediate in LDS, per-tile scale): against fp64."""
    g = torch.Generator(device="cpu").manual_seed(C + Cout2 + groups)
    A = torch.randn(P, groups * C, generator=g).to(dev)
    A[: P // 3] *= 1e-3                                   # tiles with very different magnitudes
    W1 = (torch.randn(groups, C, C, generator=g) / C ** 0.5).to(dev)
    b1 = torch.randn(groups, C, generator=g).to(dev)
    W2 = (torch.randn(groups, Cout2, C, generator=g) / C ** 0.5).to(dev)
    b2 = torch.randn(groups, Cout2, generator=g).to(dev)
    K = 64
    rows = P // K if epi == 1 else P
    out = torch.full((rows, groups * Cout2), float("nan"), device=dev)
    k16, w3 = _w3(W1)
    h2 = _h2(W1, A)
    frag2, inv2 = _h2_second(W2)
    _run(dict(loader=0, epilogue=epi, groups=groups, relu=1, P=P, Cin=C, Kpad=C, Cout=C, W=W1, bias=b1,
              w_gstride=C * C, b_gstride=C, A=A, lda=groups * C, a_gcol=C, K=K, out=out,
              ldc=groups * Cout2, c_gcol=Cout2, precision=3, Kpad16=k16, W_bf16x3=w3,
              W2_f16x2_frag=frag2, w2_inv_scale=inv2, bias2=b2, Cout2=Cout2, relu2=1, **h2), dev)
    ref = []
    for gi in range(groups):
        h = (A[:, gi * C:(gi + 1) * C].double() @ W1[gi].double().t() + b1[gi].double()).clamp_min(0)
        o = (h @ W2[gi].double().t() + b2[gi].double()).clamp_min(0)
        ref.append(o)
    ref = torch.cat(ref, dim=1)
    if epi == 1:
        ref = ref.view(rows, K, Cout2).max(dim=1)[0]
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    _check_out_amax(h2, out)


@pytest.mark.parametrize("C,Cout3,epi,P", [(256, 2048, 0, 700), (128, 320, 0, 300), (256, 512, 1, 5 * 64)])
def test_gemm_fused_three_layer_chain(dev, C, Cout3, epi, P):
    """Three layers, one launch: two intermediates stay in LDS (C -> C -> C -> Cout3)."""
    g = torch.Generator(device="cpu").manual_seed(C + Cout3)
    A = torch.randn(P, C, generator=g).to(dev)
    Ws = [(torch.randn(n, C, generator=g) / C ** 0.5).to(dev) for n in (C, C, Cout3)]
    bs = [torch.randn(n, generator=g).to(dev) for n in (C, C, Cout3)]
    K = 64
    rows = P // K if epi == 1 else P
    out = torch.full((rows, Cout3), float("nan"), device=dev)
    k16, w3 = _w3(Ws[0])
    h2 = _h2(Ws[0], A)
    f2, i2 = _h2_second(Ws[1])
    f3, i3 = _h2_second(Ws[2])
    _run(dict(loader=0, epilogue=epi, groups=1, relu=1, P=P, Cin=C, Kpad=C, Cout=C, W=Ws[0], bias=bs[0],
              A=A, lda=C, K=K, out=out, ldc=Cout3, precision=3, Kpad16=k16, W_bf16x3=w3,
              W2_f16x2_frag=f2, w2_inv_scale=i2, bias2=bs[1], Cout2=C, relu2=1,
              W3_f16x2_frag=f3, w3_inv_scale=i3, bias3=bs[2], Cout3=Cout3, relu3=1, **h2), dev)
    ref = A.double()
    for W, b in zip(Ws, bs):
        ref = (ref @ W.double().t() + b.double()).clamp_min(0)
    if epi == 1:
        ref = ref.view(rows, K, Cout3).max(dim=1)[0]
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 4e-5 * max(1.0, ref.abs().max().item())
    _check_out_amax(h2, out)


@pytest.mark.parametrize("groups,P,Cout3", [(4, 333, 128), (1, 1000, 256)])
def test_gemm_fused_chain_deep_first_layer(dev, groups, P, Cout3):
    """Head layers 1 + 2 + 3: the first layer contracts over 2 C = 512 inputs (two panel
    loads through the same accumulators), then two more layers; four column groups."""
    C, K1 = 256, 512
    g = torch.Generator(device="cpu").manual_seed(groups + P)
    A = torch.randn(P, groups * K1, generator=g).to(dev)
    W1 = (torch.randn(groups, C, K1, generator=g) / K1 ** 0.5).to(dev)
    W2 = (torch.randn(groups, C, C, generator=g) / C ** 0.5).to(dev)
    W3 = (torch.randn(groups, Cout3, C, generator=g) / C ** 0.5).to(dev)
    b1, b2, b3 = (torch.randn(groups, n, generator=g).to(dev) for n in (C, C, Cout3))
    out = torch.full((P, groups * Cout3), float("nan"), device=dev)
    k16, w3 = _w3(W1)
    h2 = _h2(W1, A)
    f2, i2 = _h2_second(W2)
    f3, i3 = _h2_second(W3)
    _run(dict(loader=0, epilogue=0, groups=groups, relu=1, P=P, Cin=K1, Kpad=K1, Cout=C, W=W1, bias=b1,
              w_gstride=C * K1, b_gstride=C, A=A, lda=groups * K1, a_gcol=K1, out=out,
              ldc=groups * Cout3, c_gcol=Cout3, precision=3, Kpad16=k16, W_bf16x3=w3,
              W2_f16x2_frag=f2, w2_inv_scale=i2, bias2=b2, Cout2=C, relu2=1,
              W3_f16x2_frag=f3, w3_inv_scale=i3, bias3=b3, Cout3=Cout3, relu3=1, **h2), dev)
    ref = []
    for gi in range(groups):
        h = (A[:, gi * K1:(gi + 1) * K1].double() @ W1[gi].double().t() + b1[gi].double()).clamp_min(0)
        h = (h @ W2[gi].double().t() + b2[gi].double()).clamp_min(0)
        ref.append((h @ W3[gi].double().t() + b3[gi].double()).clamp_min(0))
    ref = torch.cat(ref, dim=1)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 4e-5 * max(1.0, ref.abs().max().item())


def test_gemm_fused_layer_pair_mlp1_loader(dev):
    """The SA0 launch: xyz gather + first layer in the loader, two contractions, max."""
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N, M, K, C, Cout2 = 2, 500, 23, 64, 128, 256
    xyz = (torch.rand(B, 3, N, generator=g) * 0.2).to(dev)
    cidx = torch.randint(0, N, (B, M), generator=g)
    ctr = torch.stack([xyz[b][:, cidx[b]] for b in range(B)]).contiguous()
    gidx = torch.randint(0, N, (B, M, K), generator=g).int().to(dev)
    w1 = torch.randn(C, 4, generator=g).to(dev)
    W = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    W2 = (torch.randn(Cout2, C, generator=g) / C ** 0.5).to(dev)
    b2 = torch.randn(Cout2, generator=g).to(dev)
    P = B * M * K
    out = torch.full((B * M, Cout2), float("nan"), device=dev)
    k16, w3 = _w3(W)
    bound = float((w1[:, :3].abs().sum(1) * 0.4 + w1[:, 3].abs()).max())
    h2 = _h2(W, floor=bound)
    frag2, inv2 = _h2_second(W2)
    _run(dict(loader=3, epilogue=1, groups=1, relu=1, P=P, Cin=C, Kpad=C, Cout=C, W=W, bias=b, gidx=gidx,
              xyz=xyz, ctr=ctr, N=N, M=M, K=K, mlp1_w=w1, out=out, ldc=Cout2, precision=3, Kpad16=k16,
              W_bf16x3=w3, W2_f16x2_frag=frag2, w2_inv_scale=inv2, bias2=b2, Cout2=Cout2, relu2=1,
              **h2), dev)
    rel = torch.stack([xyz[bi][:, gidx[bi].long()] - ctr[bi][:, :, None] for bi in range(B)])
    rel = rel.permute(0, 2, 3, 1).reshape(P, 3).double()
    A = (rel @ w1[:, :3].double().t() + w1[:, 3].double()).clamp_min(0)
    h = (A @ W.double().t() + b.double()).clamp_min(0)
    ref = (h @ W2.double().t() + b2.double()).clamp_min(0).view(B * M, K, Cout2).max(dim=1)[0]
    assert (out.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,M,C,Cout2,scale,bias_scale", [(2, 23, 128, 256, 0.2, 1.0), (1, 23, 128, 256, 0.2, 1.0),
                                                          (2, 9, 256, 512, 0.03, 1.0), (1, 8, 128, 128, 1e-6, 40.0),
                                                          (1, 6, 128, 256, 300.0, 1e-3), (1, 4, 128, 256, 0.0, 1.0)])
def test_chain_first_layer_on_the_matrix_cores(dev, monkeypatch, B, M, C, Cout2, scale, bias_scale):
    """Round 5: with pre-gathered (xyz_j - ctr_m, 0) records the 3 -> C first layer of an SA level is ONE 16-deep MFMA
    step inside the chain kernel (operands split into scaled fp16 planes like every other layer: the wave's measured
    coordinate maximum and each channel's own weight scale) instead of 3 FMAs + ReLU per element in the loader.
    Against float64 and against the vector-ALU loader (S4G_MLP1_MFMA=0); a ragged last tile (P % 128 != 0), tiny and
    huge coordinates, large biases, all-zero records."""
    g = torch.Generator(device="cpu").manual_seed(50 + M)
    K = 64
    P = B * M * K
    rel4 = torch.zeros(P, 4)
    rel4[:, :3] = (torch.rand(P, 3, generator=g) - 0.5) * 2 * scale
    rel4[5, :3] = 0.0
    rel4 = rel4.to(dev)
    w1 = torch.randn(C, 4, generator=g)
    w1[:, 3] *= bias_scale
    w1[3] = 0.0                                # a dead channel
    w1 = w1.to(dev)
    W = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    W2 = (torch.randn(Cout2, C, generator=g) / C ** 0.5).to(dev)
    b2 = torch.randn(Cout2, generator=g).to(dev)
    k16, w3 = _w3(W)
    bound = float((w1[:, :3].abs().sum(1) * max(scale, 1e-30) + w1[:, 3].abs()).max())
    h2 = _h2(W, floor=bound)
    frag2, inv2 = _h2_second(W2)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("S4G_MLP1_MFMA", mode)
        out = torch.full((B * M, Cout2), float("nan"), device=dev)
        _run(dict(loader=3, epilogue=1, groups=1, relu=1, P=P, Cin=C, Kpad=C, Cout=C, W=W, bias=b, rel_xyz4=rel4,
                  N=999, M=M, K=K, mlp1_w=w1, out=out, ldc=Cout2, precision=3, Kpad16=k16, W_bf16x3=w3,
                  W2_f16x2_frag=frag2, w2_inv_scale=inv2, bias2=b2, Cout2=Cout2, relu2=1, **h2), dev)
        outs[mode] = out
    A = (rel4[:, :3].double() @ w1[:, :3].double().t() + w1[:, 3].double()).clamp_min(0)
    h = (A @ W.double().t() + b.double()).clamp_min(0)
    ref = (h @ W2.double().t() + b2.double()).clamp_min(0).view(B * M, K, Cout2).max(dim=1)[0]
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    e1 = (outs["1"].double() - ref).abs().max().item()
    e0 = (outs["0"].double() - ref).abs().max().item()
    assert torch.isfinite(outs["1"]).all() and e1 < tol and e0 < tol, (e1, e0, tol)
    assert e1 < 4 * max(e0, 1e-7 * max(1.0, ref.abs().max().item())), (e1, e0)     # as accurate as the fp32 FMA chain


def test_gemm_chain_supported_query_matches_dispatch(dev):
    """s4g_gemm_chain_supported answers exactly what s4g_mlp_gemm_f32 accepts."""
    from s4g_release_amd import _cabi
    lib = _cabi.lib()
    assert lib.s4g_gemm_chain_supported(0, 0, 256, 512) == 1      # plain, store, deep first layer
    assert lib.s4g_gemm_chain_supported(3, 1, 256, 256) == 1      # MLP1 loader, 256 wide
    assert lib.s4g_gemm_chain_supported(4, 1, 512, 512) == 1      # eight-wave form
    assert lib.s4g_gemm_chain_supported(5, 0, 256, 256) == 1
    assert lib.s4g_gemm_chain_supported(1, 1, 256, 256) == 0      # plain GATHER loader: never fused
    assert lib.s4g_gemm_chain_supported(0, 0, 64, 64) == 0
    assert lib.s4g_gemm_chain_supported(0, 1, 256, 512) == 0      # deep first layer only with STORE
    assert lib.s4g_gemm_chain_supported(0, 0, 512, 1024) == 0
    for loader, epi, C, k16, P in [(0, 0, 128, 128, 200), (0, 1, 512, 512, 128), (0, 0, 256, 512, 70)]:
        g = torch.Generator(device="cpu").manual_seed(C)
        A = torch.randn(P, k16, generator=g).to(dev)
        W1 = (torch.randn(C, k16, generator=g) / k16 ** 0.5).to(dev)
        W2 = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
        b = torch.zeros(C, device=dev)
        out = torch.empty((P // 64 if epi else P, C), device=dev)
        kk, w3 = _w3(W1)
        h2 = _h2(W1, A)
        f2, i2 = _h2_second(W2)
        _run(dict(loader=loader, epilogue=epi, groups=1, relu=1, P=P, Cin=k16, Kpad=k16, Cout=C, W=W1,
                  bias=b, A=A, lda=k16, K=64, out=out, ldc=C, precision=3, Kpad16=kk, W_bf16x3=w3,
                  W2_f16x2_frag=f2, w2_inv_scale=i2, bias2=b, Cout2=C, relu2=1, **h2), dev)


def test_gemm_fused_layer_pair_rejects_unsupported(dev):
    from s4g_release_amd import _cabi
    W1 = torch.randn(64, 64, device=dev)
    k16, w3 = _w3(W1)
    h2 = _h2(W1, torch.ones(1, device=dev))
    frag2, inv2 = _h2_second(torch.randn(64, 64, device=dev))
    A = torch.randn(64, 64, device=dev)
    out = torch.empty(64, 64, device=dev)
    with pytest.raises(RuntimeError):    # C = 64 is not a fused width
        _run(dict(loader=0, epilogue=0, groups=1, relu=1, P=64, Cin=64, Kpad=64, Cout=64, W=W1,
                  bias=torch.zeros(64, device=dev), A=A, lda=64, out=out, ldc=64, precision=3, Kpad16=k16,
                  W_bf16x3=w3, W2_f16x2_frag=frag2, w2_inv_scale=inv2, bias2=torch.zeros(64, device=dev),
                  Cout2=64, relu2=1, **h2), dev)


@pytest.mark.parametrize("Cin,Cout,epi,K", [(256, 256, 0, 64), (64, 160, 1, 32), (512, 512, 0, 64),
                                            (128, 128, 1, 16)])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_gather_add_loader(dev, Cin, Cout, epi, K, prec):
    """First SA layer applied per point before the grouping: the loader gathers that row, adds
    the xyz part + bias and applies the ReLU (A = relu(F[idx] + W_xyz . (xyz - ctr) + b))."""
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout + K)
    B, N, M = 2, 300, 19
    xyz = (torch.rand(B, 3, N, generator=g) * 0.2).to(dev)
    cidx = torch.randint(0, N, (B, M), generator=g)
    ctr = torch.stack([xyz[b][:, cidx[b]] for b in range(B)]).contiguous()
    gidx = torch.randint(0, N, (B, M, K), generator=g).int().to(dev)
    F = torch.randn(B * N, Cin, generator=g).to(dev)
    w1 = torch.randn(Cin, 4, generator=g).to(dev)
    W = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    P = B * M * K
    rows = B * M if epi == 1 else P
    out = torch.full((rows, Cout), float("nan"), device=dev)
    Wp = _padk(W)
    k16, w3 = _w3(W)
    bound = float((w1[:, :3].abs().sum(1) * 0.4 + w1[:, 3].abs()).max())
    h2 = _h2(W, F, floor=bound)
    _run(dict(loader=4, epilogue=epi, groups=1, relu=1, P=P, Cin=Cin, Kpad=Wp.shape[1], Cout=Cout, W=Wp,
              bias=b, gidx=gidx, feat=F, Cf=Cin, xyz=xyz, ctr=ctr, N=N, M=M, K=K, mlp1_w=w1, out=out,
              ldc=Cout, precision=prec, Kpad16=k16, W_bf16x3=w3, **h2), dev)
    rel = torch.stack([xyz[bi][:, gidx[bi].long()] - ctr[bi][:, :, None] for bi in range(B)])
    rel = rel.permute(0, 2, 3, 1).reshape(P, 3).double()
    rows_f = torch.cat([F.view(B, N, Cin)[bi][gidx[bi].long().reshape(-1)] for bi in range(B)]).double()
    A = (rows_f + rel @ w1[:, :3].double().t() + w1[:, 3].double()).clamp_min(0)
    ref = (A @ W.double().t() + b.double()).clamp_min(0)
    if epi == 1:
        ref = ref.view(B * M, K, Cout).max(dim=1)[0]
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("C,Cout,with_y,chain", [(256, 256, False, True), (128, 128, True, True),
                                                   (64, 96, True, False), (512, 512, False, False)])
def test_gemm_interp_add_loader(dev, C, Cout, with_y, chain):
    """Interpolate + add + bias + ReLU formed by the loader of the next launch (tiled kernel
    or fused chain): A = relu(y + b0 + sum_k w_k S[idx_k]), then one or two more layers."""
    g = torch.Generator(device="cpu").manual_seed(C + Cout)
    B, N1, N2 = 2, 333, 40
    P = B * N1
    S = torch.randn(B * N2, C, generator=g).to(dev)
    y = torch.randn(P, C, generator=g).to(dev) if with_y else None
    b0 = torch.randn(C, generator=g).to(dev)
    nidx = torch.randint(0, N2, (B, N1, 3), generator=g).int().to(dev)
    nw = torch.rand(B, N1, 3, generator=g)
    nw = (nw / nw.sum(dim=2, keepdim=True)).to(dev)
    W = (torch.randn(Cout, C, generator=g) / C ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    W2 = (torch.randn(Cout, Cout, generator=g) / Cout ** 0.5).to(dev)
    b2 = torch.randn(Cout, generator=g).to(dev)
    out = torch.full((P, Cout), float("nan"), device=dev)
    Wp = _padk(W)
    k16, w3 = _w3(W)
    h2 = _h2(W, S, y, floor=float(b0.abs().max()))
    kw = dict(loader=5, epilogue=0, groups=1, relu=1, P=P, Cin=C, Kpad=Wp.shape[1], Cout=Cout, W=Wp, bias=b,
              nidx=nidx, nw=nw, sparse=S, dense=y, C2=C, N2=N2, N1=N1, loader_bias=b0, out=out, ldc=Cout,
              precision=3, Kpad16=k16, W_bf16x3=w3, **h2)
    if chain:
        f2, i2 = _h2_second(W2)
        kw.update(W2_f16x2_frag=f2, w2_inv_scale=i2, bias2=b2, Cout2=Cout, relu2=1)
    _run({k: v for k, v in kw.items() if v is not None}, dev)
    rows = torch.stack([S.view(B, N2, C)[bi][nidx[bi].long()] for bi in range(B)]).double()   # (B,N1,3,C)
    A = (rows * nw.double()[..., None]).sum(dim=2).view(P, C) + b0.double()
    if y is not None:
        A = A + y.double()
    A = A.clamp_min(0)
    ref = (A @ W.double().t() + b.double()).clamp_min(0)
    if chain:
        ref = (ref @ W2.double().t() + b2.double()).clamp_min(0)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 4e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("C,with_y,relu", [(256, True, 1), (512, False, 1), (36, True, 0), (1024, True, 1)])
def test_interp_add_channels_last(dev, C, with_y, relu):
    """out = act(y + bias + sum_k w_k * sparse[idx_k]) on channels-last tensors + its amax row."""
    from s4g_release_amd import _cabi
    g = torch.Generator(device="cpu").manual_seed(C)
    B, N1, N2 = 3, 777, 50
    sp = torch.randn(B * N2, C, generator=g).to(dev)
    y = torch.randn(B * N1, C, generator=g).to(dev) if with_y else None
    bias = torch.randn(C, generator=g).to(dev)
    nidx = torch.randint(0, N2, (B, N1, 3), generator=g).int().to(dev)
    nw = torch.rand(B, N1, 3, generator=g).to(dev)
    out = torch.full((B * N1, C), float("nan"), device=dev)
    amax = torch.zeros((B, 64), device=dev)          # one 64-slot row per scene
    rc = _cabi.lib().s4g_interp_add_cl_f32(None if y is None else y.data_ptr(), sp.data_ptr(),
                                           nidx.data_ptr(), nw.data_ptr(), bias.data_ptr(), B, N1, N2, C,
                                           relu, out.data_ptr(), amax.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "interp_add")
    torch.cuda.synchronize()
    rows = torch.stack([sp.view(B, N2, C)[bi][nidx[bi].long()] for bi in range(B)])     # (B,N1,3,C)
    ref = (rows.double() * nw.double()[..., None]).sum(dim=2).view(B * N1, C) + bias.double()
    if y is not None:
        ref = ref + y.double()
    if relu:
        ref = ref.clamp_min(0)
    assert (out.double() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert amax.view(torch.int32).max().item() > 0
    per_scene = out.view(B, N1, C).abs().amax(dim=(1, 2))
    assert (amax.amax(dim=1) >= per_scene).all()                 # every scene's row bounds its rows
    # a block that straddles two scenes feeds both; otherwise a scene's row is its own maximum
    assert (amax.amax(dim=1) <= out.abs().max()).all()


@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_interp_store(dev, prec):
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N1, N2, C2, C1, Cout = 2, 333, 50, 68, 32, 72
    sparse = torch.randn(B * N2, C2, generator=g).to(dev)
    dense = torch.randn(B * N1, C1, generator=g).to(dev)
    nidx = torch.randint(0, N2, (B, N1, 3), generator=g).int().to(dev)
    nw = torch.rand(B, N1, 3, generator=g).to(dev)
    W = (torch.randn(Cout, C2 + C1, generator=g) / 8).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    P = B * N1
    for c1, dn in ((C1, dense), (0, None)):
        Wc = W[:, :C2 + c1].contiguous()
        out = torch.full((P, Cout), float("nan"), device=dev)
        k16, w3 = _w3(Wc)
        Wp = _padk(Wc)
        _run(dict(loader=2, epilogue=0, groups=1, relu=1, P=P, Cin=C2 + c1, Kpad=Wp.shape[1], Cout=Cout,
                  W=Wp, bias=b, nidx=nidx, nw=nw, sparse=sparse, dense=dn, C2=C2, C1=c1, N2=N2,
                  N1=N1, out=out, ldc=Cout, precision=prec, Kpad16=k16, W_bf16x3=w3,
                  **_h2(Wc, sparse, dn)), dev)
        sp = sparse.view(B, N2, C2)
        interp = torch.stack([(sp[bi][nidx[bi].long()] * nw[bi][:, :, None]).sum(1)
                              for bi in range(B)]).view(P, C2)
        A = interp if c1 == 0 else torch.cat([interp, dense], dim=1)
        ref = (A.double() @ Wc.double().t() + b.double()).clamp_min(0)
        assert (out.double() - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("prec", PRECISIONS)
def test_gemm_channel_first_heads(dev, prec):
    g = torch.Generator(device="cpu").manual_seed(6)
    B, N, Cin = 2, 404, 64
    A = torch.randn(B * N, Cin, generator=g).to(dev)
    chans = [3, 9, 4, 5]
    W = (torch.randn(sum(chans), Cin, generator=g) / 8).to(dev)
    b = torch.randn(sum(chans), generator=g).to(dev)
    outs = [torch.full((B, c, N), float("nan"), device=dev) for c in chans]
    starts = [0, 3, 12, 16, 21]
    k16, w3 = _w3(W)
    _run(dict(loader=0, epilogue=2, groups=1, relu=0, P=B * N, Cin=Cin, Kpad=Cin, Cout=21, W=W,
              bias=b, A=A, lda=Cin, cf_ptr=(ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs]),
              cf_start=(ctypes.c_int32 * 5)(*starts), cf_sigmoid_from=16, cf_N=N, precision=prec,
              Kpad16=k16, W_bf16x3=w3, **_h2(W, A)), dev)
    ref = (A.double() @ W.double().t() + b.double()).view(B, N, 21).permute(0, 2, 1)
    for h, o in enumerate(outs):
        r = ref[:, starts[h]:starts[h + 1]]
        if h == 3:
            r = torch.sigmoid(r)
        assert (o.double() - r).abs().max().item() < 2e-5


def test_bf16x3_error_is_fp32_class(dev):
    """The split-precision kernel against fp64, next to the exact-fp32 MFMA kernel
    on the same operands (K = 1024, |a|,|w| ~ 1): both must sit at fp32 round-off."""
    g = torch.Generator(device="cpu").manual_seed(11)
    P, Cin, Cout = 2048, 1024, 256
    A = torch.randn(P, Cin, generator=g).to(dev)
    W = torch.randn(Cout, Cin, generator=g).to(dev)
    b = torch.zeros(Cout, device=dev)
    ref = A.double() @ W.double().t()
    scale = (A.double().abs() @ W.double().abs().t())       # sum |a||w|
    errs = {}
    k16, w3 = _w3(W)
    for prec in PRECISIONS:
        out = torch.empty(P, Cout, device=dev)
        _run(dict(loader=0, epilogue=0, groups=1, relu=0, P=P, Cin=Cin, Kpad=Cin, Cout=Cout, W=W,
                  bias=b, A=A, lda=Cin, out=out, ldc=Cout, precision=prec, Kpad16=k16,
                  W_bf16x3=w3, **_h2(W, A)), dev)
        errs[prec] = ((out.double() - ref).abs() / scale).max().item()
    print("max |err| / sum|a||w|: fp32 MFMA %.3g, bf16x3 %.3g, f16x2 %.3g" % (errs[0], errs[1], errs[3]))
    assert errs[0] < 1e-6 and errs[1] < 1e-6 and errs[3] < 1e-6   # fp32 unit round-off is 6e-8; K = 1024 terms
    assert errs[1] < 4 * errs[0] + 6e-8
    assert errs[3] < 4 * errs[0] + 6e-8


def test_f16x2_wide_dynamic_range(dev):
    """Per-tensor activation scale: rows 2^-12 below the tensor's maximum keep
    fp32-class accuracy relative to their own sum|a||w|; only below 2^-18 of the
    maximum do low-order bits start to go (fp16 subnormals of the second plane)."""
    g = torch.Generator(device="cpu").manual_seed(12)
    P, Cin, Cout = 1024, 256, 128
    A = torch.randn(P, Cin, generator=g)
    A[P // 2:] *= 2.0 ** -12
    A = A.to(dev)
    W = (torch.randn(Cout, Cin, generator=g) * torch.logspace(-3, 1, Cout)[:, None]).to(dev)
    b = torch.zeros(Cout, device=dev)
    out = torch.empty(P, Cout, device=dev)
    k16, w3 = _w3(W)
    _run(dict(loader=0, epilogue=0, groups=1, relu=0, P=P, Cin=Cin, Kpad=Cin, Cout=Cout, W=W,
              bias=b, A=A, lda=Cin, out=out, ldc=Cout, precision=3, Kpad16=k16, W_bf16x3=w3,
              **_h2(W, A)), dev)
    ref = A.double() @ W.double().t()
    scale = A.double().abs() @ W.double().abs().t()
    err = ((out.double() - ref).abs() / scale)
    assert err[:P // 2].max().item() < 5e-7
    assert err[P // 2:].max().item() < 5e-7


def _check_model(dev, g, net, pts, full, precision="f16x2"):
    from s4g_release_amd.fused import FusedPointNet2
    fused = FusedPointNet2(net.to(dev).eval(), precision=precision)
    pred, inter = fused({"scene_points": torch.from_numpy(pts).to(dev)}, return_intermediates=True)
    for li in range(3):
        fps = inter["fps%d" % li].cpu().numpy().astype(np.int64)
        ball = inter["ball%d" % li].cpu().numpy().astype(np.int64)
        cnt = inter["cnt%d" % li].cpu().numpy().astype(np.int64)
        nn = inter["nn%d" % li].cpu().numpy().astype(np.int64)
        if full:
            assert GU.sha(fps) == str(g["fps%d_sha256" % li])
            assert GU.sha(ball) == str(g["ball%d_sha256" % li])
            assert GU.sha(cnt) == str(g["cnt%d_sha256" % li])
            assert GU.sha(nn) == str(g["nn%d_sha256" % li])
        else:
            assert np.array_equal(fps, g["fps%d" % li])
            assert np.array_equal(ball, g["ball%d" % li])
            assert np.array_equal(cnt, g["cnt%d" % li])
            assert np.array_equal(nn, g["nn%d" % li])
    return pred


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3", "fp32"])
def test_fused_model_small_golden(dev, precision):
    from s4g_release_amd.model import PointNet2
    g = GU.load("pn2_small.npz")
    net = PointNet2(**GU.small_config(g))
    net.load_state_dict(GU.small_state_dict(g), strict=True)
    pred = _check_model(dev, g, net, g["points"], full=False, precision=precision)
    assert pred.packed.is_contiguous()      # the four outputs are channel slices of one (B, 21, N) tensor (ABI 9)
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        assert all(pred[k][b].is_contiguous() for b in range(pred[k].shape[0]))     # what the reference's consumers index
        err = np.max(np.abs(pred[k].cpu().numpy() - g["out/" + k]))
        assert err < TOL, (k, err)


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3", "fp32"])
def test_fused_model_full_golden(dev, precision):
    from s4g_release_amd import synth
    g = GU.load("pn2_full.npz")
    net = GU.build_full_model(int(g["seed"]))
    pts = synth.make_batch([int(g["scene_id"])], 25600)
    pred = _check_model(dev, g, net, pts, full=True, precision=precision)
    pos = torch.from_numpy(g["positions"]).to(dev)
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        got = pred[k][:, :, pos].cpu().numpy()
        err = np.max(np.abs(got - g["out/" + k]))
        assert err < TOL, (k, err)


@pytest.mark.parametrize("fixture", ["pn2_real.npz", "pn2_real_replace.npz"])
@pytest.mark.parametrize("precision", ["f16x2", "bf16x3"])
def test_fused_model_real_scene_golden(dev, precision, fixture):
    """The reference's own sample scene (inference/2638_view_0.p, seeded 25 600-point
    subsample stored in the fixture; `_replace`: drawn WITH replacement as the harness draws it, a fifth
    of the points exact copies) through the reference's Python network:
    every index tensor bit-exact, outputs within 1e-4."""
    g = GU.load(fixture)
    net = GU.build_full_model(int(g["seed"]))
    assert GU.state_dict_sha256(net.state_dict()) == str(g["state_dict_sha256"])
    pred = _check_model(dev, g, net, g["points"], full=True, precision=precision)
    pos = torch.from_numpy(g["positions"]).to(dev)
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        got = pred[k][:, :, pos].cpu().numpy()
        err = np.max(np.abs(got - g["out/" + k]))
        assert err < TOL, (k, err)
        total = float(pred[k].double().sum().item())
        assert abs(total - float(g["outsum/" + k])) < 1e-4 * 25600, k


def test_bf16_single_product_mode_is_close_but_reduced(dev):
    """configs[4]: plain bf16 contraction.  Indices stay bit-exact (geometry is
    untouched); outputs agree with the fp32 path to bf16-class tolerance only."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    g = GU.load("pn2_small.npz")
    from s4g_release_amd.model import PointNet2
    net = PointNet2(**GU.small_config(g))
    net.load_state_dict(GU.small_state_dict(g), strict=True)
    net = net.to(dev).eval()
    pts = torch.from_numpy(g["points"]).to(dev)
    lo, inter = FusedPointNet2(net, precision="bf16")({"scene_points": pts}, return_intermediates=True)
    for li in range(3):
        assert np.array_equal(inter["ball%d" % li].cpu().numpy().astype(np.int64), g["ball%d" % li])
    worst = 0.0
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        err = float(np.max(np.abs(lo[k].cpu().numpy() - g["out/" + k])))
        scale = float(np.max(np.abs(g["out/" + k]))) + 1e-6
        worst = max(worst, err / scale)
    assert 1e-5 < worst < 0.1, worst     # clearly not fp32, clearly not garbage


def test_fused_equals_modules_path_batch(dev):
    """Same model, both product paths, a batch of 3 dup-heavy scenes."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import PointNet2, randomize_bn_
    cfg = dict(score_classes=3, num_centroids=(300, 70, 20), radius=(0.05, 0.12, 0.4),
               num_neighbours=(32, 16, 64), sa_channels=((32, 32, 64), (64, 64, 96), (96, 96, 128)),
               fp_channels=((128, 128), (64, 64), (32, 32, 32)), num_fp_neighbours=(3, 3, 3),
               seg_channels=(64, 32, 32, 16), num_removal_directions=5, dropout_prob=0.5)
    torch.manual_seed(99)
    pts = torch.from_numpy(synth.make_batch([1, 2, 3], 1500, variant="dup-heavy")).to(dev)
    net = GU.calibrated(PointNet2(**cfg).to(dev), 100, pts)
    with torch.no_grad():
        a = net({"scene_points": pts})
    b = FusedPointNet2(net)({"scene_points": pts})
    print(GU.check_against_float64({"modules": a, "f16x2": b}, net, pts, cfg))


@pytest.mark.parametrize("streams", [("1", "1"), ("3", "2")])
def test_pipelined_submissions_match_sequential(dev, monkeypatch, streams):
    """`submit` keeps several batches in flight on separate geometry / dense streams:
    every handle must return exactly what a lone forward of its batch returns."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import PointNet2, randomize_bn_
    monkeypatch.setenv("S4G_GEO_STREAMS", streams[0])
    monkeypatch.setenv("S4G_DENSE_STREAMS", streams[1])
    cfg = dict(score_classes=3, num_centroids=(600, 150, 40), radius=(0.04, 0.1, 0.3),
               num_neighbours=(64, 32, 16), sa_channels=((32, 32, 64), (64, 64, 128), (128, 128, 256)),
               fp_channels=((256, 256), (128, 128), (64, 64, 64)), num_fp_neighbours=(3, 3, 3),
               seg_channels=(128, 64, 64, 32), num_removal_directions=5, dropout_prob=0.5)
    torch.manual_seed(5)
    batches = [torch.from_numpy(synth.make_batch([10 * i, 10 * i + 1], 3000)).to(dev) for i in range(5)]
    net = GU.calibrated(PointNet2(**cfg).to(dev), 6, batches[0])
    fast = FusedPointNet2(net)
    with torch.no_grad():
        ref = [{k: v.clone() for k, v in fast({"scene_points": b}).items()} for b in batches]
        torch.cuda.synchronize()
        handles = [fast.submit({"scene_points": b}) for b in batches]      # all in flight
        outs = [h.result() for h in handles]
        torch.cuda.synchronize()
    for r, o in zip(ref, outs):
        for k in r:
            assert torch.equal(r[k], o[k]), k


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_fused_equals_modules_random_configs(dev, seed):
    """Random small architectures / cloud sizes (odd point counts, 16/32/64 neighbours,
    channel counts that miss every tile size): fast path == reference-shaped modules path."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import PointNet2, randomize_bn_
    rng = np.random.default_rng(100 + seed)
    n_pts = int(rng.integers(900, 2600))
    m1 = int(rng.integers(200, 400))
    m2 = int(rng.integers(48, m1 // 2))
    m3 = int(rng.integers(8, m2 // 2))
    ch = lambda lo, hi: int(rng.integers(lo, hi)) * 4       # noqa: E731  (fast path needs C % 4 == 0)
    sa = tuple((ch(4, 24), ch(4, 24), ch(8, 40)) for _ in range(3))
    fp = ((ch(8, 40), ch(8, 40)), (ch(8, 32), ch(8, 32)), (ch(4, 24), ch(4, 24), ch(4, 24)))
    cfg = dict(score_classes=3, num_centroids=(m1, m2, m3), radius=(0.05, 0.12, 0.4),
               num_neighbours=tuple(int(rng.choice([16, 32, 64])) for _ in range(3)), sa_channels=sa,
               fp_channels=fp, num_fp_neighbours=(3, 3, 3),
               seg_channels=(ch(8, 32), ch(4, 24), ch(4, 24), ch(2, 12)), num_removal_directions=5,
               dropout_prob=0.5)
    torch.manual_seed(seed)
    variant = ["tabletop-v1", "dup-heavy", "uniform-box"][seed % 3]
    pts = torch.from_numpy(synth.make_batch([seed, seed + 1, seed + 2][: 1 + seed % 3], n_pts, variant=variant)).to(dev)
    net = GU.calibrated(PointNet2(**cfg).to(dev), seed + 50, pts)
    with torch.no_grad():
        a = net({"scene_points": pts})
    preds = {"modules": a}
    for precision in ("f16x2", "bf16x3"):
        preds[precision] = FusedPointNet2(net, precision=precision)({"scene_points": pts})
    print(seed, GU.check_against_float64(preds, net, pts, cfg))


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_fused_equals_modules_chain_widths(dev, seed):
    """Random architectures whose widths are 128 / 256 (so the two- and three-layer chain
    launches, the deep first head layer and both linear-first restructurings all engage) on
    odd cloud sizes (ragged last tiles): fast path == reference-shaped modules path."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import PointNet2, randomize_bn_
    rng = np.random.default_rng(500 + seed)
    n_pts = int(rng.integers(1500, 2600)) | 1
    m1 = int(rng.integers(300, 500)) | 1
    m2 = int(rng.integers(64, 140)) | 1
    m3 = int(rng.integers(12, 30))
    w = lambda: int(rng.choice([128, 256]))      # noqa: E731
    c0, c1 = w(), w()
    sa = ((c0, c0, c0 * int(rng.choice([1, 2]))), (c1, c1, c1 * int(rng.choice([1, 2]))), (64, 96, 160))
    f2 = w()
    fp = ((192, 160), (f2 * 2, f2 * 2), (f2, f2, f2))
    h = 256
    cfg = dict(score_classes=3, num_centroids=(m1, m2, m3), radius=(0.05, 0.12, 0.4),
               num_neighbours=(64, 64, int(rng.choice([16, 32]))), sa_channels=sa, fp_channels=fp,
               num_fp_neighbours=(3, 3, 3), seg_channels=(2 * h, h, h, int(rng.choice([64, 128]))),
               num_removal_directions=5, dropout_prob=0.5)
    if f2 != h:
        cfg["fp_channels"] = ((192, 160), (f2 * 2, f2 * 2), (h, h, h))
    torch.manual_seed(seed)
    variant = ["tabletop-v1", "dup-heavy", "uniform-box"][seed % 3]
    pts = torch.from_numpy(synth.make_batch([seed, seed + 3][: 1 + seed % 2], n_pts, variant=variant)).to(dev)
    net = GU.calibrated(PointNet2(**cfg).to(dev), seed + 70, pts)
    with torch.no_grad():
        a = net({"scene_points": pts})
    fused = FusedPointNet2(net)
    b = fused({"scene_points": pts})
    print(seed, GU.check_against_float64({"modules": a, "f16x2": b}, net, pts, cfg))


def test_fused_layer_pairs_match_layer_by_layer(dev, monkeypatch):
    """The default (layer chains fused into single launches, first SA / FP layers applied
    before the grouping / interpolation) against the layer-by-layer, reference-order form
    on the bench architecture: same network outputs to fp32 round-off."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
    net = GU.shipped_net(dev)
    pts = torch.from_numpy(synth.make_batch([0, 1], 25600)).to(dev)
    a = FusedPointNet2(net)({"scene_points": pts})
    monkeypatch.setenv("S4G_GEMM_FUSE2", "0")
    monkeypatch.setenv("S4G_SA_LINEAR_FIRST", "0")
    monkeypatch.setenv("S4G_FP_LINEAR_FIRST", "0")
    b = FusedPointNet2(net)({"scene_points": pts})
    for k in a:
        scale = max(1.0, b[k].abs().max().item())
        # (calibrated weights: measured 3.1e-5 of scale; 2e-6 on the old per-channel-constant network)
        assert (a[k] - b[k]).abs().max().item() < TOL * scale, k


@pytest.mark.parametrize("precision", ["f16x2", "bf16"])
def test_shared_input_layers_as_one_launch_match_separate_launches(dev, monkeypatch, precision):
    """`sa{l}.0f` and `fp{f}.0d` read the same level's features: one launch with two output tensors
    (S4G_MERGE_SHARED=0 keeps two launches).  Per-channel weight scales: the same values either way, up to the
    accumulation order of the kernel a shape dispatches to."""
    from s4g_release_amd import functions as F
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
    net = GU.shipped_net(dev)
    batch = {"scene_points": torch.from_numpy(synth.make_batch([3, 4], 25600)).to(dev)}

    def run():
        F.OpTimer.reset(enabled=True)
        out = FusedPointNet2(net, precision=precision)(batch)
        torch.cuda.synchronize()
        F.OpTimer.enabled = False
        return out, sorted(F.OpTimer.summary())

    merged, names = run()
    assert any("sa1.0f|fp1.0d" in n for n in names) and any("sa2.0f|fp0.0d" in n for n in names)
    assert not any("[fp1.0d " in n or "[fp0.0d " in n for n in names)
    monkeypatch.setenv("S4G_MERGE_SHARED", "0")
    apart, names = run()
    assert any("[fp1.0d " in n for n in names) and any("[sa1.0f " in n for n in names)
    for k in merged:
        assert (merged[k] - apart[k]).abs().max().item() < (2e-2 if precision == "bf16" else 2e-6), k


def test_forward_recorded_as_a_hip_graph_replays_bit_identically(dev):
    """`FusedPointNet2.graph`: one pass (geometry streams, contraction stream, ~45 launches) captured as a HIP graph;
    replays on other clouds of the same shape give the eager path's tensors bit for bit, a wrong shape is refused."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
    net = GU.shipped_net(dev)
    run = FusedPointNet2(net)
    a = torch.from_numpy(synth.make_batch([0, 1], 25600)).to(dev)
    b = torch.from_numpy(synth.make_batch([7, 8], 25600, variant="dup-heavy")).to(dev)
    g = run.graph({"scene_points": a})
    for x in (a, b, a):
        ref = {k: v.clone() for k, v in run({"scene_points": x}).items()}
        got = g({"scene_points": x})
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(ref[k], got[k]), k
    with pytest.raises(RuntimeError):
        g({"scene_points": a[:1]})


@pytest.mark.parametrize("precision", ["f16x2", "bf16", "fp32"])
def test_heads_written_into_one_packed_tensor_bit_identical(dev, monkeypatch, precision):
    """ABI 9 (`s4g_heads_desc_t.out_batch_stride`; the channel-first epilogue of the layer-by-layer heads takes the
    packed tensor as one 21-channel head): the four outputs are channel slices of ONE (B, 21, N) tensor that
    `dist.pack_outputs` hands to the all-gather without a copy -- every value bit-identical to four tensors of
    their own (S4G_PACKED_OUT=0), at a batch of 3 (the batch stride matters) and a ragged last panel."""
    from s4g_release_amd import dist as sdist, synth
    from s4g_release_amd.fused import FusedPointNet2, PackedPred
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
    net = GU.shipped_net(dev)
    x = {"scene_points": torch.from_numpy(synth.make_batch([2, 3, 4], 25600 - 40)).to(dev)}
    packed = FusedPointNet2(net, precision=precision)(x)
    assert isinstance(packed, PackedPred) and packed.packed.shape == (3, 21, 25600 - 40)
    out, chans = sdist.pack_outputs(packed)
    assert out.data_ptr() == packed.packed.data_ptr() and chans == [3, 9, 4, 5]
    c0 = 0
    for k, c in zip(sdist.HEADS, chans):
        assert packed[k].data_ptr() == packed.packed[:, c0:c0 + c].data_ptr() and packed[k].shape == (3, c, 25560)
        c0 += c
    monkeypatch.setenv("S4G_PACKED_OUT", "0")
    plain = FusedPointNet2(net, precision=precision)(x)
    assert plain.packed is None and all(v.is_contiguous() for v in plain.values())
    torch.cuda.synchronize()
    for k in sdist.HEADS:
        assert torch.equal(plain[k], packed[k]), k
    assert torch.equal(sdist.pack_outputs(plain)[0], packed.packed)


def test_heads_desc_refuses_a_batch_stride_below_a_heads_own_block(dev):
    import ctypes as C
    from s4g_release_amd import _cabi
    d = _cabi.HeadsDesc()
    d.precision, d.P, d.N, d.ldx = 3, 128, 64, 256
    d.C, d.H0, d.H1, d.H2, d.H3 = 256, 512, 256, 256, 128
    buf = torch.zeros(1 << 16, device=dev)
    d.X = buf.data_ptr()
    for l in range(5):
        d.W_frag[l] = d.bias[l] = d.w_inv_scale[l] = buf.data_ptr()
    for h, c in enumerate((3, 9, 4, 5)):
        d.out[h], d.channels[h] = buf.data_ptr(), c
    d.a_amax_floor = 1.0
    d.out_batch_stride = 8 * 64          # below frame_R's 9 channels x 64 points
    assert _cabi.lib().s4g_heads_chain_f32(C.byref(d), None) == -1      # S4G_EINVAL

"""configs[4]: the single-plane bf16 form of the fused layer chains (S4G_GEMM_BF16 +
W*_f16x2_frag holding ONE bf16 plane in fragment order).  The kernel rounds its inputs, the
weights and every hidden activation to bf16 and accumulates in fp32, so a reference that rounds
at the same points and accumulates in fp64 pins it to fp32 round-off; against the un-rounded
fp64 layer the error is bf16-class (stated per test).  Layer being accelerated: the 1x1 conv ->
BN -> ReLU stacks of `nn_utils/conv.py:24-34,64-74` and the max over neighbours of
`pointnet2_utils/modules.py:242-243`."""
import ctypes

import numpy as np
import pytest
import torch

from tests import golden_util as GU

pytestmark = pytest.mark.gpu
HEADS = ("score", "frame_R", "frame_t", "movable_logits")


def _run(desc_kwargs):
    from s4g_release_amd import _cabi
    d = _cabi.GemmDesc()
    for k, v in desc_kwargs.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(d, k, v)
    rc = _cabi.lib().s4g_mlp_gemm_f32(ctypes.byref(d), torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "gemm")
    torch.cuda.synchronize()


def _bf(x):
    return x.to(torch.bfloat16).double()


def _frag(w):
    """(G, Cout, K16) or (Cout, K16) fp32 -> bf16 fragment-ordered plane + the bf16x3 planes."""
    from s4g_release_amd.fused import fragment_order, split_bf16x3
    w3 = split_bf16x3(w)
    return fragment_order(w3[:1])[:, :, :, 0].contiguous(), w3


@pytest.mark.parametrize("C,K1,Cout3,groups,P", [(128, 128, 256, 1, 300), (256, 256, 128, 4, 200),
                                                  (256, 512, 128, 4, 130), (512, 512, 1024, 1, 70),
                                                  (256, 256, 2048, 1, 64)])
@pytest.mark.parametrize("tri", [False, True])
def test_bf16_chain_plain_store(dev, C, K1, Cout3, groups, P, tri):
    """Two- and three-layer chains, plain loader, store epilogue, grouped column slices,
    one- and two-panel first layers, four- and eight-wave forms."""
    if K1 != C and not tri and C != 256:
        pytest.skip("deep first layer exists for C = 256 only")
    g = torch.Generator(device="cpu").manual_seed(C + K1 + Cout3 + groups)
    A = torch.randn(P, groups * K1, generator=g).to(dev)
    W1 = (torch.randn(groups, C, K1, generator=g) / K1 ** 0.5).to(dev)
    W2 = (torch.randn(groups, C if tri else Cout3, C, generator=g) / C ** 0.5).to(dev)
    W3 = (torch.randn(groups, Cout3, C, generator=g) / C ** 0.5).to(dev)
    b1 = torch.randn(groups, C, generator=g).to(dev)
    b2 = torch.randn(groups, W2.shape[1], generator=g).to(dev)
    b3 = torch.randn(groups, Cout3, generator=g).to(dev)
    out = torch.full((P, groups * Cout3), float("nan"), device=dev)
    f1, w3p = _frag(W1)
    f2, _ = _frag(W2)
    f3, _ = _frag(W3)
    kw = dict(loader=0, epilogue=0, groups=groups, relu=1, P=P, Cin=K1, Kpad=K1, Cout=C, W=W1, bias=b1,
              w_gstride=C * K1, b_gstride=C, A=A, lda=groups * K1, a_gcol=K1, out=out, ldc=groups * Cout3,
              c_gcol=Cout3, precision=2, Kpad16=K1, W_bf16x3=w3p, W_f16x2_frag=f1, W2_f16x2_frag=f2,
              bias2=b2, Cout2=W2.shape[1], relu2=1)
    if tri:
        kw.update(W3_f16x2_frag=f3, bias3=b3, Cout3=Cout3, relu3=1)
    _run(kw)
    ref, exact = [], []
    for gi in range(groups):
        a = A[:, gi * K1:(gi + 1) * K1]
        h = (_bf(a) @ _bf(W1[gi]).t() + b1[gi].double()).clamp_min(0)
        h = (_bf(h.float()) @ _bf(W2[gi]).t() + b2[gi].double()).clamp_min(0)
        e = (a.double() @ W1[gi].double().t() + b1[gi].double()).clamp_min(0)
        e = (e @ W2[gi].double().t() + b2[gi].double()).clamp_min(0)
        if tri:
            h = (_bf(h.float()) @ _bf(W3[gi]).t() + b3[gi].double()).clamp_min(0)
            e = (e @ W3[gi].double().t() + b3[gi].double()).clamp_min(0)
        ref.append(h)
        exact.append(e)
    ref, exact = torch.cat(ref, dim=1), torch.cat(exact, dim=1)
    assert torch.isfinite(out).all()
    scale = max(1.0, ref.abs().max().item())
    # hidden activations that sit on a bf16 rounding boundary may round the other way after fp32
    # (instead of fp64) accumulation: allow a few such flips
    assert (out.double() - ref).abs().max().item() < 2e-3 * scale
    assert (out.double() - ref).abs().mean().item() < 2e-5 * scale
    assert (out.double() - exact).abs().max().item() < 5e-2 * scale     # bf16-class against fp64


def test_bf16_chain_mlp1_loader_max(dev):
    """The SA0 launch in bf16: xyz gather + first layer in the loader, two contractions, max."""
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
    f1, w3p = _frag(W)
    f2, _ = _frag(W2)
    _run(dict(loader=3, epilogue=1, groups=1, relu=1, P=P, Cin=C, Kpad=C, Cout=C, W=W, bias=b, gidx=gidx,
              xyz=xyz, ctr=ctr, N=N, M=M, K=K, mlp1_w=w1, out=out, ldc=Cout2, precision=2, Kpad16=C,
              W_bf16x3=w3p, W_f16x2_frag=f1, W2_f16x2_frag=f2, bias2=b2, Cout2=Cout2, relu2=1))
    rel = torch.stack([xyz[bi][:, gidx[bi].long()] - ctr[bi][:, :, None] for bi in range(B)])
    rel = rel.permute(0, 2, 3, 1).reshape(P, 3)
    A = (rel @ w1[:, :3].t() + w1[:, 3]).clamp_min(0)                # fp32, like the loader (fma order aside)
    h = (_bf(A) @ _bf(W).t() + b.double()).clamp_min(0)
    ref = (_bf(h.float()) @ _bf(W2).t() + b2.double()).clamp_min(0).view(B * M, K, Cout2).max(dim=1)[0]
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 2e-3 * scale
    assert (out.double() - ref).abs().mean().item() < 3e-5 * scale


def test_bf16_chain_is_rejected_without_a_second_layer_or_with_bad_widths(dev):
    from s4g_release_amd import _cabi
    W1 = torch.randn(64, 64, device=dev)
    f, w3p = _frag(W1)
    A = torch.randn(64, 64, device=dev)
    out = torch.empty(64, 64, device=dev)
    with pytest.raises(RuntimeError):    # C = 64 is not a chain width
        _run(dict(loader=0, epilogue=0, groups=1, relu=1, P=64, Cin=64, Kpad=64, Cout=64, W=W1,
                  bias=torch.zeros(64, device=dev), A=A, lda=64, out=out, ldc=64, precision=2, Kpad16=64,
                  W_bf16x3=w3p, W_f16x2_frag=f, W2_f16x2_frag=f, bias2=torch.zeros(64, device=dev),
                  Cout2=64, relu2=1))


def _models(dev):
    from s4g_release_amd.fused import FusedPointNet2
    net = GU.shipped_net(dev)
    return FusedPointNet2(net, precision="bf16"), FusedPointNet2(net, precision="f16x2")


def test_bf16_forward_51200_points_indices_exact_outputs_bf16_close(dev):
    """configs[4] geometry + arithmetic at B = 2: every index tensor of the bf16 path equals the
    fp32-class path's (the geometry kernels are shared) AND the CPU oracle's at the sizes it
    finishes in seconds (FPS 51 200 -> 5 120 with all M, ball query, 3-NN).

    Arithmetic, LAUNCH BY LAUNCH against an fp64 reference that rounds activations and weights to
    bf16 at the same points (tests/bf16_reference.py), each launch fed with the tensors the HIP path
    itself left at the previous layer boundary, relative to max |reference| of the tensor:
    single-layer launches max < 5e-6 (fp32 accumulation only); two-layer chain launches (hidden
    activation re-rounded in LDS) mean < 2e-6, max < 5e-3 (an fp32-vs-fp64 accumulation difference
    now and then moves ONE bf16 rounding of a hidden activation); the FP tail + four heads launch
    (seven layers deep) mean < 5e-5, max < 2e-2.  Measured (tools/bf16_probe.py): 6e-7; 3e-7 / 1e-3;
    5e-6 / 5e-3.  End to end the bf16 roundings of every layer's activations compound.  On the CALIBRATED
    network (tests/golden_util.shipped_net: every layer re-normalised, which amplifies a perturbation ~750 x over the
    17 layers -- fp32's 6e-8 arrives at 5e-5 of scale, bf16's 2e-3 arrives at order one) the whole bf16 forward
    sits mean 0.04 / max 0.30 .. 0.44 of max |reference| from the fp32-class forward (measured); rounds 1-5's
    per-channel-constant network hid that (6e-3 / 1.5e-3).  configs[4] is a roofline configuration, not an
    accuracy claim: bounded here at mean <= 0.05, max <= 0.5, the launch-by-launch bounds above are the pin."""
    from oracle import oracle as O
    from s4g_release_amd import synth
    lo, hi = _models(dev)
    assert lo._fusable(lo.sa[0]["layers"][-2], lo.sa[0]["layers"][-1], 3, 1)    # the chains ARE used
    pts = torch.from_numpy(synth.make_batch([0, 1], 51200)).to(dev)
    from tests.bf16_reference import GemmCapture, bf16_stagewise_errors
    with torch.no_grad():
        with GemmCapture(lo) as cap:
            pl, il = lo({"scene_points": pts}, return_intermediates=True)
        pl = {k: v.clone() for k, v in pl.items()}
        il = {k: v.clone() for k, v in il.items()}
        ph, ih = hi({"scene_points": pts}, return_intermediates=True)
    for k in il:
        if k.startswith(("fps", "ball", "cnt", "nn")) and not k.startswith("nnw"):
            assert torch.equal(il[k], ih[k]), k
    for b in (0, 1):
        stages = bf16_stagewise_errors(lo, pts, il, cap.out, pl, b)
        assert len(stages) == 15, [n for n, _, _ in stages]     # 11 launches; the last one has 4 outputs
        for name, emax, emean in stages:
            if "heads" in name:
                assert emax < 2e-2 and emean < 5e-5, (b, name, emax, emean)
            elif "+" in name:
                assert emax < 5e-3 and emean < 2e-6, (b, name, emax, emean)
            else:
                assert emax < 5e-6, (b, name, emax, emean)
    x = pts.cpu().numpy()
    fps0 = O.fps(x, 5120)
    assert np.array_equal(il["fps0"].cpu().numpy().astype(np.int64), fps0)
    ctr = O.gather_points(x, fps0)
    ball, cnt = O.ball_query(x, ctr, 0.02, 64)
    assert np.array_equal(il["ball0"].cpu().numpy().astype(np.int64), ball)
    assert np.array_equal(il["cnt0"].cpu().numpy().astype(np.int64), cnt)
    nn, _ = O.three_nn(x, ctr)
    assert np.array_equal(il["nn2"].cpu().numpy().astype(np.int64), nn)
    for k in HEADS:
        ref = ph[k].double()
        d = (pl[k].double() - ref).abs()
        s = ref.abs().max().item()
        assert torch.isfinite(pl[k]).all()
        print("bf16 vs f16x2 %-14s max %.3g mean %.3g of scale %.3g" % (k, d.max().item() / s, d.mean().item() / s, s))
        assert d.max().item() <= 0.5 * s, (k, d.max().item(), s)
        assert d.mean().item() <= 0.05 * s, (k, d.mean().item(), s)


def test_bf16_forward_b32_full_size_is_batch_invariant_and_finite(dev):
    """configs[4] at its full size (32 scenes x 51 200 points): finite outputs, scene 0 and scene
    31 equal their single-scene results (no cross-scene term; bf16 has no scales at all), index
    tensors of scene 31 equal the single-scene run's."""
    from s4g_release_amd import synth
    lo, _ = _models(dev)
    pts = torch.from_numpy(synth.make_batch(list(range(32)), 51200)).to(dev)
    with torch.no_grad():
        full, inter = lo({"scene_points": pts}, return_intermediates=True)
        full = {k: v.clone() for k, v in full.items()}
        inter = {k: v.clone() for k, v in inter.items()}
        for s in (0, 31):
            one, i1 = lo({"scene_points": pts[s:s + 1].contiguous()}, return_intermediates=True)
            for k in HEADS:
                assert torch.equal(one[k][0], full[k][s]), (s, k)
            for k in ("fps0", "ball0", "nn2"):
                assert torch.equal(i1[k][0], inter[k][s]), (s, k)
    for k in HEADS:
        assert torch.isfinite(full[k]).all()
    assert full["score"].shape == (32, 3, 51200)

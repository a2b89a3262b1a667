"""s4g_heads_chain_f32: the four per-point heads (PointNet2_tcls.py:83-95,126-140 -- four
SharedMLP stacks 256 -> 512 -> 256 -> 256 -> 128 on a shared input, Conv1d logits, sigmoid on
the movable head) as one launch, against an fp64 restatement of the same layers."""
import ctypes

import pytest
import torch

from tests import golden_util as GU

pytestmark = pytest.mark.gpu
CH = (3, 9, 4, 5)


def _layers(dev, seed):
    from s4g_release_amd.fused import _Layer
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    W0 = (r(2048, 256) / 16).to(dev)
    W1 = (r(4, 256, 512) / 512 ** 0.5).to(dev)
    W2 = (r(4, 256, 256) / 16).to(dev)
    W3 = (r(4, 128, 256) / 16).to(dev)
    WL = torch.zeros(4, 32, 128)
    bL = torch.zeros(4, 32)
    for h, c in enumerate(CH):
        WL[h, :c] = r(c, 128) / 128 ** 0.5
        bL[h, :c] = r(c)
    b = [r(2048).to(dev), r(4, 256).to(dev), r(4, 256).to(dev), r(4, 128).to(dev), bL.to(dev)]
    Ws = [W0, W1, W2, W3, WL.to(dev)]
    layers = [_Layer(W0, b[0], 256)] + [_Layer(Ws[i], b[i], Ws[i].shape[-1], groups=4) for i in range(1, 5)]
    return Ws, b, layers


def _run(dev, layers, X, B, N, precision, amax=None, floor=0.0):
    from s4g_release_amd import _cabi
    d = _cabi.HeadsDesc()
    d.precision = precision
    d.P, d.N, d.ldx = B * N, N, X.shape[1]
    d.C, d.H0, d.H1, d.H2, d.H3 = 256, 512, 256, 256, 128
    d.X = X.data_ptr()
    for l, layer in enumerate(layers):
        d.W_frag[l] = (layer.Wfrag_bf16 if precision == 2 else layer.Wfrag).data_ptr()
        d.bias[l] = layer.bias.data_ptr()
        d.w_inv_scale[l] = layer.w_inv_scale.data_ptr()
    outs = [torch.full((B, c, N), float("nan"), device=dev) for c in CH]
    for h, o in enumerate(outs):
        d.out[h] = o.data_ptr()
        d.channels[h] = CH[h]
    d.sigmoid_head = 3
    d.a_amax = None if amax is None else amax.data_ptr()
    d.a_amax_floor = floor
    d.rows_per_scene = N
    rc = _cabi.lib().s4g_heads_chain_f32(ctypes.byref(d), torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "heads")
    torch.cuda.synchronize()
    return outs


def _reference(Ws, b, X, B, N, rnd=lambda t: t.double()):
    outs = []
    x = rnd(X)
    for h, c in enumerate(CH):
        y = (x @ rnd(Ws[0][h * 512:(h + 1) * 512]).t() + b[0][h * 512:(h + 1) * 512].double()).clamp_min(0)
        for l in (1, 2, 3):
            y = (rnd(y.float()) @ rnd(Ws[l][h]).t() + b[l][h].double()).clamp_min(0)
        o = rnd(y.float()) @ rnd(Ws[4][h, :c]).t() + b[4][h, :c].double()
        if h == 3:
            o = torch.sigmoid(o)
        outs.append(o.view(B, N, c).permute(0, 2, 1))
    return outs


@pytest.mark.parametrize("B,N", [(2, 100), (1, 64), (3, 171)])
def test_heads_chain_f16x2_is_fp32_class(dev, B, N):
    Ws, b, layers = _layers(dev, 7 + N)
    g = torch.Generator(device="cpu").manual_seed(N)
    X = (torch.randn(B * N, 256, generator=g) * torch.tensor([1.0, 40.0, 0.02])[:B, None]
         .repeat_interleave(N, dim=0)).to(dev)      # scenes of very different magnitude
    amax = torch.zeros((B, 64), device=dev)
    amax[:, 5] = X.view(B, -1).abs().amax(dim=1)
    outs = _run(dev, layers, X, B, N, 3, amax=amax)
    ref = _reference(Ws, b, X, B, N)
    for h in range(4):
        assert torch.isfinite(outs[h]).all()
        err = (outs[h].double() - ref[h]).abs()
        scale = ref[h].abs().amax(dim=(1, 2), keepdim=True).clamp_min(1.0)     # per scene
        assert (err / scale).max().item() < 2e-5, (h, (err / scale).max().item())


def test_heads_chain_bf16_matches_a_reference_rounded_at_the_same_points(dev):
    B, N = 2, 150
    Ws, b, layers = _layers(dev, 3)
    X = torch.randn(B * N, 256, generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)
    outs = _run(dev, layers, X, B, N, 2)
    ref = _reference(Ws, b, X, B, N, rnd=lambda t: t.to(torch.bfloat16).double())
    exact = _reference(Ws, b, X, B, N)
    for h in range(4):
        scale = max(1.0, ref[h].abs().max().item())
        assert torch.isfinite(outs[h]).all()
        assert (outs[h].double() - ref[h]).abs().max().item() < 3e-3 * scale
        assert (outs[h].double() - ref[h]).abs().mean().item() < 5e-5 * scale
        assert (outs[h].double() - exact[h]).abs().max().item() < 8e-2 * scale


def test_heads_chain_rejects_other_widths_and_bad_arguments(dev):
    from s4g_release_amd import _cabi
    Ws, b, layers = _layers(dev, 1)
    X = torch.randn(64, 256, device=dev)
    d = _cabi.HeadsDesc()
    d.precision, d.P, d.N, d.ldx = 3, 64, 64, 256
    d.C, d.H0, d.H1, d.H2, d.H3 = 256, 512, 256, 256, 64          # not the shipped widths
    d.X = X.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    assert _cabi.lib().s4g_heads_chain_f32(ctypes.byref(d), st) != 0
    d.H3 = 128
    assert _cabi.lib().s4g_heads_chain_f32(ctypes.byref(d), st) != 0     # no weights / outputs given


def _pre_setup(dev, B, N, N2, seed, with_dense):
    """Inputs of the feature-propagation tail in front of the heads (s4g_heads_desc_t.pre_*):
    sparse features (B N2, 256), three neighbour indices + weights per point, an optional dense
    addend, the first layer's bias, and the two 256 -> 256 layers."""
    from s4g_release_amd.fused import _Layer
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    S = (r(B * N2, 256) * torch.tensor([1.0, 25.0, 0.05])[:B, None].repeat_interleave(N2, dim=0)).to(dev)
    nidx = torch.randint(0, N2, (B * N, 3), generator=g, dtype=torch.int32).to(dev)
    w = torch.rand(B * N, 3, generator=g)
    nw = (w / w.sum(dim=1, keepdim=True)).to(dev)
    dense = (r(B * N, 256) * 0.5).to(dev) if with_dense else None
    lbias = r(256).to(dev)
    pl = [_Layer((r(256, 256) / 16).to(dev), r(256).to(dev), 256) for _ in range(2)]
    return S, nidx, nw, dense, lbias, pl


def _pre_reference(S, nidx, nw, dense, lbias, pl, B, N, N2, rnd=lambda t: t.double()):
    base = (torch.arange(B, device=S.device).repeat_interleave(N) * N2).view(-1, 1)
    rows = S.double()[(base + nidx.long())]                       # (P, 3, 256)
    x = (rows * nw.double().unsqueeze(-1)).sum(dim=1) + lbias.double()
    if dense is not None:
        x = x + dense.double()
    x = x.clamp_min(0)
    for layer in pl:
        w = layer.W[0] if layer.W.dim() == 3 else layer.W
        x = (rnd(x.float()) @ rnd(w).t() + layer.bias.double()).clamp_min(0)
    return x.float()


@pytest.mark.parametrize("precision", [3, 2])
@pytest.mark.parametrize("B,N,N2,with_dense", [(2, 100, 40, False), (3, 171, 64, True), (1, 64, 3, False)])
def test_heads_chain_with_fp_tail_in_front(dev, precision, B, N, N2, with_dense):
    """ABI 6: interpolate + add + ReLU in the loader, two 256 -> 256 layers inside LDS, then the
    heads -- against fp64 (f16x2: fp32-class) / a reference rounded to bf16 at the layer inputs."""
    from s4g_release_amd import _cabi
    Ws, b, layers = _layers(dev, 11 + N)
    S, nidx, nw, dense, lbias, pl = _pre_setup(dev, B, N, N2, 5 + N, with_dense)
    d = _cabi.HeadsDesc()
    d.precision = precision
    d.P, d.N = B * N, N
    d.C, d.H0, d.H1, d.H2, d.H3 = 256, 512, 256, 256, 128
    pick = (lambda l: l.Wfrag_bf16) if precision == 2 else (lambda l: l.Wfrag)
    for l, layer in enumerate(layers):
        d.W_frag[l], d.bias[l], d.w_inv_scale[l] = pick(layer).data_ptr(), layer.bias.data_ptr(), layer.w_inv_scale.data_ptr()
    for l, layer in enumerate(pl):
        d.pre_W_frag[l], d.pre_bias[l] = pick(layer).data_ptr(), layer.bias.data_ptr()
        d.pre_w_inv_scale[l] = layer.w_inv_scale.data_ptr()
    d.pre_nidx, d.pre_nw, d.pre_sparse, d.pre_N2 = nidx.data_ptr(), nw.data_ptr(), S.data_ptr(), N2
    d.pre_dense = None if dense is None else dense.data_ptr()
    d.pre_lbias = lbias.data_ptr()
    amax = torch.zeros((B, 64), device=dev)
    amax[:, 9] = S.view(B, -1).abs().amax(dim=1)
    d.a_amax = amax.data_ptr()
    if dense is not None:
        amax2 = torch.zeros((B, 64), device=dev)
        amax2[:, 1] = dense.view(B, -1).abs().amax(dim=1)
        d.pre_a_amax2 = amax2.data_ptr()
    d.a_amax_floor = float(lbias.abs().max())
    d.rows_per_scene = N
    outs = [torch.full((B, c, N), float("nan"), device=dev) for c in CH]
    for h, o in enumerate(outs):
        d.out[h], d.channels[h] = o.data_ptr(), CH[h]
    d.sigmoid_head = 3
    _cabi.check(_cabi.lib().s4g_heads_chain_f32(ctypes.byref(d), torch.cuda.current_stream().cuda_stream), "heads")
    torch.cuda.synchronize()
    if precision == 3:
        X = _pre_reference(S, nidx, nw, dense, lbias, pl, B, N, N2)
        ref = _reference(Ws, b, X, B, N)
        for h in range(4):
            assert torch.isfinite(outs[h]).all()
            err = (outs[h].double() - ref[h]).abs()
            scale = ref[h].abs().amax(dim=(1, 2), keepdim=True).clamp_min(1.0)
            assert (err / scale).max().item() < 3e-5, (h, (err / scale).max().item())
    else:
        rb = lambda t: t.to(torch.bfloat16).double()
        X = _pre_reference(S, nidx, nw, dense, lbias, pl, B, N, N2, rnd=rb)
        ref = _reference(Ws, b, X, B, N, rnd=rb)
        for h in range(4):
            scale = max(1.0, ref[h].abs().max().item())
            assert torch.isfinite(outs[h]).all()
            assert (outs[h].double() - ref[h]).abs().max().item() < 6e-3 * scale
            assert (outs[h].double() - ref[h]).abs().mean().item() < 1e-4 * scale


def test_fused_model_with_and_without_the_tail_in_the_heads_launch(dev, monkeypatch):
    """The whole network at a small size: S4G_HEADS_PRE=0 (separate fp2 chain launch) and the default
    (tail inside the heads launch) agree to fp32 round-off, and the launch list shrinks by one."""
    from s4g_release_amd import functions as F, synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
    net = GU.shipped_net(dev)
    pts = torch.from_numpy(synth.make_batch([0, 1], 25600)).to(dev)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("S4G_HEADS_PRE", flag)
        run = FusedPointNet2(net)
        F.OpTimer.reset(enabled=True)
        with torch.no_grad():
            out = run({"scene_points": pts})
        torch.cuda.synchronize()
        F.OpTimer.enabled = False
        res[flag] = ({k: v.clone() for k, v in out.items()}, sorted(F.OpTimer.summary()))
    a, b = res["0"], res["1"]
    assert any("fp2.1+fp2.2+heads" in n for n in b[1]) and not any("fp2.1+fp2.2+heads" in n for n in a[1])
    assert len([n for n in a[1] if n.startswith("gemm[")]) == len([n for n in b[1] if n.startswith("gemm[")]) + 1
    for k in a[0]:
        assert (a[0][k] - b[0][k]).abs().max().item() < 2e-5, k


def test_fused_model_fp1_chain_into_the_next_levels_first_layer(dev, monkeypatch):
    """S4G_FP_CHAIN_NEXT: FP level 1's (interpolate + add) -> 512 -> 512 layer and level 2's linear
    first layer as one chain launch (level 1's own output never written) against the separate
    launches: same outputs to fp32 round-off, two launches (interp_add + fp2.0s) fewer."""
    from s4g_release_amd import functions as F, synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
    net = GU.shipped_net(dev)
    pts = torch.from_numpy(synth.make_batch([2, 3], 25600)).to(dev)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("S4G_FP_CHAIN_NEXT", flag)
        run = FusedPointNet2(net)
        F.OpTimer.reset(enabled=True)
        with torch.no_grad():
            out = run({"scene_points": pts})
        torch.cuda.synchronize()
        F.OpTimer.enabled = False
        res[flag] = ({k: v.clone() for k, v in out.items()}, sorted(F.OpTimer.summary()))
    a, b = res["0"], res["1"]
    assert any("fp1.1+fp2.0s" in n for n in b[1]) and not any("fp1.1+fp2.0s" in n for n in a[1])
    assert any(n.startswith("gemm[fp2.0s") for n in a[1]) and not any(n.startswith("gemm[fp2.0s") for n in b[1])
    assert len([n for n in a[1] if n.startswith("interp_add")]) == len([n for n in b[1] if n.startswith("interp_add")]) + 1
    for k in a[0]:
        assert (a[0][k] - b[0][k]).abs().max().item() < 2e-5, k

"""fp64 restatement of the single-plane bf16 forward (`FusedPointNet2(net, precision="bf16")`) that
rounds AT THE SAME POINTS as the HIP path: every contraction's activations and weights to bf16
(round to nearest even), fp64 accumulate, bias / ReLU / neighbour max / interpolation un-rounded.
The order of operations is the fast path's (first SA / FP layers applied before the grouping /
interpolation: linear, so exact in real arithmetic, but the rounding points move with it), the
arithmetic being restated is `nn_utils/conv.py:24-34,64-74` (conv -> BN -> ReLU, BN folded),
`pointnet2_utils/modules.py:47-52,242-243` (grouping, max over neighbours), `:118-127`
(interpolation + skip concat) and `PointNet2_tcls.py:126-140` (heads).

Test infrastructure: takes the folded fp32 weights and the index tensors from the object under
test (the fold and the geometry have their own tests against the torch modules and the oracle),
so what it pins is the contraction dataflow of the bf16 configuration.

Two uses.  `bf16_forward_reference`: the whole forward, end to end.  fp32-vs-fp64 accumulation moves
a bf16 rounding now and then (one bf16 ulp of one of K inputs), and 20 layers deep those flips have
compounded to the bf16 noise level itself -- end to end the mirrored reference is no closer to the
HIP path than the exact forward is (measured: 6e-3 of max|ref| both ways).  `bf16_stagewise_errors`:
the same arithmetic LAUNCH BY LAUNCH, every stage fed with the tensors the HIP path itself produced
at the previous layer boundary (captured by wrapping `FusedPointNet2._gemm`), so that only the
flips of ONE launch separate the two: that is the tight pin (means of 1e-8 .. 1e-4 of max|ref|)."""
import torch


def _rb(t):
    """fp64 -> (fp32 ->) bf16 -> fp64, the rounding of an MFMA operand."""
    return t.float().to(torch.bfloat16).double()


def _lin(x, layer, relu, g=None):
    """x (P, cin) fp64 -> (P, cout): bf16(x) . bf16(W)^T + bias, optional ReLU; g = group index."""
    W = layer.W if g is None else layer.W[g]
    b = layer.bias if g is None else layer.bias[g]
    y = _rb(x) @ _rb(W[:, :layer.cin].double()).t() + b.double()
    return y.clamp_min(0) if relu else y


def _heads_reference(model, sparse):
    """The four heads on per-point features (N, 256) fp64 -> {name: (C, N)}."""
    hl = model.head_layers
    h0 = _lin(sparse, hl[0], relu=True)                     # (N, 4 * 512): the heads share the input
    width = hl[0].cout // 4
    names = ("score", "frame_R", "frame_t", "movable_logits")
    out = {}
    for h, name in enumerate(names):
        y = h0[:, h * width:(h + 1) * width]
        for layer in hl[1:]:
            y = _lin(y, layer, relu=True, g=h)
        c = model.head_channels[h]
        if model.heads_fused is not None:
            lg = model.heads_fused[-1]
            o = _rb(y) @ _rb(lg.W[h, :c, :lg.cin].double()).t() + lg.bias[h, :c].double()
        else:
            r0 = sum(model.head_channels[:h])
            lg = model.logit_layer
            o = _rb(y) @ _rb(lg.W[r0:r0 + c, h * y.shape[1]:(h + 1) * y.shape[1]].double()).t() + \
                lg.bias[r0:r0 + c].double()
        if h == 3:
            o = torch.sigmoid(o)
        out[name] = o.t().contiguous()
    return out


def bf16_forward_reference(model, pts, inter, b, trace=None):
    """Head outputs {name: (C, N) fp64} of scene `b`; model = the FusedPointNet2 under test (bf16),
    pts (B, 3, N) fp32 on its device, inter = its `return_intermediates` dict."""
    assert model.precision == "bf16" and model.fp_linear_first
    xyz = pts[b].double()                                   # (3, N)
    level_xyz, level_feat = [xyz], [None]
    feat = None
    for li, sa in enumerate(model.sa):
        fidx = inter["fps%d" % li][b].long()
        gidx = inter["ball%d" % li][b].long()               # (M, K), padded with the first hit
        cur = level_xyz[-1]
        ctr = cur[:, fidx]                                  # (3, M)
        rel = cur[:, gidx] - ctr[:, :, None]                # (3, M, K), exact in fp64
        layers = sa["layers"]
        if sa["mlp1"] is not None:                          # xyz-only first layer: fp32 fmas in the loader
            w = sa["mlp1"].double()                         # (C1, 4) = [W_xyz | b]
            h = torch.einsum("cd,dmk->mkc", w[:, :3], rel) + w[:, 3]
        else:                                               # feature part per POINT, xyz part in the loader
            pre = sa["pre"]
            assert pre is not None
            fpre = _lin(feat, pre["la"], relu=False).float().double()      # stored as fp32
            w = pre["w1"].double()
            h = fpre[gidx] + torch.einsum("cd,dmk->mkc", w[:, :3], rel) + w[:, 3]
        h = h.clamp_min(0)
        M, K, _ = h.shape
        h = h.reshape(M * K, -1)
        for layer in layers[1:]:
            h = _lin(h, layer, relu=True)
        feat = h.reshape(M, K, -1).max(dim=1).values.float().double()
        level_xyz.append(ctr)
        level_feat.append(feat)
        if trace is not None:
            trace["sa%d" % li] = feat

    sparse = level_feat[-1]
    for fi, fp in enumerate(model.fp):
        dense = level_feat[-2 - fi]
        nidx = inter["nn%d" % fi][b].long()                 # (n_dense, 3)
        nw = inter["nnw%d" % fi][b].double()
        fl = fp["layers"]
        c2 = sparse.shape[1]
        c1 = 0 if dense is None else dense.shape[1]
        la, lb, _ = model._fp_split(fp, fl[0], c2, c1)
        s = _lin(sparse, la, relu=False).float().double()   # W_a . sparse features, per SPARSE point
        x = (s[nidx] * nw[:, :, None]).sum(dim=1) + fl[0].bias.double()
        if lb is not None:
            x = x + _lin(dense, lb, relu=False).float().double()
        x = x.clamp_min(0)
        if trace is not None:
            trace["fp%d.s" % fi] = s
            trace["fp%d.0" % fi] = x
        for layer in fl[1:]:
            x = _lin(x, layer, relu=True)
        sparse = x
        if trace is not None:
            trace["fp%d" % fi] = x

    return _heads_reference(model, sparse)


class GemmCapture:
    """Wraps `model._gemm` and keeps a copy of every launch's output tensor by launch name
    ("+L2" appended where a second layer rides in the same launch)."""

    def __init__(self, model):
        self.model, self.out, self._orig = model, {}, model._gemm

    def __enter__(self):
        def spy(name, layer, P, loader, epi, **kw):
            self._orig(name, layer, P, loader, epi, **kw)
            if kw.get("out2") is not None:
                # two layers that read the same tensor as one launch ("a|b"): each half under its own name
                first, second = name.split("|")
                self.out[first], self.out[second] = kw["out"].clone(), kw["out2"].clone()
            elif kw.get("out") is not None:
                self.out[name + ("+L2" if kw.get("layer2") is not None else "")] = kw["out"].clone()
        self.model._gemm = spy
        return self

    def __exit__(self, *exc):
        self.model._gemm = self._orig
        return False


def bf16_stagewise_errors(model, pts, inter, cap, outs, b):
    """[(stage, max |diff| / max |ref|, mean |diff| / max |ref|)] of scene b for every launch of the
    shipped configuration's bf16 forward; `cap` = GemmCapture.out of that forward, `outs` its result."""
    res = []

    def rows(t, n):                     # scene b's block of a (B * n, C) launch output
        return t[b * n:(b + 1) * n].double()

    def cmp(name, got, want):
        s = want.abs().max().item()
        d = (got - want).abs()
        res.append((name, d.max().item() / s, d.mean().item() / s))

    xyz = pts[b].double()
    level_xyz, level_feat = [xyz], [None]
    for li, sa in enumerate(model.sa):
        fidx = inter["fps%d" % li][b].long()
        gidx = inter["ball%d" % li][b].long()
        cur = level_xyz[-1]
        ctr = cur[:, fidx]
        rel = cur[:, gidx] - ctr[:, :, None]
        layers = sa["layers"]
        if sa["mlp1"] is not None:
            w = sa["mlp1"].double()
            h = torch.einsum("cd,dmk->mkc", w[:, :3], rel) + w[:, 3]
        else:
            pre = sa["pre"]
            n_in = cur.shape[1]
            got = rows(cap["sa%d.0f" % li], n_in)
            cmp("sa%d.0f" % li, got, _lin(level_feat[-1], pre["la"], relu=False))
            w = pre["w1"].double()
            h = got[gidx] + torch.einsum("cd,dmk->mkc", w[:, :3], rel) + w[:, 3]
        h = h.clamp_min(0)
        M, K, _ = h.shape
        h = h.reshape(M * K, -1)
        for layer in layers[1:]:
            h = _lin(h, layer, relu=True)
        got = rows(cap["sa%d.1+L2" % li], M)
        cmp("sa%d.1+sa%d.2" % (li, li), got, h.reshape(M, K, -1).max(dim=1).values)
        level_xyz.append(ctr)
        level_feat.append(got)          # the next stage starts from the HIP path's tensor

    sparse = level_feat[-1]
    n_fp = len(model.fp)
    for fi, fp in enumerate(model.fp):
        dense = level_feat[-2 - fi]
        nidx = inter["nn%d" % fi][b].long()
        nw = inter["nnw%d" % fi][b].double()
        n_dense, n_sparse = nidx.shape[0], level_xyz[len(model.sa) - fi].shape[1]
        fl = fp["layers"]
        if fi == 0 or "fp%d.0s" % fi in cap:
            la, lb, _ = model._fp_split(fp, fl[0], sparse.shape[1], 0 if dense is None else dense.shape[1])
            s = rows(cap["fp%d.0s" % fi], n_sparse)
            cmp("fp%d.0s" % fi, s, _lin(sparse, la, relu=False))
        else:                            # produced by the previous level's chain launch
            lb = None
            s = sparse
        x = (s[nidx] * nw[:, :, None]).sum(dim=1) + fl[0].bias.double()
        if dense is not None:
            y = rows(cap["fp%d.0d" % fi], n_dense)
            cmp("fp%d.0d" % fi, y, _lin(dense, lb, relu=False))
            x = x + y
        x = x.clamp_min(0)
        if fi == n_fp - 1:               # fp(last).1, .2 and the heads are ONE launch
            for layer in fl[1:]:
                x = _lin(x, layer, relu=True)
            ref = _heads_reference(model, x)
            for k in ref:
                cmp("fp%d tail + heads: %s" % (fi, k), outs[k][b].double(), ref[k])
            break
        key = "fp%d.1+L2" % fi
        if key in cap:                   # second layer + the NEXT level's linear first layer
            nl = model.fp[fi + 1]["layers"][0]
            nsp = model._fp_split(model.fp[fi + 1], nl, fl[1].cout, 0)
            want = _lin(_lin(x, fl[1], relu=True), nsp[0], relu=False)
            sparse = rows(cap[key], n_dense)
            cmp("fp%d.1+fp%d.0s" % (fi, fi + 1), sparse, want)
        else:
            for l, layer in enumerate(fl[1:], 1):
                x = _lin(x, layer, relu=True)
            sparse = rows(cap["fp%d.%d" % (fi, len(fl) - 1)], n_dense)
            cmp("interp_add + fp%d.1" % fi, sparse, x)
    return res

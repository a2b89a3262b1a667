"""Inference fast path of the S4G network on MI355X.

Same function as `model.PointNet2.forward` in eval mode (reference
`network_models/models/PointNet2_tcls.py:99-148`), restructured for the hardware instead of for
torch's operator set.  What a forward pass launches (round 3, default f16x2 arithmetic):

  * BatchNorm (eval) is folded into the 1x1-conv weights once (fp64, rounded once):
    W' = W * gamma / sqrt(var + eps),  b' = beta - mean * gamma / sqrt(var + eps)
    (reference nn_utils/conv.py:28-34: conv -> bn -> relu);
  * activations are channels-last fp32 in HBM; a contraction splits its operands into two scaled
    fp16 planes and runs three `v_mfma_f32_32x32x16_f16` products per MAC with fp32 accumulation
    (fp32-class error); `precision=` selects fp32 MFMA, the exact 3 x bf16 split or plain bf16;
  * SET ABSTRACTION level l: the first shared-MLP layer is linear, so its feature part is applied
    once per POINT (`sa{l}.0f`, a plain tiled launch) and the remaining two layers + the max over
    the K neighbours are ONE chain launch whose loader gathers that row, adds the xyz part + bias
    and applies the ReLU (`sa{l}.1+sa{l}.2`, intermediate activations stay in LDS); level 0 has no
    input features: its first layer is evaluated in the loader from pre-gathered (xyz_j - centre)
    records;
  * FEATURE PROPAGATION level fi: the first layer is applied to the sparse and to the skip features
    BEFORE the 3-NN interpolation (`fp{fi}.0s`, `fp{fi}.0d`); level 0 sums through
    `interp_add_cl_kernel` and runs its 1 024-wide second layer as a tiled launch; level 1's sum
    is formed in the loader of a chain launch that also applies level 2's linear first layer
    (`fp1.1+fp2.0s`); level 2's sum and its two 256-wide layers run INSIDE the heads launch;
  * the four heads (`PointNet2_tcls.py:126-140`) with that tail in front are ONE launch
    (`s4g_heads_chain_f32`): the 64-position input panel and every hidden activation stay in LDS,
    the logits leave as channel-first rows;
  * FPS also emits the centroid coordinates (no gather launch), ball query and 3-NN emit int32
    indices, 3-NN emits the interpolation weights directly.
Twelve contraction launches per forward pass.  `submit()` runs the coordinate-only work (FPS
pyramid, ball queries, 3-NN) on high-priority geometry streams underneath the previous batch's
contractions.

The geometry uses exactly the kernels behind the operator API, so indices are bit-identical to
`functions.py`; per-scene activation scales make a scene's outputs independent of its batch.
"""
import ctypes
import os

import torch

from . import _cabi
from . import functions as _F

_i32 = ctypes.c_int32
_fp = ctypes.c_void_p
GemmDesc = _cabi.GemmDesc
HeadsDesc = _cabi.HeadsDesc

LOAD_PLAIN, LOAD_GATHER, LOAD_INTERP, LOAD_GATHER_MLP1, LOAD_GATHER_ADD, LOAD_INTERP_ADD = 0, 1, 2, 3, 4, 5
EPI_STORE, EPI_MAX, EPI_CF = 0, 1, 2


def fold_conv_bn(block):
    """(W', b') of one conv->bn(->relu) block, fp64 folding rounded once to fp32."""
    w = block.conv.weight.detach().double().flatten(1)      # (Cout, Cin)
    if block.bn is None:
        b = block.conv.bias.detach().double() if block.conv.bias is not None else \
            torch.zeros(w.shape[0], dtype=torch.float64, device=w.device)
        return w.float(), b.float()
    bn = block.bn
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    wf = w * scale[:, None]
    bf = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    if block.conv.bias is not None:
        bf = bf + block.conv.bias.detach().double() * scale
    return wf.float(), bf.float()


def _pad_k(w, mult=8):
    cout, k = w.shape
    kpad = (k + mult - 1) // mult * mult
    if kpad == k:
        return w.contiguous()
    out = w.new_zeros((cout, kpad))
    out[:, :k] = w
    return out


def split_bf16x3(w):
    """Exact 3-way bf16 split of an fp32 tensor: w == w1 + w2 + w3 (24 significand
    bits).  Returns a (3, ...) bfloat16 tensor (hi, mid, lo planes)."""
    w1 = w.to(torch.bfloat16)
    r1 = w - w1.float()
    w2 = r1.to(torch.bfloat16)
    r2 = r1 - w2.float()
    w3 = r2.to(torch.bfloat16)
    return torch.stack([w1, w2, w3], dim=0).contiguous()


def split_f16x2(w):
    """Scaled 2-way fp16 split of an fp32 tensor (..., Cout, K): per output row a
    power-of-two scale s puts max|w| in [2^14, 2^15); w*s ~= w1 + w2 (fp16,
    round-to-nearest; 22 significand bits).  Returns ((2, ..., Cout, K) float16
    planes, (..., Cout) float32 1/s)."""
    amax = w.abs().amax(dim=-1)
    _, ex = torch.frexp(amax)                       # amax < 2^ex
    ex = torch.where(amax > 0, ex, torch.zeros_like(ex)).clamp(-100, 100)
    one = torch.ones_like(amax)
    s = torch.ldexp(one, 15 - ex)
    ws = w * s[..., None]                           # exact (power of two)
    w1 = ws.to(torch.float16)
    w2 = (ws - w1.float()).to(torch.float16)
    return torch.stack([w1, w2], dim=0).contiguous(), torch.ldexp(one, ex - 15).contiguous()


def fragment_order(planes):
    """(PL, G, Cout, K16) 16-bit planes -> (G, Cout/32, K16/16, PL, 64, 8): each 32-channel x
    16-deep MFMA operand fragment as one contiguous 1 KB block per plane, lane =
    32 * (k // 8 % 2) + n % 32 (the 32x32x16 A/B operand layout)."""
    if planes.dim() == 3:
        planes = planes.unsqueeze(1)
    npl, G, cout, k16 = planes.shape
    x = planes.reshape(npl, G, cout // 32, 32, k16 // 16, 2, 8)  # pl g n32 li ks lh i
    return x.permute(1, 2, 4, 0, 5, 3, 6).contiguous()           # g n32 ks pl lh li i


class _Layer:
    """Folded weights of one launch: W (groups, Cout, Kpad), bias (groups, Cout),
    plus the bf16x3 planes (3, groups, Cout, Kpad16) and the scaled fp16x2 planes
    (2, groups, Cout, Kpad16) + per-channel inverse scales for the split kernels."""

    def __init__(self, W, bias, cin, groups=1):
        self.W = W.contiguous()
        self.bias = bias.contiguous()
        self.groups = groups
        self.cout = self.W.shape[-2]
        self.kpad = self.W.shape[-1]
        self.cin = cin
        self.kpad16 = (self.kpad + 15) // 16 * 16
        w16 = self.W.new_zeros(self.W.shape[:-1] + (self.kpad16,))
        w16[..., :self.kpad] = self.W
        self.W3 = split_bf16x3(w16)
        self.Wh2, self.w_inv_scale = split_f16x2(w16)
        self.Wfrag = fragment_order(self.Wh2) if self.cout % 32 == 0 else None
        # the hi bf16 plane alone in the same fragment order: the single-product chains
        self.Wfrag_bf16 = fragment_order(self.W3[:1])[:, :, :, 0].contiguous() if self.cout % 32 == 0 else None


_STREAM_POOL = {}


def _stream_pool(dev, n_geo, n_dense):
    """The geometry / contraction streams of a device, shared by every FusedPointNet2 in the process.  HIP multiplexes
    a process's streams onto a handful of hardware queues; a process that had built several runners (bench.py's legs:
    one per precision / network) ended up with a runner whose geometry stream and contraction stream shared a hardware
    queue -- the FPS chain then serialised with the contractions (configs[4]: 16.7 instead of 9.8 ms per step, round 6).
    One set of streams per (device, counts): a second runner queues behind the first on the same streams, which is what
    two users of one GPU want anyway."""
    key = (str(dev), int(n_geo), int(n_dense))
    pool = _STREAM_POOL.get(key)
    if pool is None:
        pool = _STREAM_POOL[key] = ([torch.cuda.Stream(device=dev, priority=-1) for _ in range(n_geo)],
                                    [torch.cuda.Stream(device=dev) for _ in range(n_dense)])
    return pool


class FusedPointNet2:
    """Callable with the reference forward's signature: {"scene_points": (B,3,N)} -> dict."""

    def __init__(self, net, precision=None, fold_only=False, check_finite=False):
        """net: `model.PointNet2` or the reference's own `PointNet2_tcls.PointNet2` instance (the attributes read
        are the reference's: `sa_modules[i].{sampler, grouper.{radius, num_neighbours}, mlp, in_channels,
        num_centroids}`, `fp_modules[i].{interpolator._eps, mlp}`, `mlp_seg / seg_logit / mlp_R / R_logit / mlp_t /
        t_logit / mlp_movable / movable_logit[0]`, blocks with `.conv` / `.bn`: `PointNet2_tcls.py:56-95`;
        tests/test_reference_dropin.py passes the reference's object).
        fold_only: fold and pack the weights wherever the model lives and stop -- no device, no library; the
        object cannot run a forward (tools / tests that compare packed weights on a host without a GPU).
        precision: "f16x2" (default; fp32 operands split into two scaled fp16
        planes, three fp16 MFMA products, fp32 accumulate: the error of a plain fp32
        dot product), "bf16x3" (three bf16 planes, six products: ~2x tighter than
        fp32 round-off, half the speed), "fp32" (fp32-input MFMA, an exact fma
        chain) or "bf16" (plain bf16 inputs, fp32 accumulate -- reduced precision,
        outside the 1e-4 bar; the bf16 roofline configuration of BASELINE.json
        configs[4]).  S4G_GEMM_MODE overrides the default.
        check_finite: (debug) `submit` refuses a batch with a NaN / inf coordinate, naming the scenes -- costs one
        device synchronisation per call.  Non-finite coordinates are out of contract (as for the reference's kernels:
        SURVEY.md Appendix A.1); without the check a non-finite scene is contained -- it can neither fault nor touch
        another scene's results (per-scene grids and per-scene activation scales: tests/test_batch_invariance_gpu.py)
        -- but ITS outputs are unspecified: the contraction kernels are built with -fno-honor-nans (csrc/Makefile), so
        where the reference would hand a visible NaN through ReLU / max-pool, these may return finite garbage."""
        self.check_finite = bool(check_finite)
        if precision is None:
            precision = os.environ.get("S4G_GEMM_MODE", "f16x2")
        if precision not in ("f16x2", "bf16x3", "fp32", "bf16"):
            raise ValueError("precision must be 'f16x2', 'bf16x3', 'fp32' or 'bf16'")
        self.precision = precision
        self.dense_streams = max(1, int(os.environ.get("S4G_DENSE_STREAMS", "1")))
        # S4G_GEMM_FUSE2=0: never fuse the last two layers of an SA level into one launch
        self.fuse2 = _cabi.knob("S4G_GEMM_FUSE2", "1") != "0"
        self.fuse3 = _cabi.knob("S4G_GEMM_FUSE3", "1") != "0"   # + first head layer
        # S4G_FP_LINEAR_FIRST=0: interpolate first, like the reference (fused into the
        # contraction's loader); default: first FP layer before the interpolation
        self.fp_linear_first = _cabi.knob("S4G_FP_LINEAR_FIRST", "1") != "0"
        # FP levels whose interpolate + add + ReLU happens in the next launch's loader:
        # S4G_FP_LOADER_ADD = "auto" (default: where that launch is a fused chain, whose panel is
        # loaded once -- the tiled kernel would repeat the gathers per column tile: measured
        # 0.41 ms against 0.14 + 0.18 ms at FP level 1), "none", or a comma list of levels
        v = _cabi.knob("S4G_FP_LOADER_ADD", "auto")
        self.fp_loader_add = v if v in ("auto", "none") else set(int(t) for t in v.split(",") if t)
        self.geo_streams = max(1, int(os.environ.get("S4G_GEO_STREAMS", "2")))
        self.rel_xyz = _cabi.knob("S4G_REL_XYZ", "1") != "0"
        # first SA level on a centroid's DISTINCT rows only (ball_query pads short balls with copies of the
        # first hit, the max over the neighbours cannot see them): S4G_SA_UNIQUE=0 contracts all K rows
        self.sa_unique = _cabi.knob("S4G_SA_UNIQUE", "1") != "0"
        self.heads_pre = _cabi.knob("S4G_HEADS_PRE", "1") != "0"
        # layers that read the SAME tensor as one launch with two outputs (an SA level's per-point layer and
        # the mirror FP level's skip-feature layer both read that level's features): S4G_MERGE_SHARED=0 splits
        self.merge_shared = _cabi.knob("S4G_MERGE_SHARED", "1") != "0"
        self.fp_chain_next = _cabi.knob("S4G_FP_CHAIN_NEXT", "1") != "0"
        self.fps_prefix = _cabi.knob("S4G_FPS_PREFIX", "1") != "0"
        # the four head tensors as channel slices of ONE (B, 21, N) tensor (`PackedPred.packed`: the payload of
        # the multi-GPU all-gather, dist.py, without a packing copy); S4G_PACKED_OUT=0: four tensors of their own
        self.packed_out = _cabi.knob("S4G_PACKED_OUT", "1") != "0"
        p = next(net.parameters())
        self.fold_only = bool(fold_only)
        if not p.is_cuda and not fold_only:
            raise RuntimeError("FusedPointNet2 needs the model on a HIP device (no CPU fallback)")
        if net.training:
            raise RuntimeError("FusedPointNet2 is inference-only: call net.eval() first")
        self.dev = p.device
        if not fold_only:
            _cabi.lib()
        self.sa = []
        for sa in net.sa_modules:
            if sa.sampler is None or sa.grouper is None:
                raise NotImplementedError("fast path covers sampled + grouped SA modules only")
            if sa.grouper.num_neighbours not in (16, 32, 64):
                raise NotImplementedError("fast path needs num_neighbours in {16, 32, 64}")
            layers = []
            mlp1 = None
            for li, blk in enumerate(sa.mlp):
                w, b = fold_conv_bn(blk)
                if li == 0 and sa.in_channels == 0 and len(sa.mlp) > 1 and w.shape[0] % 4 == 0:
                    # xyz-only first layer (3 -> C1): evaluated inside the next layer's
                    # loader (3 fmas per channel) instead of a launch that writes
                    # (B*M*K, C1) to HBM and reads it back
                    mlp1 = torch.cat([w[:, :3], b[:, None]], dim=1).contiguous()   # (C1, 4)
                if li == 0:
                    # reference K order [xyz(3), feat(C)] (modules.py:50) -> ours [feat, xyz]
                    w = torch.cat([w[:, 3:], w[:, :3]], dim=1)
                    cin = w.shape[1]
                else:
                    cin = w.shape[1]
                layers.append(_Layer(_pad_k(w), b, cin))
            radius = float(sa.grouper.radius)
            pre = None
            if (sa.in_channels > 0 and sa.in_channels % 4 == 0 and len(sa.mlp) > 1 and
                    layers[0].cout % 4 == 0 and _cabi.knob("S4G_SA_LINEAR_FIRST", "1") != "0"):
                # a level WITH input features: its first layer is linear, so the feature part is
                # applied once per POINT (N rows) instead of once per (centroid, neighbour) pair
                # (M*K rows, every point ~K*M/N times); the xyz part + bias + ReLU move into the
                # next layer's loader
                w0 = layers[0].W[:, :layers[0].cin]                 # K order [feat, xyz]
                cf = sa.in_channels
                la = _Layer(_pad_k(w0[:, :cf].contiguous()), torch.zeros_like(layers[0].bias), cf)
                w1 = torch.cat([w0[:, cf:cf + 3], layers[0].bias[:, None]], dim=1).contiguous()
                bound = float((w1[:, :3].abs().sum(dim=1) * radius + w1[:, 3].abs()).max())
                pre = dict(la=la, w1=w1, bound=bound)
            # |relu(w . rel + b)| <= (|wx|+|wy|+|wz|) r + |b| for neighbours inside the ball
            mlp1_bound = 0.0 if mlp1 is None else float(
                (mlp1[:, :3].abs().sum(dim=1) * radius + mlp1[:, 3].abs()).max())
            self.sa.append(dict(M=sa.num_centroids, radius=radius,
                                K=int(sa.grouper.num_neighbours), layers=layers,
                                cf=sa.in_channels, mlp1=mlp1, mlp1_bound=mlp1_bound, pre=pre))
        self.fp = []
        for fp in net.fp_modules:
            if fp.interpolator is None:
                raise NotImplementedError("fast path covers 3-NN FP modules only")
            layers = []
            for blk in fp.mlp:
                w, b = fold_conv_bn(blk)
                layers.append(_Layer(_pad_k(w), b, w.shape[1]))
            self.fp.append(dict(layers=layers, eps=float(fp.interpolator._eps),
                                bias0_max=float(layers[0].bias.abs().max())))
        # heads: order score, R, t, movable (PointNet2_tcls.py:126-140)
        heads = [(net.mlp_seg, net.seg_logit), (net.mlp_R, net.R_logit), (net.mlp_t, net.t_logit),
                 (net.mlp_movable, net.movable_logit[0])]
        depth = len(net.mlp_seg)
        self.head_layers = []
        folded = [[fold_conv_bn(blk) for blk in mlp] for mlp, _ in heads]
        w0 = torch.cat([f[0][0] for f in folded], dim=0)          # (4*C1, Cin) shared input
        b0 = torch.cat([f[0][1] for f in folded], dim=0)
        self.head_layers.append(_Layer(_pad_k(w0), b0, w0.shape[1]))
        for li in range(1, depth):
            W = torch.stack([_pad_k(f[li][0]) for f in folded], dim=0)
            bias = torch.stack([f[li][1] for f in folded], dim=0)
            self.head_layers.append(_Layer(W, bias, folded[0][li][0].shape[1], groups=4))
        chans = [lg.weight.shape[0] for _, lg in heads]
        self.head_channels = chans
        cl = heads[0][1].weight.shape[1]
        wl = torch.zeros((sum(chans), 4 * cl), dtype=torch.float32, device=self.dev)
        bl = torch.zeros((sum(chans),), dtype=torch.float32, device=self.dev)
        r = 0
        for h, (_, lg) in enumerate(heads):
            c = lg.weight.shape[0]
            wl[r:r + c, h * cl:(h + 1) * cl] = lg.weight.detach().flatten(1)
            bl[r:r + c] = lg.bias.detach()
            r += c
        self.logit_layer = _Layer(_pad_k(wl), bl, 4 * cl)
        self.sigmoid_from = sum(chans[:3])
        self._streams = None
        # all four heads as ONE launch (s4g_heads_chain_f32: input panel and every hidden
        # activation stay in LDS) where the widths are the shipped ones; S4G_HEADS_FUSED=0 keeps
        # the layer-chain launches
        hl = self.head_layers
        self.heads_fused = None
        if (_cabi.knob("S4G_HEADS_FUSED", "1") != "0" and precision in ("f16x2", "bf16") and depth == 4 and
                hl[0].cin == 256 and hl[0].cout == 4 * 512 and
                [(l.cout, l.cin, l.groups) for l in hl[1:]] == [(256, 512, 4), (256, 256, 4), (128, 256, 4)] and
                cl == 128 and max(chans) <= 32 and len(chans) == 4):
            wpad = torch.zeros((4, 32, cl), dtype=torch.float32, device=self.dev)
            bpad = torch.zeros((4, 32), dtype=torch.float32, device=self.dev)
            for h, (_, lg) in enumerate(heads):
                c = lg.weight.shape[0]
                wpad[h, :c] = lg.weight.detach().flatten(1)
                bpad[h, :c] = lg.bias.detach()
            self.heads_fused = list(hl) + [_Layer(wpad, bpad, cl, groups=4)]

    def packed_weights(self):
        """{name: tensor} of every folded / split / fragment-ordered weight tensor the launches read -- what must
        be equal for two networks to run the same forward (tests/test_reference_dropin.py)."""
        out = {}

        def put(prefix, layer):
            for a in ("W", "bias", "W3", "Wh2", "w_inv_scale", "Wfrag", "Wfrag_bf16"):
                t = getattr(layer, a)
                if t is not None:
                    out["%s.%s" % (prefix, a)] = t
        for li, sa in enumerate(self.sa):
            for l, layer in enumerate(sa["layers"]):
                put("sa%d.%d" % (li, l), layer)
            if sa["mlp1"] is not None:
                out["sa%d.mlp1" % li] = sa["mlp1"]
            if sa["pre"] is not None:
                put("sa%d.pre" % li, sa["pre"]["la"])
                out["sa%d.pre.w1" % li] = sa["pre"]["w1"]
            out["sa%d.meta" % li] = torch.tensor([sa["M"], sa["K"], sa["cf"], sa["radius"], sa["mlp1_bound"],
                                                  0.0 if sa["pre"] is None else sa["pre"]["bound"]],
                                                 dtype=torch.float64)
        for fi, fp in enumerate(self.fp):
            for l, layer in enumerate(fp["layers"]):
                put("fp%d.%d" % (fi, l), layer)
            out["fp%d.meta" % fi] = torch.tensor([fp["eps"], fp["bias0_max"]], dtype=torch.float64)
        for l, layer in enumerate(self.head_layers):
            put("heads.%d" % l, layer)
        put("heads.logits", self.logit_layer)
        if self.heads_fused is not None:
            put("heads.logits_padded", self.heads_fused[-1])
        return out

    def _fusable(self, l1, l2, loader=LOAD_PLAIN, epi=EPI_STORE):
        """Two consecutive layers one launch can take: widths that chain (C -> C -> Cout2) and a
        fused form of the contraction for this loader / epilogue / first-layer depth (the
        library is asked: s4g_gemm_chain_supported)."""
        c = l1.cout
        if not (self.precision in ("f16x2", "bf16") and self.fuse2 and l1.groups == l2.groups and
                l2.cin == c and l2.kpad16 == c and l2.cout % 64 == 0 and
                l1.Wfrag is not None and l2.Wfrag is not None):
            return False
        if c == 512 and _cabi.knob("S4G_GEMM_FUSE512", "1") == "0":
            return False
        if l1.kpad16 != c and not self.fuse3:
            return False
        return bool(_cabi.lib().s4g_gemm_chain_supported(loader, epi, c, l1.kpad16))

    @staticmethod
    def _fp_split(fp, layer, c2, c1):
        """First FP layer W = [W_a | W_b] over [interpolated (c2), skip (c1)] columns as two layers
        without bias (modules.py:124-127 concat order), cached on the level."""
        sp = fp.get("split")
        if sp is None or sp[2] != (c2, c1):
            w = layer.W[:, :layer.cin]
            zero = torch.zeros_like(layer.bias)
            la = _Layer(_pad_k(w[:, :c2].contiguous()), zero, c2)
            lb = _Layer(_pad_k(w[:, c2:].contiguous()), zero, c1) if c1 > 0 else None
            sp = fp["split"] = (la, lb, (c2, c1))
        return sp

    def _shared_input_layer(self, li, c1):
        """SA level li's per-point layer (`sa{li}.0f`) and FP level n_sa-1-li's skip-feature layer
        (`fp{fi}.0d`) read the same (B N_li, c1) tensor: (merged layer with W rows concatenated, the FP
        half) or None.  Per-channel weight scales: every output channel is bit-identical to the two
        separate launches."""
        sa, fi = self.sa[li], len(self.sa) - 1 - li
        if (not self.merge_shared or self.precision not in ("f16x2", "bf16") or sa["pre"] is None or
                not 0 <= fi < len(self.fp) or not self.fp_linear_first):
            return None
        fp, la = self.fp[fi], sa["pre"]["la"]
        l0 = fp["layers"][0]
        c2 = l0.cin - c1
        if c2 <= 0 or la.cin != c1 or la.cout % 256 or l0.cout % 256 or l0.cout > 1024 or len(fp["layers"]) < 2:
            return None
        _, lb, _ = self._fp_split(fp, l0, c2, c1)
        m = sa.get("merged")
        if m is None or m[1] is not lb:
            w = torch.cat([la.W[:, :c1], lb.W[:, :c1]], dim=0).contiguous()
            m = sa["merged"] = (_Layer(_pad_k(w), w.new_zeros(w.shape[0]), c1), lb)
        return m

    def _heads_take_tail(self, fi, fl, pending):
        """Last FP level, heads as one launch, the level = (linear-first layer, 256 -> 256, 256 -> 256)
        with its sum formed in a loader: the two 256-wide layers move into the heads' launch
        (S4G_HEADS_PRE=0 keeps the separate chain launch)."""
        return (self.heads_fused is not None and self.heads_pre and fi == len(self.fp) - 1 and len(fl) == 3 and
                pending["C2"] == 256 and all(l.cin == 256 and l.cout == 256 and l.groups == 1 and
                                             l.Wfrag is not None for l in fl[1:]) and
                self.head_layers[0].cin == 256)

    # ------------------------------------------------------------------ launches
    def _gemm(self, name, layer, P, loader, epi, relu=True, layer2=None, layer3=None, name3="heads.0",
              rows_per_scene=0, relu2=True, name2=None, rows_used=None, **kw):
        d = GemmDesc()
        d.loader, d.epilogue, d.groups, d.relu = loader, epi, layer.groups, int(relu)
        d.P, d.Cin, d.Kpad, d.Cout = P, layer.cin, layer.kpad, layer.cout
        # every launch's rows are B equal blocks, one per scene: the per-scene amax rows follow
        d.rows_per_scene = rows_per_scene or (P // self._batch if P % self._batch == 0 else 0)
        d.W, d.bias = layer.W.data_ptr(), layer.bias.data_ptr()
        d.w_gstride, d.b_gstride = layer.cout * layer.kpad, layer.cout
        d.precision = {"fp32": 0, "bf16x3": 1, "bf16": 2, "f16x2": 3}[self.precision]
        d.Kpad16, d.W_bf16x3 = layer.kpad16, layer.W3.data_ptr()
        d.W_f16x2, d.w_inv_scale = layer.Wh2.data_ptr(), layer.w_inv_scale.data_ptr()
        bf16 = self.precision == "bf16"      # chains take ONE bf16 plane in fragment order
        frag = (lambda l: l.Wfrag_bf16) if bf16 else (lambda l: l.Wfrag)
        if layer.Wfrag is not None:     # (bf16: the one-plane fragments; single layers take them too)
            d.W_f16x2_frag = frag(layer).data_ptr()
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                v = v.data_ptr()
            if v is not None:
                setattr(d, k, v)
        flops = 2.0 * P * layer.cout * layer.cin * layer.groups
        if layer2 is not None:   # second layer fused behind this one (intermediate stays in LDS)
            d.W2_f16x2_frag = frag(layer2).data_ptr()
            d.w2_inv_scale, d.bias2 = layer2.w_inv_scale.data_ptr(), layer2.bias.data_ptr()
            d.Cout2, d.relu2 = layer2.cout, int(relu2)
            flops += 2.0 * P * layer2.cout * layer2.cin * layer2.groups
            name = "%s+%s" % (name, name2 or name[:-1] + str(int(name[-1]) + 1))
        if layer3 is not None:   # ... and a third one (layer 2's output stays in LDS too)
            d.W3_f16x2_frag = frag(layer3).data_ptr()
            d.w3_inv_scale, d.bias3 = layer3.w_inv_scale.data_ptr(), layer3.bias.data_ptr()
            d.Cout3, d.relu3 = layer3.cout, 1
            flops += 2.0 * P * layer3.cout * layer3.cin * layer3.groups
            name += "+" + name3
        if rows_used is not None:
            # a data-dependent row count (the distinct-row SA level): the flops of the rows that exist,
            # read back when the timers are summarised (after the timed region's fence)
            per_row = flops / P
            flops = lambda: per_row * float(rows_used.sum().item())
        with _F._timed("gemm[%s P=%d K=%d N=%dx%d]" % (name, P, layer.cin, layer.groups, layer.cout),
                       0, flops):
            rc = _cabi.lib().s4g_mlp_gemm_f32(ctypes.byref(d), _F._stream())
        _cabi.check(rc, "mlp_gemm " + name)

    def _heads(self, x, x_amax, outs, B, N0, pre=None, head_mask=0):
        """heads.0 .. heads.3 + logits of all four heads: one launch (s4g_heads_chain_f32).
        pre: the last FP level's tail rides in front (its loader state + its two 256-wide layers):
        the per-point feature tensor between the FP stack and the heads never exists in HBM."""
        d = HeadsDesc()
        bf16 = self.precision == "bf16"
        d.precision = 2 if bf16 else 3
        d.P, d.N = B * N0, N0
        d.C, d.H0, d.H1, d.H2, d.H3 = 256, 512, 256, 256, 128
        flops = 0.0
        name = "heads.0-3+logits"
        if pre is None:
            d.X, d.ldx = x.data_ptr(), x.shape[1]
        else:
            frag = (lambda l: l.Wfrag_bf16) if bf16 else (lambda l: l.Wfrag)
            for i, layer in enumerate(pre["layers"]):
                d.pre_W_frag[i] = frag(layer).data_ptr()
                d.pre_bias[i] = layer.bias.data_ptr()
                d.pre_w_inv_scale[i] = layer.w_inv_scale.data_ptr()
                flops += 2.0 * B * N0 * layer.cout * layer.cin
            d.pre_nidx, d.pre_nw = pre["nidx"].data_ptr(), pre["nw"].data_ptr()
            d.pre_sparse, d.pre_N2 = pre["sparse"].data_ptr(), pre["N2"]
            d.pre_dense = None if pre["dense"] is None else pre["dense"].data_ptr()
            d.pre_lbias = pre["loader_bias"].data_ptr()
            d.pre_a_amax2 = None if pre["a_amax2"] is None else pre["a_amax2"].data_ptr()
            x_amax = pre["a_amax"]
            name = "%s+%s" % (pre["name"], name)
        nrun = len([h for h in range(4) if (head_mask or 15) >> h & 1])
        for l, layer in enumerate(self.heads_fused):
            d.W_frag[l] = (layer.Wfrag_bf16 if bf16 else layer.Wfrag).data_ptr()
            d.bias[l] = layer.bias.data_ptr()
            d.w_inv_scale[l] = layer.w_inv_scale.data_ptr()
            if l < 4:     # (layer 0 is the four first layers stacked along Cout, the others are grouped by head)
                flops += 2.0 * B * N0 * layer.cout * layer.cin * layer.groups * nrun / 4.0
        flops += 2.0 * B * N0 * sum(c for h, c in enumerate(self.head_channels) if (head_mask or 15) >> h & 1) * 128
        d.head_mask = head_mask
        run = [h for h in range(4) if (head_mask or 15) >> h & 1]
        for h, o in enumerate(outs):
            d.channels[h] = self.head_channels[h]
            if h not in run:
                continue
            d.out[h] = o.data_ptr()
            if o.stride(0) != self.head_channels[h] * N0:    # channel slices of the packed (B, 21, N) tensor
                d.out_batch_stride = o.stride(0)
        d.sigmoid_head = 3
        d.a_amax = None if x_amax is None else x_amax.data_ptr()
        d.a_amax_floor = 0.0 if pre is None else pre["a_amax_floor"]
        d.rows_per_scene = N0
        if head_mask not in (0, 15):
            name += "[heads %s]" % "".join(str(h) for h in range(4) if head_mask >> h & 1)
        with _F._timed("gemm[%s P=%d K=256 N=%dx(512,256,256,128,c)]" % (name, B * N0, nrun), 0, flops):
            rc = _cabi.lib().s4g_heads_chain_f32(ctypes.byref(d), _F._stream())
        _cabi.check(rc, "heads_chain")

    def _fps_gather(self, xyz, M, want_dist=False, run=None):
        """idx (B, M) int32 + centroids (B, 3, M).  want_dist: also the picks' min-distances (B, M), or
        None where this size's kernel cannot report them; run (B,) int32: scenes with run == 0 are
        known to sample their own prefix (`_fps_prefix_check`) and are not sampled."""
        B, _, N = xyz.shape
        idx = torch.empty((B, M), dtype=torch.int32, device=xyz.device)
        ctr = torch.empty((B, 3, M), dtype=torch.float32, device=xyz.device)
        ws, nbytes = _F._workspace(_cabi.S4G_OP_FPS, xyz.device, B, N, M, 0)
        dist = torch.empty((B, M), dtype=torch.float32, device=xyz.device) if want_dist else None
        with _F._timed("fps[N=%d,M=%d]" % (N, M), B * (12 * N + 8 * M)):
            if dist is not None or run is not None:
                rc = _cabi.lib().s4g_fps_gather_ex_i32(xyz.data_ptr(), B, N, M, idx.data_ptr(), ctr.data_ptr(),
                                                       None if dist is None else dist.data_ptr(),
                                                       None if run is None else run.data_ptr(),
                                                       _F._ptr(ws), nbytes, _F._DIST_FLAGS, _F._stream())
                if rc == _cabi.S4G_EUNSUPPORTED and run is None:
                    dist, rc = None, 0           # nothing was launched: plain call below
                    rc = _cabi.lib().s4g_fps_gather_i32(xyz.data_ptr(), B, N, M, idx.data_ptr(), ctr.data_ptr(),
                                                        _F._ptr(ws), nbytes, _F._DIST_FLAGS, _F._stream())
            else:
                rc = _cabi.lib().s4g_fps_gather_i32(xyz.data_ptr(), B, N, M, idx.data_ptr(),
                                                    ctr.data_ptr(), _F._ptr(ws), nbytes,
                                                    _F._DIST_FLAGS, _F._stream())
        _cabi.check(rc, "fps_gather")
        return (idx, ctr, dist) if want_dist else (idx, ctr)

    def _fps_prefix_check(self, ctr, dist, M2):
        """(B,) int32, 0 where FPS over these centroids (in pick order) provably re-picks 0..M2-1."""
        B, _, M1 = ctr.shape
        run = torch.empty((B,), dtype=torch.int32, device=ctr.device)
        with _F._timed("fps_prefix_check[M1=%d,M2=%d]" % (M1, M2), B * 16 * M1):
            rc = _cabi.lib().s4g_fps_prefix_check_f32(ctr.data_ptr(), dist.data_ptr(), B, M1, M2, run.data_ptr(),
                                                      _F._DIST_FLAGS, _F._stream())
        _cabi.check(rc, "fps_prefix_check")
        return run

    def _ball_query(self, xyz, ctr, radius, K):
        B, _, N = xyz.shape
        M = ctr.shape[2]
        idx = torch.empty((B, M, K), dtype=torch.int32, device=xyz.device)
        cnt = torch.empty((B, M), dtype=torch.int32, device=xyz.device)
        ws, nbytes = _F._workspace(_cabi.S4G_OP_BALL_QUERY, xyz.device, B, N, M, K)
        with _F._timed("ball_query[N=%d,M=%d,K=%d]" % (N, M, K),
                       B * (12 * N + 12 * M + 8 * M * K + 8 * M)):
            rc = _cabi.lib().s4g_ball_query_i32(xyz.data_ptr(), ctr.data_ptr(), B, N, M, radius, K,
                                                idx.data_ptr(), cnt.data_ptr(), _F._ptr(ws), nbytes,
                                                _F._DIST_FLAGS, _F._stream())
        _cabi.check(rc, "ball_query_i32")
        return idx, cnt

    def _group_rel_xyz(self, xyz, ctr, gidx):
        """(B*M*K, 4) rows (xyz_j - ctr_m, 0): the first SA layer's input, gathered on the geometry
        stream so that the contraction's loader reads one coalesced record per row."""
        B, _, N = xyz.shape
        _, M, K = gidx.shape
        rel = torch.empty((B * M * K, 4), dtype=torch.float32, device=xyz.device)
        with _F._timed("group_rel_xyz[N=%d,M=%d,K=%d]" % (N, M, K), B * M * K * (4 + 12 + 16)):
            rc = _cabi.lib().s4g_group_rel_xyz_i32(xyz.data_ptr(), ctr.data_ptr(), gidx.data_ptr(),
                                                   B, N, M, K, rel.data_ptr(), _F._stream())
        _cabi.check(rc, "group_rel_xyz_i32")
        return rel

    def _group_rel_xyz_unique(self, xyz, ctr, gidx, gcnt):
        """The same records without ball_query's padding copies (s4g_group_rel_xyz_unique_i32): rows
        (B*M*K capacity, 4), seg4 (B*M*K/4) output row per group of 4 rows, rows per scene (B), or None
        where the shape is not supported."""
        B, _, N = xyz.shape
        _, M, K = gidx.shape
        if K % 4 or (M * K) % 256:
            return None
        dev = xyz.device
        rel = torch.empty((B * M * K, 4), dtype=torch.float32, device=dev)
        seg4 = torch.empty((B * M * K // 4,), dtype=torch.int32, device=dev)
        row_start = torch.empty((B, M), dtype=torch.int32, device=dev)
        rows = torch.empty((B,), dtype=torch.int32, device=dev)
        with _F._timed("group_rel_xyz_unique[N=%d,M=%d,K=%d]" % (N, M, K), B * M * K * (4 + 12 + 16)):
            rc = _cabi.lib().s4g_group_rel_xyz_unique_i32(xyz.data_ptr(), ctr.data_ptr(), gidx.data_ptr(),
                                                          gcnt.data_ptr(), B, N, M, K, rel.data_ptr(),
                                                          seg4.data_ptr(), row_start.data_ptr(), rows.data_ptr(),
                                                          _F._stream())
        if rc == _cabi.S4G_EUNSUPPORTED:
            return None
        _cabi.check(rc, "group_rel_xyz_unique_i32")
        return rel, seg4, rows

    def _three_nn(self, q, k, eps, cell=0.0):
        B, _, N1 = q.shape
        N2 = k.shape[2]
        idx = torch.empty((B, N1, 3), dtype=torch.int32, device=q.device)
        w = torch.empty((B, N1, 3), dtype=torch.float32, device=q.device)
        if cell != 0.0 and 2048 <= N2 <= 65536 and _cabi.knob("S4G_NN_MODE", "grid") != "scan":
            # cell > 0: that edge; cell < 0: chosen on the device from the keys' measured spacing
            nbytes = _cabi.lib().s4g_three_nn_grid_workspace_bytes(B, N1, N2)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            with _F._timed("three_nn[N1=%d,N2=%d]" % (N1, N2), B * (12 * N2 + 12 * N1 + 24 * N1 + 12 * N1)):
                rc = _cabi.lib().s4g_three_nn_weights_grid_i32(q.data_ptr(), k.data_ptr(), B, N1, N2,
                                                               eps, cell, idx.data_ptr(),
                                                               w.data_ptr(), ws.data_ptr(), nbytes,
                                                               _F._DIST_FLAGS, _F._stream())
            _cabi.check(rc, "three_nn_weights_grid_i32")
            return idx, w
        with _F._timed("three_nn[N1=%d,N2=%d]" % (N1, N2), B * (12 * N2 + 12 * N1 + 24 * N1 + 12 * N1)):
            rc = _cabi.lib().s4g_three_nn_weights_i32(q.data_ptr(), k.data_ptr(), B, N1, N2, eps,
                                                      idx.data_ptr(), w.data_ptr(), None, 0,
                                                      _F._DIST_FLAGS, _F._stream())
        _cabi.check(rc, "three_nn_weights_i32")
        return idx, w

    def _sa_unique_ok(self, sa):
        """The distinct-row form needs the fused two-layer chain behind the xyz-only first layer."""
        layers = sa["layers"]
        return (self.precision in ("f16x2", "bf16") and len(layers) == 3 and layers[-1].groups == 1 and
                self._fusable(layers[-2], layers[-1], LOAD_GATHER_MLP1, EPI_MAX))

    # ------------------------------------------------------------------ forward
    def _geometry(self, xyz):
        """Everything that depends on coordinates only: the FPS pyramid, the ball
        queries and the 3-NN searches + weights of all levels."""
        B, _, N0 = xyz.shape
        geo = dict(level_xyz=[xyz], level_n=[N0], sa=[], fp=[], sa_events=[], rel=[])
        n_cur = N0
        # every level after the first samples the previous level's centroids, which are in pick
        # order: FPS then re-picks their prefix unless two of them tie (include/s4g_ops.h,
        # s4g_fps_prefix_check_f32).  One check after level 0 over the next level's M covers all
        # deeper levels; scenes that fail it are sampled for real.  S4G_FPS_PREFIX=0: always sample.
        run = None
        for li, sa in enumerate(self.sa):
            M, K = sa["M"], sa["K"]
            if not n_cur >= M:
                raise RuntimeError("num_points is not greater than or equal to num_centroids")
            dist = None
            if li == 0 and self.fps_prefix and len(self.sa) > 1 and self.sa[1]["M"] <= M:
                idx, ctr, dist = self._fps_gather(geo["level_xyz"][-1], M, want_dist=True)
            else:
                idx, ctr = self._fps_gather(geo["level_xyz"][-1], M, run=run if li > 0 else None)
            gidx, gcnt = self._ball_query(geo["level_xyz"][-1], ctr, sa["radius"], K)
            geo["sa"].append((idx, ctr, gidx, gcnt))
            # the xyz-only first layer runs inside the next layer's loader: hand it its rows
            # pre-gathered (S4G_REL_XYZ=0: the loader follows gidx itself)
            rel = None
            if sa["mlp1"] is not None and self.rel_xyz:
                if self.sa_unique and K == 64 and self._sa_unique_ok(sa):
                    # (rows, seg4, rows per scene) + the level's zero-filled output, which the chain's
                    # atomicMax epilogue merges into (zeroed here, off the contraction stream)
                    rel = self._group_rel_xyz_unique(geo["level_xyz"][-1], ctr, gidx, gcnt)
                    if rel is not None:
                        rel = rel + (torch.zeros((B * M, sa["layers"][-1].cout), dtype=torch.float32,
                                                 device=xyz.device),)
                if rel is None:
                    rel = self._group_rel_xyz(geo["level_xyz"][-1], ctr, gidx)
            geo["rel"].append(rel)
            # the contractions of this SA level only need its own sampling + grouping:
            # they may start while the deeper levels' FPS / 3-NN are still running
            geo["sa_events"].append(torch.cuda.current_stream().record_event())
            if dist is not None and all(self.sa[l + 1]["M"] <= self.sa[l]["M"] for l in range(1, len(self.sa) - 1)):
                # (behind the event: the first level's contractions do not wait for the check)
                run = self._fps_prefix_check(ctr, dist, self.sa[1]["M"])
            geo["level_xyz"].append(ctr)
            geo["level_n"].append(M)
            n_cur = M
        sparse_xyz = geo["level_xyz"][-1]
        for fi, fp in enumerate(self.fp):
            dense_xyz = geo["level_xyz"][-2 - fi]
            # grid search with the cell edge chosen on the device from the keys' measured spacing (round 4: the SA
            # level's ball radius, the edge of rounds 2-3, is the right size on surface-like clouds only -- in a
            # uniformly filled box nearly every query missed its 27 cells and took the all-keys scan: 4.8 ms
            # instead of 0.09; the measured edge costs 0.11 / 0.13 ms on either)
            geo["fp"].append(self._three_nn(dense_xyz, sparse_xyz, fp["eps"], -1.0))
            sparse_xyz = dense_xyz
        return geo

    def _dense(self, xyz, geo, topk=None):
        """The shared-MLP contractions of every layer (MFMA).

        f16x2 precision: every launch leaves max|out| PER SCENE in a (B, 64)-slot block of
        `amax` (atomicMax in its epilogue) and the launch that consumes the tensor derives
        the power-of-two activation scale of a scene's rows from that scene's slots -- a
        scene's outputs do not depend on which other scenes share its batch."""
        B, _, N0 = xyz.shape
        dev = xyz.device
        self._batch = B
        level_xyz, level_n = geo["level_xyz"], geo["level_n"]
        keep = geo.get("feats")                      # return_intermediates: every level feature tensor that exists
        n_launch = sum(len(sa["layers"]) for sa in self.sa) + \
            sum(len(fp["layers"]) for fp in self.fp) + len(self.head_layers) + 1 + len(self.sa) + 2 * len(self.fp)
        amax = torch.zeros((n_launch, B, 64), dtype=torch.float32, device=dev)
        rows = iter(amax.unbind(0))
        level_feat = [(None, None)]                  # (tensor, amax row)
        shared = {}       # FP level -> (skip-feature product, amax row) made by the SA launch that read the tensor
        feat = feat_amax = None
        cur = torch.cuda.current_stream()
        for li, sa in enumerate(self.sa):
            M, K = sa["M"], sa["K"]
            _, ctr, gidx, _ = geo["sa"][li]
            cur.wait_event(geo["sa_events"][li])
            P = B * M * K
            layers = sa["layers"]
            x = x_amax = None
            # the last two layers as ONE launch (C -> C -> Cout2 with C = 128 or 256, intermediate
            # in LDS); the first of the pair then reads through the MLP1 or the plain loader
            pre = sa["pre"] if self.precision in ("f16x2", "bf16") else None
            lp = len(layers) - 2                  # first layer of the candidate pair
            first_loader = (LOAD_GATHER_MLP1 if (lp == 1 and sa["mlp1"] is not None) else
                            LOAD_GATHER_ADD if (lp == 1 and pre is not None) else LOAD_PLAIN)
            fuse2 = (K == 64 and len(layers) >= 3 and layers[-1].groups == 1 and
                     self._fusable(layers[-2], layers[-1], first_loader, EPI_MAX))
            if pre is not None:
                # F = W_feat . features, one row per point of the level
                fpre = torch.empty((B * level_n[li], layers[0].cout), dtype=torch.float32, device=dev)
                fpre_amax = next(rows)
                mg = self._shared_input_layer(li, feat.shape[1])
                if mg is not None:
                    fi = len(self.sa) - 1 - li
                    y = torch.empty((B * level_n[li], mg[1].cout), dtype=torch.float32, device=dev)
                    y_amax = next(rows)
                    self._gemm("sa%d.0f|fp%d.0d" % (li, fi), mg[0], B * level_n[li], LOAD_PLAIN, EPI_STORE,
                               relu=False, out=fpre, ldc=layers[0].cout, A=feat, lda=feat.shape[1],
                               a_amax=feat_amax, out_amax=fpre_amax, out2=y, ldc2=mg[1].cout,
                               split_n=layers[0].cout, out_amax2=y_amax)
                    shared[fi] = (y, y_amax)
                else:
                    self._gemm("sa%d.0f" % li, pre["la"], B * level_n[li], LOAD_PLAIN, EPI_STORE, relu=False,
                               out=fpre, ldc=layers[0].cout, A=feat, lda=feat.shape[1], a_amax=feat_amax,
                               out_amax=fpre_amax)
            for l, layer in enumerate(layers):
                if l == 0 and (sa["mlp1"] is not None or pre is not None):
                    continue                      # folded into layer 1's loader
                if fuse2 and l == len(layers) - 1:
                    continue                      # fused behind the previous layer's launch
                last = l == len(layers) - 1 or (fuse2 and l == len(layers) - 2)
                nrow = B * M if last else P
                l2 = layers[-1] if (fuse2 and l == len(layers) - 2) else None
                cout = l2.cout if l2 is not None else layer.cout
                rel = geo["rel"][li] if (l == 1 and sa["mlp1"] is not None) else None
                uniq = isinstance(rel, tuple) and l2 is not None
                out = rel[3] if uniq else torch.empty((nrow, cout), dtype=torch.float32, device=dev)
                out_amax = next(rows)
                kw = dict(out=out, ldc=cout, K=K, out_amax=out_amax, layer2=l2)
                if l == 1 and sa["mlp1"] is not None:
                    kw.update(gidx=gidx, xyz=level_xyz[li], ctr=ctr, N=level_n[li], M=M,
                              mlp1_w=sa["mlp1"], a_amax_floor=sa["mlp1_bound"])
                    if uniq:      # distinct rows only; flops are counted over the rows that exist
                        kw.update(rel_xyz4=rel[0], seg4=rel[1], seg_rows=rel[2], rows_used=rel[2])
                    elif isinstance(rel, tuple):
                        raise RuntimeError("distinct-row records without the fused chain that consumes them")
                    elif rel is not None:
                        kw.update(rel_xyz4=rel)
                    loader = LOAD_GATHER_MLP1
                elif l == 1 and pre is not None:
                    kw.update(gidx=gidx, feat=fpre, Cf=layers[0].cout, xyz=level_xyz[li], ctr=ctr,
                              N=level_n[li], M=M, mlp1_w=pre["w1"], a_amax=fpre_amax,
                              a_amax_floor=pre["bound"])
                    loader = LOAD_GATHER_ADD
                elif l == 0:
                    kw.update(gidx=gidx, feat=feat, xyz=level_xyz[li], ctr=ctr, Cf=sa["cf"],
                              N=level_n[li], M=M, a_amax=feat_amax, a_amax_floor=sa["radius"])
                    loader = LOAD_GATHER
                else:
                    kw.update(A=x, lda=x.shape[1], a_amax=x_amax)
                    loader = LOAD_PLAIN
                self._gemm("sa%d.%d" % (li, l), layer, P, loader, EPI_MAX if last else EPI_STORE,
                           **kw)
                x, x_amax = out, out_amax
            feat, feat_amax = x, x_amax
            level_feat.append((feat, feat_amax))
            if keep is not None:
                keep["sa%d" % li] = feat.view(B, M, -1)

        cur.wait_event(geo["done"])               # the 3-NN searches of the FP path
        (sparse_feat, sparse_amax), n_sparse = level_feat[-1], level_n[-1]
        heads0_fused = False
        heads_pre = None
        carried = None      # (s_out, s_amax) of the NEXT level, produced by this level's last launch
        sparse_channels = 0
        for fi, fp in enumerate(self.fp):
            (dense_feat, dense_amax), n_dense = level_feat[-2 - fi], level_n[-2 - fi]
            nidx, nw = geo["fp"][fi]
            P = B * n_dense
            x = x_amax = None
            fl = fp["layers"]
            pending = None
            # a two-layer level whose successor starts with a linear layer on ITS output alone (no skip
            # features there): interpolate + add in the loader, the level's second layer in LDS, the
            # successor's first layer (no bias, no ReLU) as the chain's final layer -- the level's own
            # output tensor is never written (S4G_FP_CHAIN_NEXT=0: separate launches)
            nxt = self.fp[fi + 1] if fi + 1 < len(self.fp) else None
            chain_next = (self.fp_chain_next and nxt is not None and len(fl) == 2 and self.fp_linear_first and
                          self.precision in ("f16x2", "bf16") and level_feat[-3 - fi][0] is None and
                          fl[1].cin == fl[1].cout and fl[1].kpad16 == fl[1].cout and fl[1].Wfrag is not None and
                          nxt["layers"][0].cin == fl[1].cout and nxt["layers"][0].cout % 64 == 0 and
                          nxt["layers"][0].cout <= 1024 and len(nxt["layers"]) >= 2 and
                          bool(_cabi.lib().s4g_gemm_chain_supported(LOAD_INTERP_ADD, EPI_STORE, fl[1].cout,
                                                                     fl[1].kpad16)))
            fuse2 = len(fl) >= 3 and self._fusable(fl[-2], fl[-1])
            # last FP level: the first head layer (shared input, groups == 1) rides along as a
            # third layer, so the per-point features never go through HBM before the heads
            h0 = self.head_layers[0]
            fuse3 = (fuse2 and fi == len(self.fp) - 1 and self.fuse3 and self.heads_fused is None and
                     fl[-1].cout == fl[-2].cout and
                     h0.groups == 1 and h0.cin == fl[-1].cout and h0.kpad16 == fl[-1].cout and
                     h0.cout % 64 == 0 and h0.Wfrag is not None)
            for l, layer in enumerate(fl):
                if fuse2 and l == len(fl) - 1:
                    continue                      # fused behind the previous layer's launch
                if pending is not None and self._heads_take_tail(fi, fl, pending):
                    # the level's last two layers run in front of the heads, inside their launch
                    heads_pre = dict(pending, layers=[fl[-2], fl[-1]], name="fp%d.1+fp%d.2" % (fi, fi))
                    pending = None
                    x = x_amax = None
                    break
                if pending is not None and chain_next and l == 1:
                    # this level's second layer + the next level's linear first layer: one chain launch
                    nl = nxt["layers"][0]
                    nsp = self._fp_split(nxt, nl, layer.cout, 0)
                    s_next = torch.empty((P, nl.cout), dtype=torch.float32, device=dev)
                    s_next_amax = next(rows)
                    self._gemm("fp%d.%d" % (fi, l), layer, P, LOAD_INTERP_ADD, EPI_STORE, out=s_next,
                               ldc=nl.cout, out_amax=s_next_amax, layer2=nsp[0], relu2=False,
                               name2="fp%d.0s" % (fi + 1), **pending)
                    carried = (s_next, s_next_amax)
                    pending = None
                    x = x_amax = None             # the level's own output is never materialised
                    sparse_channels = layer.cout
                    break
                l2 = fl[-1] if (fuse2 and l == len(fl) - 2) else None
                l3 = h0 if (fuse3 and l2 is not None) else None
                out = torch.empty((P, (l3 or l2 or layer).cout), dtype=torch.float32, device=dev)
                out_amax = next(rows)
                if l == 0 and self.fp_linear_first and layer.cout % 4 == 0 and layer.cout <= 1024:
                    # the layer is linear: apply it to the sparse features (and to the skip
                    # features) first, interpolate the narrow result afterwards -- the big
                    # tensor's contraction shrinks from K = C2 + C1 to K = C1 (or vanishes)
                    c2 = sparse_channels if sparse_feat is None else sparse_feat.shape[1]
                    c1 = 0 if dense_feat is None else dense_feat.shape[1]
                    la, lb, _ = self._fp_split(fp, layer, c2, c1)
                    s_out = None if carried is not None else \
                        torch.empty((B * n_sparse, layer.cout), dtype=torch.float32, device=dev)
                    # the sum is formed by the NEXT launch's loader where that is faster
                    # (S4G_FP_LOADER_ADD = list of FP levels), else by interp_add_cl_kernel
                    if self.fp_loader_add == "auto":
                        in_loader = chain_next or (fuse2 and len(fl) == 3 and
                                                   self._fusable(fl[-2], fl[-1], LOAD_INTERP_ADD, EPI_STORE))
                    else:
                        in_loader = self.fp_loader_add != "none" and fi in self.fp_loader_add
                    in_loader = in_loader and self.precision in ("f16x2", "bf16") and len(fl) >= 2
                    if in_loader and fuse2 and not self._fusable(fl[-2], fl[-1], LOAD_INTERP_ADD, EPI_STORE):
                        in_loader = False
                    if carried is not None:      # the previous level's chain already produced it
                        s_out, s_amax = carried
                        carried = None
                    else:
                        s_amax = next(rows) if in_loader else None
                        self._gemm("fp%d.0s" % fi, la, B * n_sparse, LOAD_PLAIN, EPI_STORE, relu=False,
                                   out=s_out, ldc=layer.cout, A=sparse_feat, lda=c2, a_amax=sparse_amax,
                                   out_amax=s_amax)
                    y = y_amax = None
                    if lb is not None and fi in shared:
                        y, y_amax = shared.pop(fi)   # made by sa{li}.0f's launch (same input tensor)
                    elif lb is not None:
                        y = torch.empty((P, layer.cout), dtype=torch.float32, device=dev)
                        y_amax = next(rows) if in_loader else None
                        self._gemm("fp%d.0d" % fi, lb, P, LOAD_PLAIN, EPI_STORE, relu=False, out=y,
                                   ldc=layer.cout, A=dense_feat, lda=c1, a_amax=dense_amax,
                                   out_amax=y_amax)
                    if in_loader:
                        pending = dict(sparse=s_out, dense=y, C2=layer.cout, N2=n_sparse, N1=n_dense,
                                       nidx=nidx, nw=nw, loader_bias=layer.bias, a_amax=s_amax,
                                       a_amax2=y_amax, a_amax_floor=fp["bias0_max"])
                        continue
                    with _F._timed("interp_add[P=%d,C=%d]" % (P, layer.cout),
                                   P * layer.cout * (8 if y is not None else 4) + P * 24):
                        rc = _cabi.lib().s4g_interp_add_cl_f32(
                            None if y is None else y.data_ptr(), s_out.data_ptr(), nidx.data_ptr(),
                            nw.data_ptr(), layer.bias.data_ptr(), B, n_dense, n_sparse, layer.cout, 1,
                            out.data_ptr(), out_amax.data_ptr(), _F._stream())
                    _cabi.check(rc, "interp_add_cl")
                elif l == 0:
                    c1 = 0 if dense_feat is None else dense_feat.shape[1]
                    self._gemm("fp%d.%d" % (fi, l), layer, P, LOAD_INTERP, EPI_STORE, out=out,
                               ldc=layer.cout, nidx=nidx, nw=nw, sparse=sparse_feat,
                               dense=dense_feat, C2=sparse_feat.shape[1], C1=c1, N2=n_sparse,
                               N1=n_dense, a_amax=sparse_amax, a_amax2=dense_amax,
                               out_amax=out_amax)
                elif pending is not None:
                    self._gemm("fp%d.%d" % (fi, l), layer, P, LOAD_INTERP_ADD, EPI_STORE, out=out,
                               ldc=out.shape[1], out_amax=out_amax, layer2=l2, layer3=l3, **pending)
                    heads0_fused = l3 is not None
                    pending = None
                else:
                    self._gemm("fp%d.%d" % (fi, l), layer, P, LOAD_PLAIN, EPI_STORE, out=out,
                               ldc=out.shape[1], A=x, lda=x.shape[1], a_amax=x_amax,
                               out_amax=out_amax, layer2=l2, layer3=l3)
                    heads0_fused = l3 is not None
                x, x_amax = out, out_amax
            sparse_feat, sparse_amax, n_sparse = x, x_amax, n_dense
            if keep is not None and x is not None and not heads0_fused:      # (a level folded into its consumer's launch has no tensor)
                keep["fp%d" % fi] = x.view(B, n_dense, -1)

        # heads
        P = B * N0
        x, x_amax = sparse_feat, sparse_amax
        names = ("score", "frame_R", "frame_t", "movable_logits")
        if topk is not None:
            if self.heads_fused is None or heads_pre is None:
                raise NotImplementedError("topk= needs the one-launch heads with the FP tail in front (the shipped widths)")
            return self._heads_topk(heads_pre, B, N0, int(topk), dev)
        if self.heads_fused is not None and (heads_pre is not None or x.shape[1] == 256):
            packed, outs = self._head_outputs(B, N0, dev)
            self._heads(x, x_amax, outs, B, N0, pre=heads_pre)
            return PackedPred(zip(names, outs), packed=packed)
        l0 = self.head_layers[0]
        if not heads0_fused:
            h = torch.empty((P, l0.cout), dtype=torch.float32, device=dev)
            h_amax = next(rows)
            self._gemm("heads.0", l0, P, LOAD_PLAIN, EPI_STORE, out=h, ldc=l0.cout, A=x,
                       lda=x.shape[1], a_amax=x_amax, out_amax=h_amax)
            x, x_amax = h, h_amax
        hl = self.head_layers
        l = 1
        while l < len(hl):
            layer = hl[l]
            # two or three consecutive grouped layers as one launch where the widths allow it
            l2 = hl[l + 1] if (l + 1 < len(hl) and self._fusable(layer, hl[l + 1])) else None
            l3 = None
            if (l2 is not None and self.fuse3 and l + 2 < len(hl) and l2.cout == layer.cout and
                    self._fusable(l2, hl[l + 2])):
                l3 = hl[l + 2]
            if l2 is not None and l3 is None and layer.kpad16 != layer.cout:
                l2 = None                # a deep first layer only as part of a three-layer chain
            cout = (l3 or l2 or layer).cout
            h = torch.empty((P, layer.groups * cout), dtype=torch.float32, device=dev)
            h_amax = next(rows)
            self._gemm("heads.%d" % l, layer, P, LOAD_PLAIN, EPI_STORE, out=h,
                       ldc=layer.groups * cout, c_gcol=cout, A=x, lda=x.shape[1],
                       a_gcol=layer.cin, a_amax=x_amax, out_amax=h_amax, layer2=l2, layer3=l3,
                       name3="heads.%d" % (l + 2))
            x, x_amax = h, h_amax
            l += 3 if l3 is not None else (2 if l2 is not None else 1)
        names = ("score", "frame_R", "frame_t", "movable_logits")
        packed, outs = self._head_outputs(B, N0, dev)
        starts = [0]
        for c in self.head_channels:
            starts.append(starts[-1] + c)
        if packed is not None:
            # the channel-first epilogue addresses head h as base_h + (b ch_h + c) N: the packed tensor is ONE
            # head of sum(ch) channels to it (the sigmoid boundary is a channel number either way)
            cf_ptr = (_fp * 4)(*[packed.data_ptr()] * 4)
            cf_start = (_i32 * 5)(0, *[starts[-1]] * 4)
        else:
            cf_ptr = (_fp * 4)(*[o.data_ptr() for o in outs])
            cf_start = (_i32 * 5)(*starts)
        self._gemm("heads.logits", self.logit_layer, P, LOAD_PLAIN, EPI_CF, relu=False, A=x,
                   lda=x.shape[1], cf_ptr=cf_ptr, cf_start=cf_start,
                   cf_sigmoid_from=self.sigmoid_from, cf_N=N0, a_amax=x_amax)
        return PackedPred(zip(names, outs), packed=packed)

    def _heads_topk(self, pre, B, N0, K, dev):
        """The heads for a serving path that only decodes a scene's best-scoring points (row f1): the score head
        (+ the FP tail in front of it) on EVERY point, then the rotation / translation / movable heads on the K points
        per scene with the largest expected score -- the heads are 46 % of a step and three quarters of them is work
        on points nobody decodes.  Same kernel both times (`s4g_heads_desc_t.head_mask`, ABI 11); the second launch
        re-forms its panel from the gathered interpolation indices / weights / skip rows of the kept points, so nothing
        per-point is stored in between.  Returns a PackedPred over the kept points: score / frame_R / frame_t /
        movable_logits as (B, c, K) slices of one packed tensor, plus "index" (B, K) int64, best first.  The kept
        points' values are the full forward's up to the hidden layers' per-tile scales (f16x2: 1e-7 of the output
        scale; tests/test_sparse_heads_gpu.py)."""
        K = min(K, N0)
        score = torch.empty((B, self.head_channels[0], N0), dtype=torch.float32, device=dev)
        self._heads(None, None, [score, None, None, None], B, N0, pre=pre, head_mask=1)
        # the expected score over the classes decides -- the decode's own kernel and class values, so that its top-K of
        # the kept points is its top-K of the scene (the detector's convention differs by a constant: same order)
        from . import postprocess as _pp
        es = _pp.expected_score(score, "demo")                                     # (B, N0)
        sel = torch.topk(es, K, dim=1, largest=True, sorted=True)[1]               # (B, K), best first
        flat = (sel + torch.arange(B, device=dev).view(B, 1) * N0).reshape(-1)     # rows of the (B N0, .) tensors
        sub = dict(pre, nidx=pre["nidx"].reshape(B * N0, 3).index_select(0, flat).contiguous(),
                   nw=pre["nw"].reshape(B * N0, 3).index_select(0, flat).contiguous(),
                   dense=None if pre["dense"] is None else pre["dense"].index_select(0, flat).contiguous())
        packed = torch.empty((B, sum(self.head_channels), K), dtype=torch.float32, device=dev)
        outs = list(packed.split(self.head_channels, dim=1))
        outs[0].copy_(torch.gather(score, 2, sel.unsqueeze(1).expand(-1, score.shape[1], -1)))
        self._heads(None, None, outs, B, K, pre=sub, head_mask=0b1110)
        pred = PackedPred(zip(("score", "frame_R", "frame_t", "movable_logits"), outs), packed=packed)
        pred["index"] = sel
        return pred

    def _head_outputs(self, B, N0, dev):
        """(packed (B, sum c_h, N) tensor or None, the four (B, c_h, N) head tensors -- its channel slices)."""
        if not self.packed_out:
            return None, [torch.empty((B, c, N0), dtype=torch.float32, device=dev) for c in self.head_channels]
        packed = torch.empty((B, sum(self.head_channels), N0), dtype=torch.float32, device=dev)
        return packed, list(packed.split(self.head_channels, dim=1))

    @torch.no_grad()
    def submit(self, data_batch, topk=None, keep_feats=False):
        """Enqueue one forward pass and return a `Handle` without waiting.

        The coordinate-only work (FPS pyramid, ball queries, 3-NN) runs on a
        high-priority geometry stream, the contractions on a dense stream that
        waits for it.  FPS is a latency chain on one CU per scene; submitting
        batch i+1 before collecting batch i lets that chain run underneath the
        previous batch's contractions instead of in front of its own."""
        if self.fold_only:
            raise RuntimeError("FusedPointNet2(fold_only=True) holds packed weights only: it cannot run a forward")
        xyz = _F._f32c(data_batch["scene_points"], "scene_points")
        if xyz.dim() != 3 or xyz.size(1) != 3:
            raise RuntimeError("scene_points must be (B, 3, N)")
        dev = xyz.device
        if self.check_finite:
            bad = (~torch.isfinite(xyz)).flatten(1).any(dim=1).nonzero().flatten().tolist()
            if bad:
                raise ValueError("scene_points: non-finite coordinates in scene(s) %s of the batch" % bad)
        _F.OpTimer.begin_pass()
        if self._streams is None or self._streams[0][0].device != dev:
            # high-priority geometry streams: FPS is a latency chain on one CU per
            # scene, so S4G_GEO_STREAMS (default 2) consecutive batches may run
            # their chains side by side (needs >= 2 batches in flight to matter);
            # S4G_DENSE_STREAMS=2 lets consecutive batches' contractions fill each
            # other's kernel tails (+1.5 % measured), the default of 1 keeps
            # per-kernel event timings clean
            self._streams = _stream_pool(dev, self.geo_streams, self.dense_streams)
            self._submitted = 0
        gs = self._streams[0][self._submitted % len(self._streams[0])]
        ds = self._streams[1][self._submitted % len(self._streams[1])]
        self._submitted += 1
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            ev_in = cur.record_event()
            gs.wait_event(ev_in)
            with torch.cuda.stream(gs):
                geo = self._geometry(xyz)
                geo["done"] = gs.record_event()
                if keep_feats:
                    geo["feats"] = {}
            ds.wait_event(ev_in)
            with torch.cuda.stream(ds):
                pred = self._dense(xyz, geo, topk=topk)
                ev_out = ds.record_event()
            # tensors cross streams: tell the caching allocator
            xyz.record_stream(gs)
            xyz.record_stream(ds)
            for idx, ctr, gidx, gcnt in geo["sa"]:
                for t in (idx, ctr, gidx, gcnt):
                    t.record_stream(ds)
            for t in geo["rel"]:
                for u in (t if isinstance(t, tuple) else (t,)):
                    if u is not None:
                        u.record_stream(ds)
            for nidx, nw in geo["fp"]:
                nidx.record_stream(ds)
                nw.record_stream(ds)
        return Handle(pred, geo, ev_out)

    def graph(self, example_batch):
        """Record one forward pass of `example_batch`'s shape as a HIP graph: see `GraphedForward`."""
        return GraphedForward(self, example_batch)

    def __call__(self, data_batch, return_intermediates=False, topk=None):
        h = self.submit(data_batch, topk=topk, keep_feats=return_intermediates)
        pred = h.result()
        if return_intermediates:
            inter = {}
            h.event.synchronize()
            for name, t in h.geo["feats"].items():      # (B, n, C) rows -> the reference's (B, C, n)
                inter["feat_" + name] = t.transpose(1, 2)
            for li, (idx, ctr, gidx, gcnt) in enumerate(h.geo["sa"]):
                inter["fps%d" % li], inter["ball%d" % li], inter["cnt%d" % li] = idx, gidx, gcnt
            for fi, (nidx, nw) in enumerate(h.geo["fp"]):
                inter["nn%d" % fi], inter["nnw%d" % fi] = nidx, nw
            for li, rel in enumerate(h.geo["rel"]):
                if isinstance(rel, tuple):      # distinct-row form: rows each scene occupies (M K = plain layout)
                    inter["sa%d_rows" % li] = rel[2]
            return pred, inter
        return pred


class PackedPred(dict):
    """The forward's output dict (`PointNet2_tcls.py:142-147`: score / frame_R / frame_t / movable_logits).  On the
    fast path the four tensors are channel slices of `packed`, ONE (B, 21, N) tensor in the head order of
    `dist.HEADS` that the heads launch wrote directly -- `dist.pack_outputs` hands it to the all-gather as it
    is.  `packed` is None when the tensors are separate allocations (S4G_PACKED_OUT=0)."""

    def __init__(self, items=(), packed=None):
        super().__init__(items)
        self.packed = packed


class GraphedForward:
    """One forward pass of a fixed batch shape recorded as a HIP graph (`FusedPointNet2.graph`).

    The forward is sync-free and allocation-stable (every launch takes its stream, nothing reads a device
    value on the host), so the ~45 launches of a pass -- geometry on its own streams, contractions behind
    their events -- are captured once and replayed with ONE host call: a pass then costs the host ~20 us
    instead of ~1 ms of Python, and the launch gaps of the one-scene latency go away.  Inputs are copied
    into the graph's static buffer; the returned tensors are the graph's static outputs (valid until the
    next replay -- clone what must survive)."""

    def __init__(self, model, example):
        xyz = _F._f32c(example["scene_points"], "scene_points")
        self.static_in = xyz.clone()
        self.model = model
        dev = xyz.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up off the capture: lazy kernel attributes, allocator pools
            for _ in range(2):
                model({"scene_points": self.static_in})
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        was = _F.OpTimer.enabled
        _F.OpTimer.enabled = False             # event pairs with timing cannot be captured
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self.static_out = model({"scene_points": self.static_in})
        finally:
            _F.OpTimer.enabled = was

    def __call__(self, data_batch):
        xyz = data_batch["scene_points"]
        if tuple(xyz.shape) != tuple(self.static_in.shape):
            raise RuntimeError("graph was recorded for scene_points of shape %s" % (tuple(self.static_in.shape),))
        if xyz.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(xyz, non_blocking=True)
        self.graph.replay()
        return self.static_out


class Handle:
    """An in-flight forward pass (see `FusedPointNet2.submit`)."""

    def __init__(self, pred, geo, event):
        self.pred, self.geo, self.event = pred, geo, event

    def result(self):
        """Make the caller's current stream wait for the outputs and return them."""
        cur = torch.cuda.current_stream(next(iter(self.pred.values())).device)
        cur.wait_event(self.event)
        for t in self.pred.values():
            t.record_stream(cur)
        for li, (idx, ctr, gidx, gcnt) in enumerate(self.geo["sa"]):
            for t in (idx, ctr, gidx, gcnt):
                t.record_stream(cur)
        for nidx, nw in self.geo["fp"]:
            nidx.record_stream(cur)
            nw.record_stream(cur)
        return self.pred

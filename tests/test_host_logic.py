"""CPU tests of the host-side logic of the fast path (no kernels involved):
BatchNorm folding, the exact 3-way bf16 split, K permutation / padding of the
packed weights, and the GEMM descriptor's ABI layout."""
import ctypes

import numpy as np
import torch

from s4g_release_amd import _cabi
from s4g_release_amd.fused import _pad_k, fold_conv_bn, split_bf16x3, split_f16x2
from s4g_release_amd.model import PointNet2, randomize_bn_
from s4g_release_amd.nn_utils import Conv1d, Conv2d


def test_fold_conv_bn_equals_conv_then_bn_eval():
    torch.manual_seed(0)
    for blk, shape in ((Conv1d(7, 5, 1), (2, 7, 11)), (Conv2d(4, 6, 1), (2, 4, 3, 5))):
        blk.bn.weight.data.uniform_(0.5, 1.5)
        blk.bn.bias.data.normal_()
        blk.bn.running_mean.normal_()
        blk.bn.running_var.uniform_(0.5, 2.0)
        blk.eval()
        x = torch.randn(*shape)
        ref = blk(x.clone())
        w, b = fold_conv_bn(blk)
        y = torch.einsum("oc,bc...->bo...", w, x) + b.view(1, -1, *([1] * (x.dim() - 2)))
        assert torch.allclose(torch.relu(y), ref, atol=1e-5)


def test_split_f16x2_scales_and_precision():
    """Per-row power-of-two scale into [2^14, 2^15); two fp16 planes carry >= 22
    significand bits of the scaled value; zero rows stay zero."""
    g = torch.Generator().manual_seed(4)
    w = torch.randn(2, 48, 64, generator=g) * torch.logspace(-6, 3, 48)[None, :, None]
    w[1, 5] = 0.0
    planes, inv = split_f16x2(w)
    assert planes.dtype == torch.float16 and planes.shape == (2, 2, 48, 64) and inv.shape == (2, 48)
    m, e = torch.frexp(inv)
    assert torch.all(m == 0.5)                                    # exact powers of two
    scaled = w.double() / inv.double()[..., None]
    amax = scaled.abs().amax(dim=-1)
    nz = amax > 0
    assert torch.all(amax[nz] >= 2.0 ** 14) and torch.all(amax[nz] < 2.0 ** 15)
    rec = planes[0].double() + planes[1].double()
    err = (rec - scaled).abs()
    # second plane rounds at 2^-11 of the first plane's half-ulp (2^-11 |x|); below
    # 2^-14 the second plane is subnormal with a 2^-25 absolute grid
    assert torch.all(err <= torch.maximum(scaled.abs() * 2.0 ** -22, torch.tensor(2.0 ** -25, dtype=torch.float64)))
    assert torch.all(planes[:, 1, 5] == 0)


def test_split_bf16x3_is_exact():
    g = torch.Generator().manual_seed(1)
    w = torch.cat([torch.randn(4000, generator=g), torch.randn(4000, generator=g) * 1e-6,
                   torch.randn(4000, generator=g) * 1e6, torch.tensor([0.0, 1.0, -1.0, 3.1415927])])
    planes = split_bf16x3(w)
    assert planes.dtype == torch.bfloat16 and planes.shape == (3, w.numel())
    back = planes[0].double() + planes[1].double() + planes[2].double()
    assert torch.equal(back, w.double())            # 8 + 8 + 8 significand bits: nothing lost
    assert (planes[1].float().abs() <= planes[0].float().abs() * 2.0 ** -7 + 1e-45).all()


def test_pad_k_and_sa_weight_permutation():
    w = torch.arange(2 * 11, dtype=torch.float32).view(2, 11)
    p = _pad_k(w)
    assert p.shape == (2, 16) and torch.equal(p[:, :11], w) and (p[:, 11:] == 0).all()
    # SA first layers: reference K order [xyz(3), feat(C)] -> ours [feat(C), xyz(3)]
    perm = torch.cat([w[:, 3:], w[:, :3]], dim=1)
    assert torch.equal(perm[:, -3:], w[:, :3]) and torch.equal(perm[:, :8], w[:, 3:])


def test_gemm_desc_matches_the_c_struct_layout():
    d = _cabi.GemmDesc
    # natural alignment as the C compiler lays out s4g_gemm_desc_t (include/s4g_ops.h)
    assert d.W.offset == 32 and d.bias.offset == 40 and d.A.offset == 56
    assert d.gidx.offset % 8 == 0 and d.cf_ptr.offset % 8 == 0
    assert d.W_bf16x3.offset % 8 == 0 and d.mlp1_w.offset == d.W_bf16x3.offset + 8
    assert ctypes.sizeof(d) % 8 == 0


def test_heads_desc_matches_the_c_struct_layout(tmp_path):
    """ctypes mirror of s4g_heads_desc_t against the C compiler's layout of include/s4g_ops.h
    (sizes and the offsets of the ABI-6 members)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "s4g_ops.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu", sizeof(s4g_heads_desc_t), '
                   'offsetof(s4g_heads_desc_t, pre_W_frag), offsetof(s4g_heads_desc_t, pre_nidx), '
                   'offsetof(s4g_heads_desc_t, pre_N2), sizeof(s4g_gemm_desc_t));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    c = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    d = _cabi.HeadsDesc
    assert c == [ctypes.sizeof(d), d.pre_W_frag.offset, d.pre_nidx.offset, d.pre_N2.offset,
                 ctypes.sizeof(_cabi.GemmDesc)]


def test_randomize_bn_is_deterministic_and_nontrivial():
    cfg = dict(score_classes=3, num_centroids=(8, 4, 2), radius=(0.1, 0.2, 0.4),
               num_neighbours=(16, 16, 16), sa_channels=((8, 8), (8, 8), (8, 8)),
               fp_channels=((8, 8), (8, 8), (8, 8)), num_fp_neighbours=(3, 3, 3),
               seg_channels=(8, 8), num_removal_directions=5, dropout_prob=0.5)
    torch.manual_seed(3)
    a = randomize_bn_(PointNet2(**cfg), 4).state_dict()
    torch.manual_seed(3)
    b = randomize_bn_(PointNet2(**cfg), 4).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    k = "sa_modules.0.mlp.0.bn.running_var"
    assert not torch.allclose(a[k], torch.ones_like(a[k]))


def test_detector_oracle_literal_restatement_and_its_index_mixup():
    """oracle/postprocess.py restates GraspDetector.post_processing twice: `literal` follows the
    reference as written (grasp_detector.py:150-154: positions inside `high_score_index` used as
    point indices for frame_R, and a numpy `.transpose(0, 1)` that is the identity, so the (9, n)
    array is reshaped row-major into n blocks) -- pinned against the reference's own output in
    tests/test_post_golden.py; the default pairs every pose with its own point.  The two coincide
    only in the degenerate case of ONE survivor that is point 0, and differ otherwise."""
    import numpy as np
    from oracle import postprocess as OP
    rng = np.random.default_rng(2)
    N = 400
    pred = {"score": np.zeros((3, N), np.float32), "frame_R": rng.standard_normal((9, N)).astype(np.float32),
            "frame_t": rng.standard_normal((4, N)).astype(np.float32)}
    pts = rng.random((3, N)).astype(np.float32)
    pred["score"][0] = 6.0
    pred["score"][:, 0] = (0.0, 0.0, 9.0)              # point 0 alone clears the threshold
    a = OP.detector_post_processing(pred, pts, 0.7, -2.0, np.eye(3))
    b = OP.detector_post_processing(pred, pts, 0.7, -2.0, np.eye(3), literal=True)
    assert len(a[2]) == 1 and np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    pred["score"][0] = 0.0
    pred["score"][2] = rng.standard_normal(N) * 4      # generic scores: the literal result differs
    a = OP.detector_post_processing(pred, pts, 0.7, 0.1, np.eye(3))
    b = OP.detector_post_processing(pred, pts, 0.7, 0.1, np.eye(3), literal=True)
    assert len(a[2]) > 10 and len(b[2]) > 10
    assert (np.diff(a[1]) <= 0).all() and (np.diff(a[2]) < 0).any()        # best score first
    assert (np.diff(b[2]) > 0).all() and not (np.diff(b[1]) <= 0).all()    # ascending point index
    # block m of the literal mode: flat elements 9 m .. 9 m + 8 of frame_R[:, index_high2low]
    sc = OP.expected_scores(pred["score"])
    high = np.nonzero(sc > 0.7)[0]
    h2l = np.argsort(sc[high])[::-1]
    flat = pred["frame_R"][:, h2l].ravel()
    m = int(np.nonzero(high == b[2][0])[0][0])
    x = flat[9 * m:9 * m + 9].reshape(3, 3)[:, 0].astype(np.float64)
    assert np.allclose(b[0][0][:3, 0], (OP.TRAIN2REAL[:3, :3] @ (x / np.linalg.norm(x))), atol=1e-6)
    # frames are orthonormal and carry the caller's frame change
    R = a[0][:, :3, :3]
    assert np.allclose(np.einsum("nij,nik->njk", R, R), np.eye(3), atol=1e-5)
    assert np.allclose(np.abs(np.linalg.det(R)), 1.0, atol=1e-5)


def test_small_matrix_cache_is_bounded_and_skips_nan():
    """postprocess._small_on_device: a caller's direction_matrix changes per capture on a moving camera -- the by-value
    cache must stay bounded (LRU of 16), must not keep NaN-valued entries (NaN != NaN: a miss and a leak per call), and
    must hand back the SAME device tensor for a repeated value (that is the point: no per-call host -> device copy)."""
    from s4g_release_amd import postprocess as pp
    cpu = torch.device("cpu")
    pp._SMALL_LRU.clear()
    a = pp._small_on_device(((1., 0., 0.), (0., 1., 0.), (0., 0., 1.)), torch.float32, cpu)
    assert pp._small_on_device(np.eye(3), torch.float32, cpu) is a and len(pp._SMALL_LRU) == 1
    for i in range(100):                                          # a moving camera: 100 distinct poses
        m = pp._small_on_device(np.eye(3) * (1.0 + i), torch.float32, cpu)
        assert float(m[0, 0]) == 1.0 + i and m.dtype == torch.float32
    assert len(pp._SMALL_LRU) == pp._SMALL_LRU_MAX == 16
    assert pp._small_on_device(np.eye(3), torch.float32, cpu) is not a      # evicted long ago, rebuilt
    n = len(pp._SMALL_LRU)
    for _ in range(5):
        bad = pp._small_on_device((float("nan"), 0., 1.), torch.float32, cpu)
        assert bad.shape == (3,) and bool(torch.isnan(bad[0]))
    assert len(pp._SMALL_LRU) == n                                # NaN values never enter the cache
    last = pp._small_on_device(np.eye(3) * 100.0, torch.float32, cpu)
    assert pp._small_on_device(np.eye(3) * 100.0, torch.float64, cpu) is not last    # dtype is part of the key

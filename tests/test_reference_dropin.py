"""The drop-in levels of INTEGRATION.md against the REFERENCE'S OWN classes (in-container only: the GPU box has no
/root/reference, so these tests skip there; nothing of the reference is stored in the repo).

 * level 1: `s4g_release_amd.pn2_ext` registered under the name the reference imports
   (`pointnet2_utils/functions.py:2`: `from . import pn2_ext`) -- the reference's `functions.py`, `modules.py`
   and `PointNet2_tcls.py` import unchanged and their operator calls land in our seven entry points
   (`csrc/main.cpp:7-13`), which refuse CPU tensors exactly like the reference's CHECK_CUDA;
 * level 3: the reference's own `PointNet2_tcls.PointNet2` INSTANCE handed to `FusedPointNet2` folds and packs to
   the very tensors `model.PointNet2` with the same `state_dict` gives (`PointNet2_tcls.py:56-95`) -- the fast
   path accepts the maintainer's object, not only the repo's mirror of it."""
import os
import sys

import pytest
import torch

REF = "/root/reference/inference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")

FULL = dict(score_classes=3, num_centroids=(5120, 1024, 256), radius=(0.02, 0.08, 0.32), num_neighbours=(64, 64, 64),
            sa_channels=((128, 128, 256), (256, 256, 512), (512, 512, 1024)),
            fp_channels=((1024, 1024), (512, 512), (256, 256, 256)), num_fp_neighbours=(3, 3, 3),
            seg_channels=(512, 256, 256, 128), num_removal_directions=5, dropout_prob=0.5)
SMALL = dict(score_classes=3, num_centroids=(512, 128, 32), radius=(0.05, 0.12, 0.4), num_neighbours=(16, 16, 16),
             sa_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128)), fp_channels=((128, 128), (64, 64), (32, 32, 32)),
             num_fp_neighbours=(3, 3, 3), seg_channels=(64, 32, 32, 16), num_removal_directions=5, dropout_prob=0.5)
EXT = "grasp_proposal.network_models.models.pointnet2_utils.pn2_ext"


@pytest.fixture(scope="module")
def ref():
    """The reference's modules imported over OUR extension shim (the one-line binding of INTEGRATION.md level 1)."""
    from s4g_release_amd import pn2_ext as ours
    saved = {k: v for k, v in sys.modules.items() if k.startswith("grasp_proposal")}
    for k in saved:
        del sys.modules[k]
    sys.modules[EXT] = ours
    sys.path.insert(0, REF)
    try:
        from grasp_proposal.network_models.models import PointNet2_tcls
        from grasp_proposal.network_models.models.pointnet2_utils import functions as ref_F
        yield PointNet2_tcls, ref_F, ours
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k.startswith("grasp_proposal")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_reference_modules_import_over_our_extension_and_call_into_it(ref):
    PointNet2_tcls, ref_F, ours = ref
    assert ref_F.pn2_ext is ours
    for name in ("farthest_point_sample", "ball_query", "group_points_forward", "group_points_backward",
                 "point_search", "interpolate_forward", "interpolate_backward"):       # main.cpp:7-13
        assert callable(getattr(ours, name)), name
    # the reference's operator wrapper reaches our entry point, which refuses a CPU tensor (CHECK_CUDA)
    with pytest.raises(RuntimeError, match="(?i)cuda|hip|device"):
        ref_F.farthest_point_sample(torch.zeros(1, 3, 8), 4)
    net = PointNet2_tcls.PointNet2(**SMALL).eval()
    with pytest.raises(RuntimeError, match="(?i)cuda|hip|device"):
        net({"scene_points": torch.zeros(1, 3, 2048)})


@pytest.mark.parametrize("cfg", [FULL, SMALL], ids=["shipped", "small"])
@pytest.mark.parametrize("precision", ["f16x2", "bf16"])
def test_fast_path_accepts_the_reference_instance_and_packs_identical_weights(ref, cfg, precision):
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import PointNet2, randomize_bn_
    PointNet2_tcls = ref[0]
    torch.manual_seed(77)
    ref_net = PointNet2_tcls.PointNet2(**cfg)
    randomize_bn_(ref_net, 78)
    ref_net.eval()
    ours = PointNet2(**cfg).eval()
    missing = ours.load_state_dict(ref_net.state_dict(), strict=True)        # same 200 keys, same shapes
    assert not missing.missing_keys and not missing.unexpected_keys
    a = FusedPointNet2(ref_net, precision=precision, fold_only=True)         # the REFERENCE'S object
    b = FusedPointNet2(ours, precision=precision, fold_only=True)
    pa, pb = a.packed_weights(), b.packed_weights()
    assert set(pa) == set(pb) and len(pa) > 60
    for k in pa:
        assert pa[k].dtype == pb[k].dtype and torch.equal(pa[k], pb[k]), k
    assert a.head_channels == b.head_channels == [3, 9, 4, 5]
    assert (a.heads_fused is None) == (b.heads_fused is None)
    with pytest.raises(RuntimeError, match="fold_only"):
        a({"scene_points": torch.zeros(1, 3, 64)})


def test_training_mode_reference_instance_is_refused(ref):
    from s4g_release_amd.fused import FusedPointNet2
    net = ref[0].PointNet2(**SMALL)          # nn.Module default: training mode
    with pytest.raises(RuntimeError, match="eval"):
        FusedPointNet2(net, fold_only=True)

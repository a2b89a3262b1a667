"""float64 restatement of the network's arithmetic on fp32 geometry -- TEST INFRASTRUCTURE ONLY.

Same composition as oracle/pn2_forward.py (PointNet2_tcls.py:99-148; SA modules.py:208-244, FP :102-129,498-507,
conv -> eval BN -> ReLU nn_utils/conv.py:28-34), but every floating-point tensor behind the index operators is
float64: the yardstick that says how far an fp32 forward (torch CPU, or the device's split-fp16 contraction) is
from the exact result.  FPS / ball query / 3-NN run on the fp32 coordinates through the C oracle, so the indices
are the reference's."""
import numpy as np
import torch

from oracle import oracle as O


def _mlp(x, sd, prefix):
    i = 0
    while "%s.%d.conv.weight" % (prefix, i) in sd:
        w = sd["%s.%d.conv.weight" % (prefix, i)].double().flatten(1)
        x = w @ x
        g, b = sd["%s.%d.bn.weight" % (prefix, i)].double(), sd["%s.%d.bn.bias" % (prefix, i)].double()
        m, v = sd["%s.%d.bn.running_mean" % (prefix, i)].double(), sd["%s.%d.bn.running_var" % (prefix, i)].double()
        x = (x - m[:, None]) / torch.sqrt(v[:, None] + 1e-5) * g[:, None] + b[:, None]
        x = x.clamp_min(0)
        i += 1
    return x


def forward64(state_dict, points, num_centroids, radius, num_neighbours):
    """points (1, 3, N) float32 numpy -> dict of float64 numpy outputs (1, C, N)."""
    assert points.shape[0] == 1
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    xyz = np.ascontiguousarray(points, dtype=np.float32)
    feat = None                                    # (C, n) float64 torch
    lv_xyz, lv_feat = [xyz], [None]
    with torch.no_grad():
        for li, (M, r, K) in enumerate(zip(num_centroids, radius, num_neighbours)):
            idx = O.fps(xyz, M)
            ctr = O.gather_points(xyz, idx)
            gidx, _ = O.ball_query(xyz, ctr, r, K)
            j = torch.from_numpy(gidx[0].reshape(-1).astype(np.int64))
            gx = torch.from_numpy(xyz[0]).double()[:, j].reshape(3, M, K) - torch.from_numpy(ctr[0]).double()[:, :, None]
            # (group_xyz -= new_xyz is ONE fp32 subtraction in the reference: round it like that)
            gx = (torch.from_numpy(xyz[0])[:, j].reshape(3, M, K) - torch.from_numpy(ctr[0])[:, :, None]).double()
            g = gx if feat is None else torch.cat([gx, feat[:, j].reshape(-1, M, K)], dim=0)
            y = _mlp(g.reshape(g.shape[0], M * K), sd, "sa_modules.%d.mlp" % li)
            feat = y.reshape(-1, M, K).max(dim=2)[0]
            xyz = ctr
            lv_xyz.append(xyz)
            lv_feat.append(feat)
        sparse_xyz, sparse = xyz, feat
        for fi in range(len(num_centroids)):
            dense_xyz, dense = lv_xyz[-2 - fi], lv_feat[-2 - fi]
            nidx, d2 = O.three_nn(dense_xyz, sparse_xyz)
            w = torch.from_numpy(O.interp_weights(d2, 1e-10)[0]).double()          # (n, 3): fp32 weights, as computed
            ni = torch.from_numpy(nidx[0].astype(np.int64))
            interp = (sparse[:, ni[:, 0]] * w[:, 0] + sparse[:, ni[:, 1]] * w[:, 1]) + sparse[:, ni[:, 2]] * w[:, 2]
            x = interp if dense is None else torch.cat([interp, dense], dim=0)
            sparse = _mlp(x, sd, "fp_modules.%d.mlp" % fi)
            sparse_xyz = dense_xyz
        out = {}
        for name, mlp, logit in (("score", "mlp_seg", "seg_logit"), ("frame_R", "mlp_R", "R_logit"),
                                 ("frame_t", "mlp_t", "t_logit"), ("movable_logits", "mlp_movable", "movable_logit.0")):
            h = _mlp(sparse, sd, mlp)
            o = sd[logit + ".weight"].double().flatten(1) @ h + sd[logit + ".bias"].double()[:, None]
            if name == "movable_logits":
                o = torch.sigmoid(o)
            out[name] = o.numpy()[None]
    return out

"""CPU restatement of the pose decode that follows the network in the reference's
demo (`utils/file_logger_cls.py:34-47,66-68,196-218`).  TEST INFRASTRUCTURE ONLY.

Parity status: unpinned -- the reference function writes files and needs open3d
(not importable here); this follows its arithmetic line by line (torch softmax
in fp32, numpy float64 afterwards), without the collision filter.
"""
import numpy as np
import torch
import torch.nn.functional as F


def decode_top_poses(pred, scene_points, K=50, convention="demo"):
    """pred: dict of numpy (B,C,N); scene_points (B,3,N).  Returns H (B,K,4,4) f64,
    score (B,K) f64, index (B,K)."""
    Hs, Ss, Is = [], [], []
    for b in range(scene_points.shape[0]):
        pts = scene_points[b].T                                              # (N,3)  :27
        logits = F.softmax(torch.from_numpy(pred["score"][b]), dim=0).numpy().T      # :35
        R = pred["frame_R"][b].T.reshape(-1, 3, 3)                           # :38-41
        t = F.softmax(torch.from_numpy(pred["frame_t"][b]), dim=0).numpy().T         # :43
        t_score = np.array([0.08, 0.06, 0.04, 0.02])[np.newaxis, :t.shape[1]]        # :45
        frame_t = -(t * t_score).sum(1, keepdims=True) * R[:, :, 0] + pts            # :46
        C = logits.shape[1]
        vals = np.linspace(0, 1, C + 1)
        vals = (vals[:-1] if convention == "demo" else vals[1:])[np.newaxis, :]      # :67
        scene_pred = np.sum(vals * logits, axis=1)                                   # :68
        top = np.argsort(-scene_pred, kind="stable")[:K]                             # :197
        H = np.tile(np.eye(4), (len(top), 1, 1))
        for i, ind in enumerate(top):                                                # :203-216
            Rm = R[ind].astype(np.float64)
            x = Rm[:, 0] / np.linalg.norm(Rm[:, 0])
            y = Rm[:, 1] - np.sum(x * Rm[:, 1]) * x
            y = y / np.linalg.norm(y)
            z = np.cross(x, y)
            H[i, :3, :3] = np.stack([x, y, z], axis=1)
            H[i, :3, 3] = frame_t[ind]
        Hs.append(H); Ss.append(scene_pred[top]); Is.append(top)
    return np.stack(Hs), np.stack(Ss), np.stack(Is)


def view_non_collision(poses, scene_points, half_bottom_width=0.057, bottom_length=0.16,
                       finger_width=0.023, half_hand_thickness=0.012, finger_length=0.09,
                       back_margin=0.0, back_threshold=10 * np.sqrt(8), finger_threshold=10):
    """Restatement of CloudCollisionChecker.view_non_collision
    (cloud_processor/view_collision_checker.py:37-65) for (B,K,4,4) poses; returns
    (ok (B,K) bool, counts (B,K,2))."""
    B, K = poses.shape[:2]
    ok = np.zeros((B, K), dtype=bool)
    counts = np.zeros((B, K, 2), dtype=np.int64)
    hbs = half_bottom_width - finger_width
    for b in range(B):
        homo = np.concatenate([scene_points[b], np.ones((1, scene_points.shape[2]), np.float32)], 0)
        for k in range(K):
            g2l = np.linalg.inv(poses[b, k].astype(np.float64)).astype(np.float32)
            local = torch.matmul(torch.from_numpy(g2l), torch.from_numpy(homo)).numpy()      # :38
            close = (local[0] < finger_length) & (local[0] > -bottom_length)                  # :39-40
            lc = local[:, close][0:3]                                                         # :42
            zc = (lc[2] < half_hand_thickness) & (lc[2] > -half_hand_thickness)               # :44-45
            back = (lc[1] < half_bottom_width) & (lc[1] > -half_bottom_width) & \
                   (lc[0] < -back_margin) & zc                                                # :47-49
            left = (lc[1] < half_bottom_width) & (lc[1] > hbs)                                # :54-55
            right = (lc[1] > -half_bottom_width) & (lc[1] < -hbs)                             # :56-57
            fing = zc & (left | right)                                                        # :59-60
            counts[b, k] = (back.sum(), fing.sum())
            ok[b, k] = not (back.sum() > back_threshold) and not (fing.sum() > finger_threshold)
    return ok, counts

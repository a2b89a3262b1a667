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

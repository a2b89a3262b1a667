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


REAL2TRAIN = np.array([[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64)   # grasp_detector.py:26
TRAIN2REAL = np.linalg.inv(REAL2TRAIN)                                                                 # :27


def _orthogonalization(rot, trans):
    """GraspDetector.orthogonalization, grasp_detector.py:124-135."""
    x = rot[:, :, 0]
    x = x / np.linalg.norm(x, axis=1, keepdims=True)
    y = rot[:, :, 1]
    y = y - np.sum(x * y, axis=1, keepdims=True) * x
    y = y / np.linalg.norm(y, axis=1, keepdims=True)
    z = np.cross(x, y)
    mat44 = np.tile(np.eye(4), [rot.shape[0], 1, 1])
    mat44[:, :3, :3] = np.stack([x, y, z], axis=2)
    mat44[:, :3, 3] = trans
    return mat44


def detector_post_processing(pred, points, score_threshold, vertical_degree_threshold,
                             direction_matrix, vertical_direction=(0.0, 0.0, 1.0), frame=TRAIN2REAL,
                             literal=False):
    """GraspDetector.post_processing (grasp_detector.py:137-185) for ONE scene: pred values are
    numpy (C, N), points (3, N).  direction_matrix = camera2base[:3,:3] @ TRAIN2REAL[:3,:3] (:155).
    Returns (mat44 (n,4,4) f64 in `frame`, scores (n,), point index (n,)).

    literal=False (what the product implements): survivors of the score threshold in DESCENDING
    score order (`argsort(...)[::-1]`, :151), each with ITS OWN rotation / translation / position,
    filtered by the verticalness test.
    literal=True: the reference's indexing exactly as written -- `index_high2low` (positions inside
    `high_score_index`) is used as a POINT index for `frame_R` (:154) and `index_good_direction`
    (positions in that permuted list) as positions inside `high_score_index` (:160), so row i pairs
    the rotation of point index_high2low[j] with position / score / translation of point
    high_score_index[j] (a defect of the reference: SURVEY Appendix D style, not reproduced by the
    product; restated here so the difference is testable)."""
    all_scores = F.softmax(torch.from_numpy(pred["score"]), dim=0).numpy()                    # :143
    C = all_scores.shape[0]
    score_value = np.linspace(0, 1, C + 1)[1:][:, np.newaxis]                                 # :145
    all_scores = np.sum(score_value * all_scores, axis=0)                                     # :146
    high = np.nonzero(all_scores > score_threshold)[0]                                        # :149
    high2low = np.argsort(all_scores[high])[::-1]                                             # :150
    vdir = np.asarray(vertical_direction, dtype=np.float32)[np.newaxis, :]                    # :80
    dm = np.asarray(direction_matrix, dtype=np.float64)
    pts = points.T if points.shape[0] == 3 else points                                        # :161-162
    t_score = np.array([0.08, 0.06, 0.04, 0.02])[np.newaxis, :pred["frame_t"].shape[0]]       # :177
    if literal:
        rotation = pred["frame_R"][:, high2low].transpose(1, 0).reshape([-1, 3, 3])           # :153-154
        x_direction = -dm @ rotation[:, :, 0].T                                               # :155
        vertical_degree = np.sum(x_direction.T * vdir, axis=1)                                # :156
        good = np.nonzero(vertical_degree > vertical_degree_threshold)[0]                     # :157
        valid = high[good]                                                                    # :160
        rotation = rotation[good]                                                             # :164
    else:
        order = high[high2low]                                  # point indices, best score first
        rotation = pred["frame_R"][:, order].transpose(1, 0).reshape([-1, 3, 3])
        x_direction = -dm @ rotation[:, :, 0].T
        vertical_degree = np.sum(x_direction.T * vdir, axis=1)
        good = np.nonzero(vertical_degree > vertical_degree_threshold)[0]
        valid = order[good]
        rotation = rotation[good]
    p = pts[valid, :]                                                                         # :163
    translation = F.softmax(torch.from_numpy(pred["frame_t"][:, valid]), dim=0).numpy().T     # :165-166
    scores = all_scores[valid]                                                                # :167
    gt = -(translation * t_score).sum(1, keepdims=True) * rotation[:, :, 0] + p              # :178
    mat44 = _orthogonalization(rotation.astype(np.float64), gt)                               # :179
    mat44 = np.matmul(np.asarray(frame, dtype=np.float64)[np.newaxis], mat44)                 # :180
    return mat44, scores, valid


def importance_sampling(scores, random_numbers):
    """grasp_detector.py:237-251 with the uniform draws passed in (the reference calls
    np.random.rand(num_selected) unseeded): indices of the selected poses."""
    scores_cum = np.cumsum(np.exp(5 * scores))                                                # :239
    random_score = np.sort(np.asarray(random_numbers)) * scores_cum[-1]                       # :240
    out, index = [], 0
    for target in random_score:                                                               # :243-247
        while scores_cum[index] < target:
            index += 1
        out.append(index)
    return np.array(out)


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

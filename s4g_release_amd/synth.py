"""Synthetic single-view table-top clouds (`tabletop-v1`) for benchmarks and tests.

There is no dataset in the image and the reference ships one scene
(`inference/2638_view_0.p`, (3, 48902) fp32; x in [-0.38,0.24], y in
[-0.35,0.35], z in [-1.47,-0.80]), so benchmark inputs are generated.  The
generator is counter based (splitmix64 of (seed, scene, point, lane)), i.e.
independent of numpy's RNG streams and reproducible across versions.

Scene: 40 % of the points on a tilted table patch, 60 % on the surfaces of 8
boxes / spheres / cylinders (0.04-0.15 m) standing on it, +-0.5 mm jitter, and
a final seeded permutation so that index order carries no spatial locality --
as in the reference after `np.random.choice`
(`inference/grasp_proposal/grasp_proposal_test.py:26-29`).

Variants (stress cases of SURVEY.md section 8d):
  `uniform-box`  uniform in the scene box: ~1 neighbour at r = 0.02, every
                 ball padded, no early exit;
  `dup-heavy`    10 000 unique tabletop points sampled with replacement to N:
                 exact distance ties (mirrors `replace=True` at
                 grasp_proposal_test.py:29).
"""
import numpy as np

DEFAULT_SEED = 20260101
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _hash(seed, scene, pid, lane):
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed))
        h = _splitmix64(h ^ np.uint64(scene))
        h = _splitmix64(h ^ np.asarray(pid, dtype=np.uint64))
        h = _splitmix64(h ^ np.uint64(lane))
    return h


def _u01(seed, scene, pid, lane):
    """Uniform [0,1) with a 24-bit mantissa, as float64 for exact scaling."""
    return (_hash(seed, scene, pid, lane) >> np.uint64(40)).astype(np.float64) / float(1 << 24)


_OBJ_BASE = 1 << 40  # point ids reserved for per-scene object parameters


def _tabletop_points(seed, scene, n):
    pid = np.arange(n, dtype=np.uint64)
    u = [_u01(seed, scene, pid, lane) for lane in range(6)]
    # per-scene parameters
    pu = _u01(seed, scene, np.arange(_OBJ_BASE, _OBJ_BASE + 80, dtype=np.uint64), 0)
    tilt_x = (pu[0] - 0.5) * 0.30
    tilt_y = (pu[1] - 0.5) * 0.30
    z_table = -1.35 + (pu[2] - 0.5) * 0.10
    pts = np.empty((n, 3), dtype=np.float64)
    on_table = u[0] < 0.40
    # table patch (0.62 x 0.60 m, inside the real scene's x/y extent)
    tx = -0.38 + u[1] * 0.62
    ty = -0.30 + u[2] * 0.60
    pts[:, 0] = tx
    pts[:, 1] = ty
    pts[:, 2] = z_table + tilt_x * tx + tilt_y * ty
    # objects
    obj = np.minimum((u[1] * 8).astype(np.int64), 7)
    for k in range(8):
        sel = (~on_table) & (obj == k)
        if not sel.any():
            continue
        q = pu[8 + k * 8: 16 + k * 8]
        kind = int(q[0] * 3)
        ox = -0.30 + q[1] * 0.46
        oy = -0.22 + q[2] * 0.44
        sx = 0.04 + q[3] * 0.11
        sy = 0.04 + q[4] * 0.11
        sz = 0.04 + q[5] * 0.11
        base = z_table + tilt_x * ox + tilt_y * oy
        a, b, c = u[2][sel], u[3][sel], u[4][sel]
        if kind == 0:  # box: 5 visible faces
            face = np.minimum((c * 5).astype(np.int64), 4)
            px = (a - 0.5) * sx
            py = (b - 0.5) * sy
            pz = np.full_like(a, sz)
            m1 = face == 1
            px = np.where(m1, -0.5 * sx, px); py = np.where(m1, (a - 0.5) * sy, py); pz = np.where(m1, b * sz, pz)
            m2 = face == 2
            px = np.where(m2, 0.5 * sx, px); py = np.where(m2, (a - 0.5) * sy, py); pz = np.where(m2, b * sz, pz)
            m3 = face == 3
            px = np.where(m3, (a - 0.5) * sx, px); py = np.where(m3, -0.5 * sy, py); pz = np.where(m3, b * sz, pz)
            m4 = face == 4
            px = np.where(m4, (a - 0.5) * sx, px); py = np.where(m4, 0.5 * sy, py); pz = np.where(m4, b * sz, pz)
        elif kind == 1:  # sphere of radius sx/2 resting on the table
            r = 0.5 * sx
            ct = 2.0 * a - 1.0
            st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
            ph = 2.0 * np.pi * b
            px, py, pz = r * st * np.cos(ph), r * st * np.sin(ph), r + r * ct
        else:  # cylinder: side (80 %) + top cap
            r = 0.5 * sx
            ph = 2.0 * np.pi * a
            side = c < 0.8
            rr = np.where(side, r, r * np.sqrt(b))
            px, py = rr * np.cos(ph), rr * np.sin(ph)
            pz = np.where(side, b * sz, sz)
        pts[sel, 0] = ox + px
        pts[sel, 1] = oy + py
        pts[sel, 2] = base + pz
    # +-0.5 mm jitter
    for d in range(3):
        pts[:, d] += (_u01(seed, scene, pid, 8 + d) - 0.5) * 1e-3
    return pts


def _permute(seed, scene, pts):
    n = pts.shape[0]
    key = _hash(seed, scene, np.arange(n, dtype=np.uint64), 31)
    order = np.argsort(key, kind="stable")
    return pts[order]


def make_scene(scene_id, num_points=25600, seed=DEFAULT_SEED, variant="tabletop-v1"):
    """Return one cloud as (3, N) float32, channel-first like the reference input."""
    if variant == "tabletop-v1":
        pts = _permute(seed, scene_id, _tabletop_points(seed, scene_id, num_points))
    elif variant == "uniform-box":
        pid = np.arange(num_points, dtype=np.uint64)
        lo = np.array([-0.40, -0.35, -1.47])
        hi = np.array([0.40, 0.35, -0.80])
        pts = np.stack([lo[d] + _u01(seed, scene_id, pid, 40 + d) * (hi[d] - lo[d])
                        for d in range(3)], axis=1)
    elif variant == "dup-heavy":
        uniq = _permute(seed, scene_id, _tabletop_points(seed, scene_id, 10000))
        pick = (_u01(seed, scene_id, np.arange(num_points, dtype=np.uint64), 50) * 10000)
        pts = uniq[np.minimum(pick.astype(np.int64), 9999)]
    elif variant == "lattice":
        # the table-top scene snapped to a 2^-8 m (3.9 mm) lattice: exact distance ties everywhere, so FPS
        # over the level-1 centroids is NOT their prefix in general (the tie rule decides) -- the scenes
        # that make the conditional level-2 / level-3 samplers run
        pts = _permute(seed, scene_id, _tabletop_points(seed, scene_id, num_points))
        pts = np.round(pts * 256.0) / 256.0
    else:
        raise ValueError("unknown variant %r" % (variant,))
    return np.ascontiguousarray(pts.T.astype(np.float32))


def make_batch(scene_ids, num_points=25600, seed=DEFAULT_SEED, variant="tabletop-v1"):
    """(B, 3, N) float32 for the given scene ids."""
    return np.stack([make_scene(s, num_points, seed, variant) for s in scene_ids], axis=0)

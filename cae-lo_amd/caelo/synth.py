"""Seeded synthetic KITTI-shaped LiDAR scans (no KITTI data exists in this environment).

SURVEY.md section 8(d): a fixed scene (ground plane + axis-aligned boxes + thin poles) is
ray-cast by a 64-beam (+2.0 .. -24.8 deg) x 2000-azimuth spinning sensor that moves
~0.9 m / frame with a small yaw, sigma = 1 cm range noise, U(0,1) intensity.

The generator is written so that the produced float32 cloud is bit-identical under
NumPy 1.26 (the conda python3.9 used to import the reference for the goldens) and
NumPy 2.x (tests / bench): only scalar libm calls through ``math`` and element-wise
IEEE +,-,*,/ and comparisons are used (no vectorised transcendental functions, no
reductions whose order could differ), and the legacy ``RandomState`` streams.
``cloud_sha256`` lets a test verify that a regenerated cloud equals the one the goldens
were made from.
"""
import hashlib
import math

import numpy as np

SCENE_SEED = 1234
GROUND_Z = -1.73
MAX_RANGE = 80.0


def _scene(seed=SCENE_SEED):
    rs = np.random.RandomState(seed)
    boxes = []
    nbox = 60
    for _ in range(nbox):
        # centre on a ring 6..60 m away, footprint 1.5..12 m, height 1..9 m
        ang = rs.uniform(0.0, 2.0 * math.pi)
        dist = rs.uniform(6.0, 60.0)
        sx = rs.uniform(1.5, 12.0)
        sy = rs.uniform(1.5, 12.0)
        h = rs.uniform(1.0, 9.0)
        cx = dist * math.cos(ang)
        cy = dist * math.sin(ang)
        # keep every box at least 4 m from the sensor track (x in [-5, 40], |y| < 3)
        if (cx - sx / 2 < 45.0 and cx + sx / 2 > -6.0) and abs(cy) - sy / 2 < 4.0:
            continue
        boxes.append((cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, GROUND_Z, GROUND_Z + h))
    for _ in range(40):
        ang = rs.uniform(0.0, 2.0 * math.pi)
        dist = rs.uniform(5.0, 45.0)
        h = rs.uniform(2.0, 7.0)
        cx = dist * math.cos(ang)
        cy = dist * math.sin(ang)
        if (cx < 45.0 and cx > -6.0) and abs(cy) < 4.0:
            continue
        boxes.append((cx - 0.15, cx + 0.15, cy - 0.15, cy + 0.15, GROUND_Z, GROUND_Z + h))
    return np.array(boxes, dtype=np.float64)


def _clutter(seed):
    """The "clutter" scene (round 3): vegetation instead of architecture.  No wall, no box: ~10 000 small spheres -- tree crowns as
    balls of 45 leaves-clumps (radius 10 .. 30 cm) that the rays enter to different depths, bushes as clusters of eight, trunks
    as thin boxes -- so that a key point's neighbourhood is an irregular VOLUME of returns (dense 16 cm / 64 cm patches, the
    496-nearest cut of Voxel.py:182 biting on real key points, hardly two equal patches), where the box scene gives planes, edges
    and many duplicate patches.  Returns (boxes [n,6], spheres [m,4] = centre + radius)."""
    rs = np.random.RandomState(seed + 77)
    spheres, boxes = [], []

    def clear_of_track(cx, cy, r):
        return not ((cx - r < 45.0 and cx + r > -6.0) and abs(cy) - r < 2.5)

    for _ in range(64):                     # trees: a trunk and a crown
        ang, dist = rs.uniform(0.0, 2.0 * math.pi), rs.uniform(5.0, 40.0)
        cx, cy = dist * math.cos(ang), dist * math.sin(ang)
        h = rs.uniform(2.2, 5.5)
        offs = rs.normal(0.0, 0.6, size=(45, 3))
        rad = rs.uniform(0.10, 0.30, size=45)
        if not clear_of_track(cx, cy, 2.0):
            continue
        boxes.append((cx - 0.12, cx + 0.12, cy - 0.12, cy + 0.12, GROUND_Z, GROUND_Z + h))
        for o, r in zip(offs, rad):
            spheres.append((cx + o[0], cy + o[1], GROUND_Z + h + 0.7 * o[2], r))
    for _ in range(220):                    # bushes
        ang, dist = rs.uniform(0.0, 2.0 * math.pi), rs.uniform(4.0, 36.0)
        cx, cy = dist * math.cos(ang), dist * math.sin(ang)
        offs = rs.normal(0.0, 0.25, size=(8, 3))
        rad = rs.uniform(0.10, 0.25, size=8)
        if not clear_of_track(cx, cy, 1.0):
            continue
        for o, r in zip(offs, rad):
            spheres.append((cx + o[0], cy + o[1], GROUND_Z + 0.25 + abs(o[2]), r))
    for _ in range(260):                    # a belt of thicket 30 .. 48 m out: it hides the far ground, whose sparse rings would
        ang, dist = rs.uniform(0.0, 2.0 * math.pi), rs.uniform(30.0, 48.0)   # otherwise collect most of the key points
        cx, cy = dist * math.cos(ang), dist * math.sin(ang)
        offs = rs.normal(0.0, 1.0, size=(26, 3))
        rad = rs.uniform(0.15, 0.45, size=26)
        for o, r in zip(offs, rad):
            spheres.append((cx + 0.9 * o[0], cy + 0.9 * o[1], GROUND_Z + 1.4 + 1.1 * o[2], r))
    return np.array(boxes, dtype=np.float64).reshape(-1, 6), np.array(spheres, dtype=np.float64).reshape(-1, 4)


_SCENE_CACHE = {}
_CLUTTER_CACHE = {}


def clutter(seed=SCENE_SEED):
    if seed not in _CLUTTER_CACHE:
        _CLUTTER_CACHE[seed] = _clutter(seed)
    return _CLUTTER_CACHE[seed]


def scene(seed=SCENE_SEED):
    if seed not in _SCENE_CACHE:
        _SCENE_CACHE[seed] = _scene(seed)
    return _SCENE_CACHE[seed]


TRAJECTORIES = ("line", "circuit")
TILE = 54.0            # metres: the period of the "circuit" world along x = 60 frames of 0.9 m
TILE_X0 = -7.0         # a tile holds the scene's objects whose centre has x in [TILE_X0, TILE_X0 + TILE)
CIRCUIT_PERIOD = 600   # frames: lcm(60 frames per tile, 200 per lateral weave, 150 per yaw swing) -- frame f and f + 600 see the same world


def sensor_pose(frame, step=(0.9, 0.05, 0.0), yaw_step=0.01, trajectory="line"):
    """World pose of the sensor at frame index ``frame`` (translation, yaw).

    "line" (rounds 1-5; every golden of tests/golden was made with it): a straight line with a steady lateral drift and yaw --
    the sensor LEAVES the ~100 m scene: from frame ~150 on a scan holds a few hundred non-ground returns, from ~200 on none.
    "circuit" (round 6): 0.9 m per frame along x through a world that repeats every TILE metres (see ``tiled``), a +-1.5 m
    lateral weave and a +-0.2 rad yaw swing -- KITTI-like frame-to-frame motion (0.9 m, <= 0.047 m sideways, <= 0.0084 rad)
    with structure on both sides at EVERY frame index, and closed: the world seen from frame f + 600 is the one seen from f."""
    if trajectory == "line":
        return (step[0] * frame, step[1] * frame, step[2] * frame), yaw_step * frame
    assert trajectory == "circuit", trajectory
    return ((0.9 * frame, 1.5 * math.sin(2.0 * math.pi * frame / 200.0 + 0.7), 0.0),
            0.2 * math.sin(2.0 * math.pi * frame / 150.0 + 0.3))


_TILE_CACHE = {}


def tiled(scene_kind, seed=SCENE_SEED):
    """One period of the "circuit" world: the objects of ``scene(seed)`` / ``clutter(seed)`` whose centre has x in
    [TILE_X0, TILE_X0 + TILE) and that stay >= 4 m (boxes, poles, trunks) / >= 2.5 m (spheres) clear of the track's axis y = 0.
    The world is this set repeated at every multiple of TILE along x.  Returns (boxes [n,6], spheres [m,4])."""
    key = (scene_kind, seed)
    if key not in _TILE_CACHE:
        if scene_kind == "clutter":
            solids, balls = clutter(seed)
        else:
            solids, balls = scene(seed), np.zeros((0, 4))
        cx = 0.5 * (solids[:, 0] + solids[:, 1])
        clear = np.minimum(np.abs(solids[:, 2]), np.abs(solids[:, 3]))
        straddles = (solids[:, 2] < 0.0) & (solids[:, 3] > 0.0)
        keep = (cx >= TILE_X0) & (cx < TILE_X0 + TILE) & (clear >= 4.0) & ~straddles
        solids = solids[keep]
        if len(balls):
            keepb = (balls[:, 0] >= TILE_X0) & (balls[:, 0] < TILE_X0 + TILE) & (np.abs(balls[:, 1]) - balls[:, 3] >= 2.5)
            balls = balls[keepb]
        _TILE_CACHE[key] = (solids, balls)
    return _TILE_CACHE[key]


def _circuit_world(scene_kind, seed, tx):
    """The tiles of the periodic world within reach (80 m range + the largest half extent) of a sensor at x = ``tx``, in
    coordinates RELATIVE to the sensor's own tile (so that float64 magnitudes stay small for any frame index): returns
    (local tx, boxes, spheres)."""
    solids, balls = tiled(scene_kind, seed)
    k0 = math.floor(tx / TILE)
    local = tx - k0 * TILE
    bs, ss = [], []
    for k in range(-2, 3):
        off = k * TILE
        b = solids.copy(); b[:, 0] += off; b[:, 1] += off
        bs.append(b[(b[:, 0] - local < MAX_RANGE) & (local - b[:, 1] < MAX_RANGE)])      # (out of range: cannot return a point)
        if len(balls):
            s_ = balls.copy(); s_[:, 0] += off
            ss.append(s_[np.abs(s_[:, 0] - local) - s_[:, 3] < MAX_RANGE])
    return local, np.concatenate(bs), (np.concatenate(ss) if ss else np.zeros((0, 4)))


def make_scan(frame=0, n_beams=64, n_az=2000, seed=None, scene_seed=SCENE_SEED,
              pose=None, noise_sigma=0.01, quantum=None, scene_kind="boxes", trajectory="line"):
    """Return a [N,4] float32 cloud (x,y,z,intensity) in the sensor frame, file order
    beam-major (all azimuths of beam 0, then beam 1, ...).  ``quantum`` (metres, e.g. 1e-3): coordinates rounded to
    multiples of it, like the metrically quantised values real scanners deliver -- such clouds put points exactly on
    voxel faces (x = 4.0), the case the reference's float64 index arithmetic (Voxel.py:118-152) resolves in its own
    way and a voxelization has to reproduce.  ``scene_kind``: "boxes" (ground, boxes, poles: the scene of rounds 1-2) or
    "clutter" (ground, trees and bushes made of spheres).  ``trajectory``: see ``sensor_pose``; "circuit" ray-casts the periodic
    world of ``tiled``."""
    if seed is None:
        seed = frame
    (tx, ty, tz), yaw = sensor_pose(frame, trajectory=trajectory) if pose is None else pose
    world = None
    if trajectory == "circuit":
        assert scene_kind in ("boxes", "clutter")
        tx, *world = _circuit_world(scene_kind, scene_seed, tx)
    else:
        assert trajectory == "line", trajectory
    cy_, sy_ = math.cos(yaw), math.sin(yaw)
    # direction table in the sensor frame via scalar libm
    elev = [math.radians(2.0 + (-24.8 - 2.0) * i / (n_beams - 1)) for i in range(n_beams)]
    az = [2.0 * math.pi * (j + 0.5) / n_az - math.pi for j in range(n_az)]
    ce = np.array([math.cos(e) for e in elev])
    se = np.array([math.sin(e) for e in elev])
    ca = np.array([math.cos(a) for a in az])
    sa = np.array([math.sin(a) for a in az])
    dx = (ce[:, None] * ca[None, :]).reshape(-1)
    dy = (ce[:, None] * sa[None, :]).reshape(-1)
    dz = (se[:, None] * np.ones(n_az)[None, :]).reshape(-1)
    # rotate directions into the world frame (yaw about z)
    wx = cy_ * dx - sy_ * dy
    wy = sy_ * dx + cy_ * dy
    wz = dz
    t = np.full(dx.shape, np.inf)
    # ground plane z = GROUND_Z
    down = wz < 0.0
    tg = np.where(down, (GROUND_Z - tz) / np.where(down, wz, -1.0), np.inf)
    t = np.minimum(t, tg)
    big = 1e30
    balls = np.zeros((0, 4))
    if world is not None:
        solids, balls = world
    elif scene_kind == "clutter":
        solids, balls = clutter(scene_seed)
    else:
        assert scene_kind == "boxes"
        solids = scene(scene_seed)
    if len(balls):
        # ray / sphere with element-wise IEEE +, -, *, sqrt only (see the module docstring).  A sphere is only tested against the
        # azimuth columns that can see it (a conservative slice of the [beam, azimuth] grid: skipping a ray that cannot hit changes
        # nothing), which makes thousands of spheres affordable.
        t2 = t.reshape(n_beams, n_az)
        wx2, wy2, wz2 = wx.reshape(n_beams, n_az), wy.reshape(n_beams, n_az), wz.reshape(n_beams, n_az)
        for cxs, cys, czs, rad in balls:
            ox, oy, oz = tx - cxs, ty - cys, tz - czs
            dxy = math.hypot(ox, oy)
            if dxy > rad * 1.05:
                half = math.asin(min(1.0, rad * 1.05 / dxy)) + 2.5 * (2.0 * math.pi / n_az)
                rel = math.atan2(-oy, -ox) - yaw                   # azimuth of the centre in the sensor frame
                j0 = int(math.floor((rel - half + math.pi) / (2.0 * math.pi) * n_az - 0.5))
                j1 = int(math.ceil((rel + half + math.pi) / (2.0 * math.pi) * n_az - 0.5))
                cols = np.arange(j0, j1 + 1) % n_az
            else:
                cols = np.arange(n_az)
            ax, ay, az_ = wx2[:, cols], wy2[:, cols], wz2[:, cols]
            bq = ox * ax + oy * ay + oz * az_
            cq = ox * ox + oy * oy + oz * oz - rad * rad
            disc = bq * bq - cq
            ok = disc > 0.0
            th = -bq - np.sqrt(np.where(ok, disc, 0.0))
            hit = ok & (th > 0.5)
            t2[:, cols] = np.where(hit, np.minimum(t2[:, cols], th), t2[:, cols])
        t = t2.reshape(-1)
    for b in solids:
        x0, x1, y0, y1, z0, z1 = b
        with np.errstate(divide="ignore", invalid="ignore"):
            ix = 1.0 / wx
            iy = 1.0 / wy
            iz = 1.0 / wz
            ta = (x0 - tx) * ix
            tb = (x1 - tx) * ix
            tmin = np.minimum(ta, tb)
            tmax = np.maximum(ta, tb)
            ta = (y0 - ty) * iy
            tb = (y1 - ty) * iy
            tmin = np.maximum(tmin, np.minimum(ta, tb))
            tmax = np.minimum(tmax, np.maximum(ta, tb))
            ta = (z0 - tz) * iz
            tb = (z1 - tz) * iz
            tmin = np.maximum(tmin, np.minimum(ta, tb))
            tmax = np.minimum(tmax, np.maximum(ta, tb))
        hit = (tmax >= tmin) & (tmin > 0.5) & (tmin < big)
        t = np.where(hit, np.minimum(t, tmin), t)
    rs = np.random.RandomState(seed)
    noise = rs.normal(0.0, 1.0, size=t.shape) * noise_sigma
    inten = rs.uniform(0.0, 1.0, size=t.shape)
    keep = t < MAX_RANGE
    r = t + noise
    pc = np.empty((int(keep.sum()), 4), dtype=np.float32)
    xyz = (r * dx, r * dy, r * dz)
    if quantum:
        xyz = tuple(np.rint(v / quantum) * quantum for v in xyz)   # element-wise IEEE ops only (see module docstring)
    pc[:, 0] = xyz[0][keep]
    pc[:, 1] = xyz[1][keep]
    pc[:, 2] = xyz[2][keep]
    pc[:, 3] = inten[keep]
    return pc


def cloud_sha256(pc):
    return hashlib.sha256(np.ascontiguousarray(pc).tobytes()).hexdigest()


def relative_pose_gt(frame0, frame1, trajectory="line"):
    """Ground-truth (R, T) with P0 ~ R @ P1 + T for clouds of frame0 / frame1."""
    (t0, y0), (t1, y1) = sensor_pose(frame0, trajectory=trajectory), sensor_pose(frame1, trajectory=trajectory)

    def rot(a):
        c, s = math.cos(a), math.sin(a)
        return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])

    R0, R1 = rot(y0), rot(y1)
    R = R0.T @ R1
    T = R0.T @ (np.array(t1) - np.array(t0))
    return R, T.reshape(3, 1)


def shuffle_scan(pc, seed, dup_fraction=0.01):
    """The same scan in a hostile FILE ORDER: ``dup_fraction`` of the points repeated with another intensity (same x, y, z),
    then everything randomly permuted.  Two rules of the reference depend on the order of the points: the LAST point of a ring
    pixel wins it (SphericalRing.py:91-93: intensity and range of the pixel) and the FIRST point of a 2 cm voxel decides which
    16 cm / 64 cm voxels it marks (Voxel.py:139-158).  A beam-major scan never exercises either across distant file
    positions; this one does."""
    rng = np.random.RandomState(seed)
    n = pc.shape[0]
    dup = pc[rng.randint(0, n, int(round(n * dup_fraction)))].copy()
    dup[:, 3] = rng.random_sample(len(dup)).astype(np.float32)
    out = np.concatenate([pc, dup], axis=0)
    return np.ascontiguousarray(out[rng.permutation(len(out))])

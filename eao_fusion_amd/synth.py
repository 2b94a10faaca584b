"""Deterministic synthetic workloads.  There is no dataset in the reference tree
(data/rgb_full_demo.txt is an index of file names only), so every benchmark/parity input is generated here.

frames  : config 2 -- 640x480 u8 "rectangle world" images, seed 1000+f
hamming : config 3 -- 1000x32-byte descriptor sets, seed 2000 (+ planted variant, seed 2001)
ba      : config 4 -- 24 cameras (4 fixed + 20 free) x 3000 points, stereo observations, seed 3000
pose    : PoseOptimization problem (one frame, N matched map points)
"""
import numpy as np

# reference camera (ros_test/config/TUM3.yaml:8-11,25)
FX, FY, CX, CY, BF = 535.4, 539.2, 320.1, 247.6, 40.0


def synth_frame(seed, w=640, h=480, n_rect=400, n_small=1000):
    """"Rectangle world": background 96; n_rect large (20..w/2 x 20..h/2) then n_small small (3..23 px)
    axis-aligned rectangles with uniform origins and intensity U[0,255] painted in order (clipped at the
    image edge); 3x3 box blur; additive integer noise U[-3,3]; clamp.  Returns (h, w) uint8.
    Defaults give ~3000 FAST corners at threshold 20 on level 0 (every per-level quota over-subscribed,
    so the quad-tree distribution is exercised); n_rect=40, n_small=0 is the low-texture variant that
    drives cells into the minThFAST retry."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 96, dtype=np.int32)
    n = n_rect + n_small
    x0 = rng.integers(0, w, n)
    y0 = rng.integers(0, h, n)
    bw = np.concatenate([rng.integers(20, max(w // 2, 21), n_rect), rng.integers(3, 24, n_small)])
    bh = np.concatenate([rng.integers(20, max(h // 2, 21), n_rect), rng.integers(3, 24, n_small)])
    val = rng.integers(0, 256, n)
    for i in range(n):
        img[y0[i]:y0[i] + bh[i], x0[i]:x0[i] + bw[i]] = val[i]
    pad = np.pad(img, 1, mode="edge")
    acc = np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            acc += pad[dy:dy + h, dx:dx + w]
    img = (acc + 4) // 9
    img = img + rng.integers(-3, 4, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_frames(batch, seed0=1000, w=640, h=480, n_rect=400, n_small=1000):
    return np.stack([synth_frame(seed0 + f, w, h, n_rect, n_small) for f in range(batch)])


def synth_descriptors(n=1000, seed=2000):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    return a, b


def synth_descriptors_planted(n=1000, seed=2001, flip=0.08):
    """B = permuted A with each bit flipped w.p. `flip` so ratio-test paths fire."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    perm = rng.permutation(n)
    bits = np.unpackbits(a[perm], axis=1)
    bits ^= (rng.random(bits.shape) < flip).astype(np.uint8)
    return a, np.packbits(bits, axis=1), perm


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def synth_ba(n_free=20, n_fixed=4, n_points=3000, seed=3000, sigma=1.0, outlier_frac=0.05, mono_frac=0.0,
             rot_noise_deg=0.5, trans_noise=0.01, point_noise=0.02, band=0):
    """Local-BA window.  Cameras on a 1 m arc facing a common volume; ids 0..n_fixed-1 fixed.
    Point i is observed by m_i = 2 + (i mod 7) cameras (i*7919 + j) mod n_cams.
    band > 0 (round 5, a MAP instead of a window): the cameras follow a trajectory of n_cams / 24 m (the same 4 cm between neighbours), point i is observed by
    m_i = 2 + (i mod (band - 1)) CONSECUTIVE cameras and lies in front of them -- every keyframe is covisible with its +-(band - 1) neighbours only, the
    reduced camera system is a band matrix.
    Returns a dict of float32/int32 arrays shaped as the C-ABI wants them (eao_ba_problem):
      poses   (n_cams, 4, 4) f32 Tcw   (perturbed for free cameras)
      fixed   (n_cams,) u8
      points  (n_points, 3) f32 (perturbed)
      edge_point, edge_cam (E,) i32 ; obs (E,3) f32 (u, v, ur; ur<0 = monocular) ; inv_sigma2 (E,) f32
      plus ground truth: poses_gt, points_gt
    Edge order = for each point (index order), its observations in ascending camera id (the reference iterates
    a std::map<KeyFrame*,size_t>, i.e. pointer order; ascending id is this build's deterministic stand-in)."""
    rng = np.random.default_rng(seed)
    n_cams = n_free + n_fixed
    pts = np.empty((n_points, 3))
    pts[:, 2] = rng.uniform(2.0, 5.0, n_points)
    pts[:, 0] = rng.uniform(-0.3, 0.3, n_points) * pts[:, 2]   # inside every camera's frustum
    pts[:, 1] = rng.uniform(-0.25, 0.25, n_points) * pts[:, 2]
    poses = np.zeros((n_cams, 4, 4))
    arc = max(1.0, n_cams / 24.0) if band else 1.0
    centres = np.zeros((n_cams, 3))
    for c in range(n_cams):
        t = c / max(n_cams - 1, 1) - 0.5            # camera centres along a 1 m arc
        if band:      # a straight trajectory with a gentle weave: every camera looks along +z, its neighbours 4 cm to either side
            centre = np.array([t * arc, 0.05 * np.sin(6 * t * arc), 0.02 * np.sin(2 * t * arc)])
            Rwc = _rot(0.02 * np.sin(3 * t * arc), 0.03 * np.sin(5 * t * arc), 0.01 * np.sin(t * arc))
        else:
            centre = np.array([t, 0.05 * np.sin(6 * t), 0.1 * t * t])
            Rwc = _rot(0.02 * np.sin(3 * t), -0.15 * t, 0.01 * t)   # look roughly at the volume centre
        Rcw = Rwc.T
        centres[c] = centre
        poses[c, :3, :3] = Rcw
        poses[c, :3, 3] = -Rcw @ centre
        poses[c, 3, 3] = 1
    e_pt, e_cam = [], []
    for i in range(n_points):
        if band:
            m = min(2 + (i % max(1, band - 1)), n_cams)
            c0 = (i * 7919) % (n_cams - m + 1)
            cams = list(range(c0, c0 + m))
            pts[i, 0] += centres[c0:c0 + m, 0].mean()          # in front of its observers
        else:
            m = 2 + (i % 7)
            cams = sorted({(i * 7919 + j) % n_cams for j in range(m)})
        for c in cams:
            e_pt.append(i)
            e_cam.append(c)
    e_pt = np.array(e_pt, dtype=np.int32)
    e_cam = np.array(e_cam, dtype=np.int32)
    E = len(e_pt)
    octave = rng.integers(0, 8, size=E)
    inv_sigma2 = (np.float32(1.0) / (np.float32(1.2) ** (2 * octave)).astype(np.float32)).astype(np.float32)
    Xc = np.einsum("eij,ej->ei", poses[e_cam, :3, :3], pts[e_pt]) + poses[e_cam, :3, 3]
    u = FX * Xc[:, 0] / Xc[:, 2] + CX
    v = FY * Xc[:, 1] / Xc[:, 2] + CY
    noise = rng.normal(0.0, 1.0, size=(E, 3)) * (sigma * 1.2 ** octave)[:, None]
    out = rng.random(E) < outlier_frac
    noise[out, :2] += rng.uniform(10, 30, size=(int(out.sum()), 2)) * rng.choice([-1, 1], size=(int(out.sum()), 2))
    ur = u - BF / Xc[:, 2]
    obs = np.stack([u + noise[:, 0], v + noise[:, 1], ur + noise[:, 0] + 0.3 * noise[:, 2]], axis=1)
    obs[:, 2] = np.maximum(obs[:, 2], 0.0)      # ur < 0 is the reference's "monocular" marker
    mono = rng.random(E) < mono_frac
    obs[mono, 2] = -1.0
    fixed = np.zeros(n_cams, dtype=np.uint8)
    fixed[:n_fixed] = 1
    poses_init = poses.copy()
    for c in range(n_fixed, n_cams):
        d = _rot(*(rng.normal(0, np.deg2rad(rot_noise_deg), 3)))
        poses_init[c, :3, :3] = d @ poses[c, :3, :3]
        poses_init[c, :3, 3] = d @ poses[c, :3, 3] + rng.normal(0, trans_noise, 3)
    pts_init = pts + rng.normal(0, point_noise, size=pts.shape)
    return dict(
        poses=poses_init.astype(np.float32), fixed=fixed, points=pts_init.astype(np.float32),
        edge_point=e_pt, edge_cam=e_cam, obs=obs.astype(np.float32), inv_sigma2=inv_sigma2,
        poses_gt=poses.astype(np.float32), points_gt=pts.astype(np.float32),
        fx=np.float32(FX), fy=np.float32(FY), cx=np.float32(CX), cy=np.float32(CY), bf=np.float32(BF),
    )


def synth_pose(n=1000, seed=4000, sigma=1.0, outlier_frac=0.1, mono_frac=0.3, n_planes=0, plane_noise=0.002):
    """PoseOptimization problem: one camera, n matched map points (world xyz f32), observations
    (u, v, ur) with ur<0 for monocular matches, octave-dependent inv_sigma2, initial pose perturbed.
    n_planes > 0 adds associated map planes (src/Optimizer.cc:456-535): plane_world (n_planes,4) = (normal, -distance) in
    the world, plane_obs the same plane seen from the true pose plus noise (the last plane is a gross outlier when
    n_planes >= 3), plane_seen (MapPlane::mbSeen)."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 3))
    pts[:, 2] = rng.uniform(2.0, 6.0, n)
    pts[:, 0] = rng.uniform(-0.45, 0.45, n) * pts[:, 2]
    pts[:, 1] = rng.uniform(-0.35, 0.35, n) * pts[:, 2]
    Rcw = _rot(0.03, -0.05, 0.02)
    tcw = np.array([0.1, -0.05, 0.2])
    Xc = pts @ Rcw.T + tcw
    octave = rng.integers(0, 8, size=n)
    inv_sigma2 = (np.float32(1.0) / (np.float32(1.2) ** (2 * octave)).astype(np.float32)).astype(np.float32)
    u = FX * Xc[:, 0] / Xc[:, 2] + CX
    v = FY * Xc[:, 1] / Xc[:, 2] + CY
    ur = u - BF / Xc[:, 2]
    noise = rng.normal(0, 1.0, size=(n, 3)) * (sigma * 1.2 ** octave)[:, None]
    out = rng.random(n) < outlier_frac
    noise[out, :2] += rng.uniform(10, 40, size=(int(out.sum()), 2)) * rng.choice([-1, 1], size=(int(out.sum()), 2))
    obs = np.stack([u + noise[:, 0], v + noise[:, 1], ur + noise[:, 0] + 0.3 * noise[:, 2]], axis=1)
    obs[:, 2] = np.maximum(obs[:, 2], 0.0)
    mono = rng.random(n) < mono_frac
    obs[mono, 2] = -1.0
    T = np.eye(4)
    d = _rot(*(rng.normal(0, np.deg2rad(1.0), 3)))
    T[:3, :3] = d @ Rcw
    T[:3, 3] = d @ tcw + rng.normal(0, 0.03, 3)
    Tgt = np.eye(4)
    Tgt[:3, :3] = Rcw
    Tgt[:3, 3] = tcw
    prob = dict(Tcw=T.astype(np.float32), Tcw_gt=Tgt.astype(np.float32), points=pts.astype(np.float32),
                obs=obs.astype(np.float32), inv_sigma2=inv_sigma2,
                fx=np.float32(FX), fy=np.float32(FY), cx=np.float32(CX), cy=np.float32(CY), bf=np.float32(BF))
    if n_planes:
        nw = rng.normal(0, 1, (n_planes, 3))
        nw /= np.linalg.norm(nw, axis=1, keepdims=True)
        dw = rng.uniform(1.0, 4.0, n_planes)
        world = np.concatenate([nw, -dw[:, None]], 1)
        nc = nw @ Rcw.T                                       # local plane: (R n, c3 - t . R n)
        local = np.concatenate([nc, (world[:, 3] - nc @ tcw)[:, None]], 1)
        local[:, :3] += rng.normal(0, plane_noise, (n_planes, 3))
        local[:, 3] += rng.normal(0, plane_noise, n_planes)
        if n_planes >= 3:
            local[-1, :3] = _rot(0.5, 0.4, 0.0) @ local[-1, :3]
            local[-1, 3] += 0.8
        local *= rng.uniform(0.5, 2.0, (n_planes, 1)) * rng.choice([-1.0, 1.0], (n_planes, 1))   # un-normalised, either sign
        prob.update(plane_world=world.astype(np.float32), plane_obs=local.astype(np.float32),
                    plane_seen=(rng.random(n_planes) < 0.6).astype(np.uint8))
    return prob


def synth_tracking(n=1000, seed=7000, flip=0.06, moved=0.03, mono_frac=0.3, occupied_frac=0.05):
    """A tracked frame pair for the guided matchers (SearchByProjection): `n` map points seen by the LAST frame, the
    CURRENT frame sees them from a slightly moved pose at noisy pixel positions (plus 10% clutter keypoints), each
    with the last descriptor corrupted by `flip` bit noise.  Returns (cur, last, mps): the frame view of the
    current frame, the last-frame arrays, and the in-view map-point arrays (TrackLocalMap query form)."""
    rng = np.random.default_rng(seed)
    scale = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)])).astype(np.float32)
    Xw = np.empty((n, 3))
    Xw[:, 2] = rng.uniform(2.0, 6.0, n)
    Xw[:, 0] = rng.uniform(-0.5, 0.5, n) * Xw[:, 2]
    Xw[:, 1] = rng.uniform(-0.4, 0.4, n) * Xw[:, 2]
    Tl = np.eye(4)
    Tc = np.eye(4)
    Tc[:3, :3] = _rot(0.01, -0.015, 0.005)
    Tc[:3, 3] = [0.02, -0.01, -moved]
    Tl32, Tc32 = Tl.astype(np.float32), Tc.astype(np.float32)
    Xc = Xw @ Tc[:3, :3].T + Tc[:3, 3]
    u = FX * Xc[:, 0] / Xc[:, 2] + CX
    v = FY * Xc[:, 1] / Xc[:, 2] + CY
    octave = rng.integers(0, 8, n).astype(np.int32)
    desc_last = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    angle_last = rng.uniform(0, 360, n).astype(np.float32)
    n_clutter = n // 10
    N = n + n_clutter
    perm = rng.permutation(N)
    kx = np.concatenate([u + rng.normal(0, 1.5, n) * scale[octave], rng.uniform(0, 640, n_clutter)])
    ky = np.concatenate([v + rng.normal(0, 1.5, n) * scale[octave], rng.uniform(0, 480, n_clutter)])
    koct = np.concatenate([np.clip(octave + rng.integers(-1, 2, n), 0, 7), rng.integers(0, 8, n_clutter)]).astype(np.int32)
    bits = np.unpackbits(desc_last, axis=1) ^ (rng.random((n, 256)) < flip).astype(np.uint8)
    kdesc = np.concatenate([np.packbits(bits, axis=1), rng.integers(0, 256, (n_clutter, 32), dtype=np.uint8)])
    kang = np.concatenate([(angle_last + 12.0 + rng.normal(0, 3, n)) % 360, rng.uniform(0, 360, n_clutter)]).astype(np.float32)
    ur_true = np.concatenate([u - BF / Xc[:, 2] + rng.normal(0, 1.0, n), rng.uniform(0, 600, n_clutter)])
    mono = rng.random(N) < mono_frac
    ur = np.where(mono, -1.0, np.maximum(ur_true, 0.5))
    occ = (rng.random(N) < occupied_frac).astype(np.uint8)
    cur = dict(kp_x=kx[perm].astype(np.float32), kp_y=ky[perm].astype(np.float32), kp_octave=koct[perm], kp_angle=kang[perm],
               u_right=ur[perm].astype(np.float32), descriptors=np.ascontiguousarray(kdesc[perm]), occupied=occ[perm],
               min_x=np.float32(0), min_y=np.float32(0), max_x=np.float32(640), max_y=np.float32(480), scale_factors=scale,
               Tcw=Tc32, fx=np.float32(FX), fy=np.float32(FY), cx=np.float32(CX), cy=np.float32(CY), mbf=np.float32(BF),
               mb=np.float32(BF / FX))
    valid = (rng.random(n) < 0.9).astype(np.uint8)
    last = dict(Tcw=Tl32, valid=valid, Xw=Xw.astype(np.float32), descriptors=desc_last, octave=octave, angle=angle_last)
    view_cos = np.where(rng.random(n) < 0.5, 0.9995, 0.97).astype(np.float32)
    mps = dict(proj_x=u.astype(np.float32), proj_y=v.astype(np.float32), proj_xr=(u - BF / Xc[:, 2]).astype(np.float32),
               view_cos=view_cos, level=octave, descriptors=desc_last, skip=(rng.random(n) < 0.1).astype(np.uint8))
    return cur, last, mps


def synth_search_scene(n=800, seed=8000, flip=0.05, clutter=0.15, mono_frac=0.3, n_nodes=60):
    """Two keyframes K1 / K2 (frame views with grid bounds, level sigmas, log scale factor) observing `n` map points from
    two poses, for the remaining guided searches (SearchByBoW, SearchForTriangulation, SearchForInitialization, Fuse,
    SearchBySim3, the loop / relocalisation SearchByProjection variants).
    Returns a dict:
      K1, K2     frame dicts; keypoint k < n of K1 observes map point perm1[k] (-1 = clutter), likewise K2 / perm2
      T1w, T2w   4x4 float32 poses;  K = (fx, fy, cx, cy), bf
      points     map-point arrays (active, Xw, normal, min/max distance invariance, max_dist, descriptors)
      mp1, mp2   per keypoint: index of its map point or -1
      fv1, fv2   DBoW2-like feature vectors (node = map point id mod n_nodes for inliers, with 10 % reassigned)
      F12, ex, ey fundamental matrix (x1' F12 x2 = 0 convention of the reference) and the epipole in image 2
      Scw        sim3 of K2's pose with scale 1.03 (float32 4x4)"""
    rng = np.random.default_rng(seed)
    nlev = 8
    scale = np.cumprod(np.concatenate([[1.0], np.full(nlev - 1, 1.2)])).astype(np.float32)
    sigma2 = (scale * scale).astype(np.float32)
    inv_sigma2 = (np.float32(1.0) / sigma2).astype(np.float32)
    logsf = np.float32(np.log(np.float32(1.2)))
    Xw = np.empty((n, 3))
    Xw[:, 2] = rng.uniform(2.5, 6.0, n)
    Xw[:, 0] = rng.uniform(-0.45, 0.45, n) * Xw[:, 2]
    Xw[:, 1] = rng.uniform(-0.35, 0.35, n) * Xw[:, 2]
    T1 = np.eye(4)
    T2 = np.eye(4)
    T2[:3, :3] = _rot(0.02, -0.05, 0.01)
    T2[:3, 3] = [-0.25, 0.03, 0.05]
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    level = rng.integers(0, nlev - 1, n).astype(np.int32)
    angle0 = rng.uniform(0, 360, n)

    def observe(T, rot_off, seed_off):
        r = np.random.default_rng(seed + seed_off)
        Xc = Xw @ T[:3, :3].T + T[:3, 3]
        u = FX * Xc[:, 0] / Xc[:, 2] + CX
        v = FY * Xc[:, 1] / Xc[:, 2] + CY
        nc = int(n * clutter)
        N = n + nc
        oct_ = np.concatenate([np.clip(level + r.integers(-1, 2, n), 0, nlev - 1), r.integers(0, nlev, nc)]).astype(np.int32)
        kx = np.concatenate([u + r.normal(0, 0.8, n) * scale[oct_[:n]], r.uniform(0, 640, nc)])
        ky = np.concatenate([v + r.normal(0, 0.8, n) * scale[oct_[:n]], r.uniform(0, 480, nc)])
        bits = np.unpackbits(desc, axis=1) ^ (r.random((n, 256)) < flip).astype(np.uint8)
        kd = np.concatenate([np.packbits(bits, axis=1), r.integers(0, 256, (nc, 32), dtype=np.uint8)])
        ka = np.concatenate([(angle0 + rot_off + r.normal(0, 3, n)) % 360, r.uniform(0, 360, nc)]).astype(np.float32)
        ur = np.concatenate([u - BF / Xc[:, 2] + r.normal(0, 0.5, n), r.uniform(0, 600, nc)])
        mono = r.random(N) < mono_frac
        ur = np.where(mono, -1.0, np.maximum(ur, 0.5))
        mp = np.concatenate([np.arange(n), np.full(nc, -1)]).astype(np.int32)
        inside = (kx > 1) & (kx < 639) & (ky > 1) & (ky < 479)
        mp = np.where(inside, mp, -1)
        perm = r.permutation(N)
        node = np.where(mp >= 0, mp % n_nodes, r.integers(0, n_nodes, N))
        node = np.where(r.random(N) < 0.1, r.integers(0, n_nodes, N), node)[perm]
        frame = dict(kp_x=kx[perm].astype(np.float32), kp_y=ky[perm].astype(np.float32), kp_octave=oct_[perm], kp_angle=ka[perm],
                     u_right=ur[perm].astype(np.float32), descriptors=np.ascontiguousarray(kd[perm]),
                     min_x=np.float32(0), min_y=np.float32(0), max_x=np.float32(640), max_y=np.float32(480), scale_factors=scale,
                     log_scale_factor=logsf, level_sigma2=sigma2, inv_level_sigma2=inv_sigma2)
        ids = np.unique(node)
        start, index = [0], []
        for nid in ids:   # keypoints of a node in ascending index order (DBoW2 appends them in feature order)
            idx = np.nonzero(node == nid)[0]
            index.extend(idx.tolist())
            start.append(len(index))
        fv = dict(node_id=ids.astype(np.uint32), node_start=np.asarray(start, np.int32), index=np.asarray(index, np.uint32))
        return frame, mp[perm], fv

    K1, mp1, fv1 = observe(T1, 0.0, 1)
    K2, mp2, fv2 = observe(T2, 15.0, 2)
    # MapPoint::UpdateNormalAndDepth (src/MapPoint.cc:330-370) with K1 as the reference keyframe
    O1 = -T1[:3, :3].T @ T1[:3, 3]
    O2 = -T2[:3, :3].T @ T2[:3, 3]
    n1 = Xw - O1
    n2 = Xw - O2
    normal = n1 / np.linalg.norm(n1, axis=1, keepdims=True) + n2 / np.linalg.norm(n2, axis=1, keepdims=True)
    normal /= 2.0
    dist = np.linalg.norm(n1, axis=1).astype(np.float32)
    max_dist = (dist * scale[level]).astype(np.float32)
    min_dist = (max_dist / scale[nlev - 1]).astype(np.float32)
    points = dict(active=(rng.random(n) < 0.92).astype(np.uint8), Xw=Xw.astype(np.float32), normal=normal.astype(np.float32),
                  min_dist_inv=(np.float32(0.8) * min_dist).astype(np.float32), max_dist_inv=(np.float32(1.2) * max_dist).astype(np.float32),
                  max_dist=max_dist, min_dist=min_dist, descriptors=desc)
    # F12 = K^-T [t12]x R12 K^-1 (LocalMapping::ComputeF12, src/LocalMapping.cc): x1' F12 x2 = 0
    R12 = T1[:3, :3] @ T2[:3, :3].T
    t12 = -R12 @ T2[:3, 3] + T1[:3, 3]
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    Km = np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1.0]])
    F12 = np.linalg.inv(Km).T @ tx @ R12 @ np.linalg.inv(Km)
    C2 = T2[:3, :3] @ O1 + T2[:3, 3]
    ex, ey = FX * C2[0] / C2[2] + CX, FY * C2[1] / C2[2] + CY
    Scw = T2.copy()
    Scw[:3, :] *= 1.03
    return dict(K1=K1, K2=K2, T1w=T1.astype(np.float32), T2w=T2.astype(np.float32), K=(np.float32(FX), np.float32(FY), np.float32(CX), np.float32(CY)),
                bf=np.float32(BF), points=points, mp1=mp1, mp2=mp2, fv1=fv1, fv2=fv2, F12=F12.astype(np.float32), ex=np.float32(ex),
                ey=np.float32(ey), Scw=Scw.astype(np.float32), R12=R12.astype(np.float32), t12=t12.astype(np.float32))


def synth_stereo_pair(seed=9000, w=640, h=480, n_planes=6):
    """A rectified stereo pair from the rectangle world: the right image is the left one re-sampled with a piecewise
    constant disparity (horizontal bands of 4 .. 40 px, i.e. fronto-parallel planes at different depths) plus independent
    sensor noise.  Returns (left, right) uint8 images."""
    rng = np.random.default_rng(seed)
    left = synth_frame(seed, w, h).astype(np.int32)
    right = np.empty_like(left)
    edges = np.sort(rng.integers(0, h, n_planes - 1))
    bands = np.concatenate([[0], edges, [h]])
    for b in range(n_planes):
        d = int(rng.integers(4, 41))
        y0, y1 = bands[b], bands[b + 1]
        right[y0:y1, :w - d] = left[y0:y1, d:]          # a point at column u in the left image sits at u - d in the right one
        right[y0:y1, w - d:] = left[y0:y1, w - 1:w]
    right = right + rng.integers(-2, 3, size=right.shape)
    return left.astype(np.uint8), np.clip(right, 0, 255).astype(np.uint8)


def add_ba_planes(prob, n_planes=5, seed=7000, angle_noise_deg=0.3, dist_noise=0.01, init_angle_deg=2.0, init_dist=0.03, outlier_edges=1):
    """Map planes for Optimizer::BundleAdjustment (reference src/Optimizer.cc:203-252) on top of a synth_ba problem: n_planes
    world planes (n, c) with n . X + c = 0 (Plane3D coefficients, c >= 0), each observed by a contiguous run of cameras;
    an observation is the plane in the camera frame (R n, c - t . R n; sign so that the fourth coefficient is >= 0) with a
    little noise on the normal and the distance, float32 like KeyFrame::mvPlaneCoefficients.  The initial world planes are
    the true ones turned by init_angle_deg and shifted by init_dist.  `outlier_edges` observations get a grossly wrong
    normal (the Huber kernel of every plane edge must take them)."""
    rng = np.random.default_rng(seed)
    poses = prob["poses_gt"].astype(np.float64)
    n_cams = len(poses)

    def unit(v):
        return v / np.linalg.norm(v)

    def canon(c):
        c = c / np.linalg.norm(c[:3])
        return -c if c[3] < 0 else c

    def turn(n, deg):
        axis = unit(np.cross(n, rng.normal(size=3)))
        a = np.deg2rad(deg)
        return unit(n * np.cos(a) + np.cross(axis, n) * np.sin(a))

    gt, init, pe_pl, pe_cam, pe_obs = [], [], [], [], []
    for i in range(n_planes):
        n = unit(rng.normal(size=3) + np.array([0.0, 0.0, 1.5]))
        c = rng.uniform(2.0, 5.0)
        w = canon(np.concatenate([n, [c]]))
        gt.append(w)
        init.append(canon(np.concatenate([turn(w[:3], init_angle_deg * rng.uniform(0.5, 1.0)), [w[3] + rng.uniform(-init_dist, init_dist)]])))
        m = int(rng.integers(max(2, n_cams // 3), n_cams + 1))
        first = int(rng.integers(0, n_cams - m + 1))
        for cam in range(first, first + m):
            R, t = poses[cam, :3, :3], poses[cam, :3, 3]
            nl = R @ w[:3]
            loc = canon(np.concatenate([nl, [w[3] - t @ nl]]))
            meas = canon(np.concatenate([turn(loc[:3], angle_noise_deg * rng.uniform(0.0, 1.0)), [loc[3] + rng.normal(0.0, dist_noise)]]))
            pe_pl.append(i); pe_cam.append(cam); pe_obs.append(meas)
    pe_obs = np.array(pe_obs)
    for k in rng.choice(len(pe_obs), size=min(outlier_edges, len(pe_obs)), replace=False):
        pe_obs[k] = canon(np.concatenate([turn(pe_obs[k][:3], 25.0), [pe_obs[k][3] + 0.4]]))
    out = dict(prob)
    out.update(planes=np.array(init, np.float32), planes_gt=np.array(gt, np.float32), pedge_plane=np.array(pe_pl, np.int32),
               pedge_cam=np.array(pe_cam, np.int32), pedge_obs=pe_obs.astype(np.float32))
    return out

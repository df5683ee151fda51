"""Deterministic synthetic inputs for the hot path (numpy only).

The licensed SMPL pickles, the images and every checkpoint are absent from the
reference checkout, so benchmarks and parity tests run
on a synthetic SMPL-like body + a synthetic canonical field:

* body  : 24 joints on the SMPL kinematic tree, 6890 vertices on capsules
          around the bones, smooth 4-nearest-bone skinning weights, a
          J_regressor whose rows average a symmetric ring around each joint
          (so J_regressor @ v_template == joints), zero shape/pose dirs.
* field : hash-table / MLP weights such that sigma ~ +s inside the canonical
          body and ~ -s outside (one dense level carries a clamped signed
          distance, two hidden units read it with +-k), everything else random.
          Memory access pattern and FLOPs are those of a trained field.

Pose tracks are read by the caller (bench.py ships a procedural track because
/root/reference does not exist on the GPU box).
"""
import numpy as np

SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int32)

# approximate SMPL neutral rest joints (metres; x = subject left, y = up)
_JOINTS = np.array([
    [-0.0018, -0.2233, 0.0282], [0.0677, -0.3147, 0.0214], [-0.0695, -0.3139, 0.0239],
    [-0.0043, -0.1144, 0.0015], [0.1020, -0.6899, 0.0169], [-0.1078, -0.6964, 0.0150],
    [0.0012, 0.0208, 0.0026], [0.0884, -1.0879, -0.0268], [-0.0920, -1.0932, -0.0272],
    [0.0026, 0.0737, 0.0280], [0.1148, -1.1437, 0.0925], [-0.1174, -1.1430, 0.0961],
    [-0.0002, 0.2876, -0.0148], [0.0815, 0.1955, -0.0060], [-0.0791, 0.1926, -0.0106],
    [0.0050, 0.3526, 0.0365], [0.1724, 0.2260, -0.0149], [-0.1752, 0.2251, -0.0197],
    [0.4320, 0.2132, -0.0424], [-0.4289, 0.2118, -0.0411], [0.6813, 0.2222, -0.0435],
    [-0.6802, 0.2195, -0.0408], [0.7653, 0.2140, -0.0585], [-0.7687, 0.2134, -0.0570],
], dtype=np.float64)

# capsule radius of the segment that ENDS at joint j (parent -> j)
_RADIUS = np.array([
    0.00, 0.11, 0.11, 0.13, 0.075, 0.075, 0.13, 0.05, 0.05, 0.13, 0.04, 0.04,
    0.06, 0.07, 0.07, 0.09, 0.06, 0.06, 0.045, 0.045, 0.035, 0.035, 0.03, 0.03])

N_VERTS = 6890
INIT_BONES = [0, 1, 2, 4, 5, 10, 11, 12, 15, 16, 17, 18, 19]  # deformer_torch.py:28


def _segments(joints):
    """(a, b, radius, controlling joint) for every capsule of the body."""
    segs = []
    for j in range(1, 24):
        p = int(SMPL_PARENTS[j])
        segs.append((joints[p], joints[j], _RADIUS[j], p))
    # leaf extensions controlled by the leaf joint itself
    for j, ext, r in [(15, [0.0, 0.13, 0.02], 0.10), (22, [0.08, 0, 0], 0.03),
                      (23, [-0.08, 0, 0], 0.03), (10, [0, 0, 0.08], 0.035),
                      (11, [0, 0, 0.08], 0.035)]:
        segs.append((joints[j], joints[j] + np.array(ext), r, j))
    return segs


def _seg_dist(x, a, b):
    ab = b - a
    t = np.clip(((x - a) @ ab) / max(float(ab @ ab), 1e-12), 0.0, 1.0)
    return np.linalg.norm(x - (a + t[:, None] * ab), axis=1)


def capsule_sdf(x, joints, margin=0.0):
    """> 0 inside the union of capsules placed on `joints` (any pose)."""
    x = np.asarray(x, np.float64)
    best = np.full(len(x), -1e9)
    for a, b, r, _ in _segments(np.asarray(joints, np.float64)):
        best = np.maximum(best, (r + margin) - _seg_dist(x, a, b))
    return best


def _frame(v):
    v = v / (np.linalg.norm(v) + 1e-12)
    h = np.array([1.0, 0, 0]) if abs(v[0]) < 0.9 else np.array([0, 1.0, 0])
    u = np.cross(v, h)
    u /= np.linalg.norm(u)
    return u, np.cross(v, u)


#: the shape coefficients the blend-shape test world is initialised with (tests/world.py:build_blend, the *_blend goldens)
BLEND_BETAS = np.array([1.3, -0.9, 0.7, -1.1, 0.5, 0.8, -0.6, 1.0, -0.4, 0.9], np.float32)


def _smooth_field(rng, pts, n, amp, freq):
    """n smooth 3-vector displacement fields over `pts` [V,3]: amp_k * sin(F_k p + phi_k) per component with low spatial
    frequencies (rad / m) -- blend shapes that bend and swell the body instead of adding per-vertex noise -> [n, V, 3]"""
    F = rng.uniform(-freq, freq, (n, 3, 3))
    phi = rng.uniform(0, 2 * np.pi, (n, 1, 3))
    a = rng.uniform(0.5, 1.0, (n, 1, 3)) * amp * rng.choice([-1.0, 1.0], (n, 1, 3))
    return a * np.sin(np.einsum("vj,nkj->nvk", pts, F) + phi)


def make_body(seed=42, blendshapes=False):
    """Synthetic SMPL-like body model (dict of numpy arrays, fp32).

    blendshapes=False: zero shapedirs / posedirs and a one-hot-ring J_regressor (the body of rounds 1-5: betas
    are then a no-op).  blendshapes=True: what a real SMPL pickle has -- 10 non-zero shape directions (the first scales the
    body about its centre, the others are smooth ~2 cm displacement fields), 207 smooth ~1 cm pose-corrective directions and
    a DENSE joint regressor (every joint a positive combination of its ~64 nearest vertices), so that betas move the rest
    joints, the rest-pose vertices (-> voxelised weights) and the posed vertices (lbs.py:185-222)."""
    rng = np.random.RandomState(seed)
    J = _JOINTS.copy()
    segs = _segments(J)
    verts = []
    # 8-vertex symmetric ring around every joint -> J_regressor rows
    ring_ids = []
    for j in range(24):
        p = int(SMPL_PARENTS[j])
        axis = (J[j] - J[p]) if p >= 0 else np.array([0.0, 1.0, 0.0])
        u, w = _frame(axis)
        r = max(_RADIUS[j], 0.05)
        ids = []
        for k in range(8):
            a = 2 * np.pi * k / 8
            ids.append(len(verts))
            verts.append(J[j] + r * (np.cos(a) * u + np.sin(a) * w))
        ring_ids.append(ids)
    n_left = N_VERTS - len(verts)
    area = np.array([2 * np.pi * r * (np.linalg.norm(b - a) + 2 * r) for a, b, r, _ in segs])
    cnt = np.floor(area / area.sum() * n_left).astype(int)
    cnt[0] += n_left - cnt.sum()
    for (a, b, r, _), n in zip(segs, cnt):
        L = np.linalg.norm(b - a)
        u, w = _frame(b - a)
        ax = (b - a) / (L + 1e-12)
        # points on the capsule surface: cylinder part + two hemispherical caps
        t = rng.uniform(-r, L + r, n)
        phi = rng.uniform(0, 2 * np.pi, n)
        tc = np.clip(t, 0, L)
        # radial distance so that the point lies on the capsule surface
        rad = np.sqrt(np.maximum(r * r - (t - tc) ** 2, 0.0))
        pts = a + t[:, None] * ax + rad[:, None] * (np.cos(phi)[:, None] * u + np.sin(phi)[:, None] * w)
        verts.extend(pts)
    verts = np.asarray(verts)
    assert verts.shape == (N_VERTS, 3)
    # skinning weights: inverse 4th-power distance to the 4 nearest bones
    dist = np.full((N_VERTS, 24), 1e9)
    for a, b, r, ctrl in segs:
        dist[:, ctrl] = np.minimum(dist[:, ctrl], _seg_dist(verts, a, b))
    wgt = 1.0 / (dist + 0.02) ** 4
    order = np.argsort(-wgt, axis=1)
    mask = np.zeros_like(wgt)
    np.put_along_axis(mask, order[:, :4], 1.0, axis=1)
    wgt = wgt * mask
    wgt /= wgt.sum(1, keepdims=True)
    Jreg = np.zeros((24, N_VERTS))
    for j in range(24):
        Jreg[j, ring_ids[j]] = 1.0 / 8
    shapedirs = np.zeros((N_VERTS, 3, 10), np.float32)
    posedirs = np.zeros((207, N_VERTS * 3), np.float32)
    if blendshapes:
        rb = np.random.RandomState(seed + 7)      # (own stream: the vertices above stay those of the default body)
        shapedirs = np.transpose(_smooth_field(rb, verts, 10, 0.02, 4.0), (1, 2, 0)).copy()
        shapedirs[:, :, 0] = 0.03 * (verts - verts.mean(0))                       # "size": +-3 % per unit of beta_0
        shapedirs[:, :, 1] = 0.03 * (verts - verts.mean(0)) * np.array([1.0, -0.5, 1.0])   # "build": wider and shorter
        posedirs = _smooth_field(rb, verts, 207, 0.01, 3.0).reshape(207, N_VERTS * 3)
        d2 = ((J[:, None, :] - verts[None]) ** 2).sum(-1)                          # [24, V]
        near = np.argsort(d2, axis=1)[:, :64]
        for j in range(24):
            Jreg[j, near[j]] += rb.uniform(0.2, 1.0, 64) / 64
        Jreg /= Jreg.sum(1, keepdims=True)
    return dict(
        v_template=verts.astype(np.float32),
        shapedirs=shapedirs.astype(np.float32),
        posedirs=posedirs.astype(np.float32),
        J_regressor=Jreg.astype(np.float32),
        parents=SMPL_PARENTS.copy(),
        lbs_weights=wgt.astype(np.float32),
        joints_template=J.astype(np.float32),
    )


def cano_pose(name="A_pose"):
    """snarf_deformer.py:6-18 (get_predefined_rest_pose)."""
    p = np.zeros(69, np.float32)
    if name.lower() == "da_pose":
        p[2] = np.pi / 6
        p[5] = -np.pi / 6
    elif name.lower() == "a_pose":
        p[2], p[5], p[47], p[50] = 0.2, -0.2, -0.8, 0.8
    else:
        raise ValueError("Unknown cano_pose: {}".format(name))
    return p


def hash_level_table(n_levels=16, log2_T=19, base=16, pls=1.5):
    """tcnn-v1.6 level table (scale, res, offset) from the library's host routine
    ia_hash_desc_init, so data generation and kernels can never disagree."""
    from . import _lib
    hd = _lib.make_hash_desc(n_levels, log2_T, base, pls)
    return (np.array(hd.scale[:n_levels], np.float32), np.array(hd.res[:n_levels], np.uint32),
            np.array(hd.offset[:n_levels + 1], np.uint32))


def make_field(cano_joints, bbox, seed=42, n_levels=16, log2_T=19, sigma_in=120.0,
               sdf_level=3):
    """Synthetic NeRFNGPNet parameters (fp16 arrays) whose density follows the
    canonical capsule body.  bbox: [2,3] as NeRFNGPNet.initialize gets it."""
    rng = np.random.RandomState(seed + 1)
    scale, res, off = hash_level_table(n_levels, log2_T)
    n_entries = int(off[-1])
    table = (rng.uniform(-1, 1, (n_entries, 2)) * 0.5).astype(np.float32)
    center = ((bbox[0] + bbox[1]) / 2).astype(np.float32)
    fscale = (bbox[1] - bbox[0]).astype(np.float32)
    sdf_level = min(sdf_level, n_levels - 1)
    r = int(res[sdf_level])
    assert r ** 3 <= off[sdf_level + 1] - off[sdf_level], "sdf level must be dense"
    # grid vertex g of level l sits at x_n = (g - 0.5) / scale_l  (pos = x*s+0.5)
    g = np.arange(r)
    gx, gy, gz = np.meshgrid(g, g, g, indexing="ij")
    xn = (np.stack([gx, gy, gz], -1).reshape(-1, 3) - 0.5) / scale[sdf_level]
    xw = (xn - 0.5) * fscale + center
    sdf = capsule_sdf(xw, cano_joints)
    val = np.clip(sdf / 0.03, -1, 1)
    idx = gx.reshape(-1) + gy.reshape(-1) * r + gz.reshape(-1) * r * r
    table[int(off[sdf_level]) + idx, 0] = val

    def xavier(o, i):
        a = np.sqrt(6.0 / (o + i))
        return rng.uniform(-a, a, (o, i)).astype(np.float32)

    k, c = 4.0, sigma_in / 4.0
    sig_w1 = xavier(64, 2 * n_levels)
    sig_w1[0] = 0; sig_w1[1] = 0
    sig_w1[0, sdf_level * 2] = k
    sig_w1[1, sdf_level * 2] = -k
    sig_w2 = xavier(16, 64)
    sig_w2[0] *= 0.1
    sig_w2[0, 0], sig_w2[0, 1] = c, -c
    col_w1, col_w2, col_w3 = xavier(64, 16), xavier(64, 64), xavier(16, 64)
    f16 = lambda a: np.ascontiguousarray(a.astype(np.float16))
    return dict(center=center, scale=fscale, n_levels=n_levels, log2_T=log2_T,
                table=f16(table), sig_w1=f16(sig_w1), sig_w2=f16(sig_w2),
                col_w1=f16(col_w1), col_w2=f16(col_w2), col_w3=f16(col_w3),
                level_scale=scale, level_res=res, level_offset=off)


def make_camera_rays(res, H0=1080, f0=2000.0):
    """animate.py:17-44: pinhole, c2w = I, principal point at the centre,
    focal 2000 * (res / 1080); returns (rays_o, rays_d) fp32 [res*res,3]."""
    f = f0 * res / H0
    c = (H0 // 2) * res / H0
    u, v = np.meshgrid(np.arange(res, dtype=np.float64), np.arange(res, dtype=np.float64), indexing="xy")
    d = np.stack([(u - c) / f, (v - c) / f, np.ones_like(u)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.zeros_like(d, dtype=np.float32), d.astype(np.float32)


def procedural_pose_track(n_frames, seed=42):
    """Smooth dance-like SMPL pose track (stand-in for data/animation/
    aist_demo.npz, which is not available on the GPU box): poses [n,72],
    transl [n,3] with transl = (dx, 0.15, 5) as animate.py:48-50."""
    rng = np.random.RandomState(seed + 7)
    t = np.linspace(0, 2 * np.pi, n_frames, endpoint=False)
    amp = rng.uniform(0.1, 0.5, (72,))
    ph = rng.uniform(0, 2 * np.pi, (72,))
    frq = rng.randint(1, 4, (72,))
    poses = amp[None] * np.sin(frq[None] * t[:, None] + ph[None])
    poses[:, 0:3] *= 0.3
    poses[:, 0] += np.pi  # face the camera the way aist_demo does (y down in camera)
    tr = np.stack([0.2 * np.sin(t), np.full_like(t, 0.15), np.full_like(t, 5.0)], 1)
    return poses.astype(np.float32), tr.astype(np.float32)


def load_animation_track(path):
    """animate.py:46-50: `poses[..., :72]` and `trans - trans[0] + (0, 0.15, 5)` of an AIST-style pose file
    (tests/golden/aist_demo_200.npz = the first 200 frames of the reference's data/animation/aist_demo.npz)."""
    z = np.load(path)
    poses = z["poses"][..., :72].astype(np.float32)
    tr = (z["trans"] - z["trans"][0:1] + np.array([0, 0.15, 5.0])).astype(np.float32)
    return poses, tr

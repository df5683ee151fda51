"""Python side of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see the header of
ia_oracle.c; nothing under instantavatar_amd/ may import this module).

ctypes bindings to libia_oracle.so plus numpy restatements of the reference's
torch-level glue, each citing the reference lines it follows:

  lbs / SMPL.forward          deformers/smplx/lbs.py:152-250,295-401
                              deformers/smplx/body_models.py:337-360
  prepare_deformer            deformers/snarf_deformer.py:41-93
  switch_to_explicit          deformers/fast_snarf/deformer_torch.py:130-169
  deform_test                 deformers/snarf_deformer.py:109-141
  transform_rays_w2s          deformers/snarf_deformer.py:95-103
  DensityGrid.initialize      models/structures/density_grid.py:95-110
  Raymarcher.render_test      renderers/raymarcher_acc.py:83-138
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
F32P = C.POINTER(C.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libia_oracle.so")
        src_t = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ia_oracle.c", "Makefile"))
        fma_here = " fma " in open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else False
        if not os.path.exists(so) or os.path.getmtime(so) < src_t or not fma_here:
            if os.path.exists(so) and not fma_here:
                os.remove(so)  # prebuilt with -mfma but this host lacks it: rebuild portable
            build()
        _LIB = C.CDLL(so)
    return _LIB


class HashDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("scale", C.c_float * 16),
                ("res", C.c_uint32 * 16), ("offset", C.c_uint32 * 17)]


class Field(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("scale", C.c_float * 3), ("hash", HashDesc),
                ("table", C.c_void_p), ("sig_w1", C.c_void_p), ("sig_w2", C.c_void_p),
                ("col_w1", C.c_void_p), ("col_w2", C.c_void_p), ("col_w3", C.c_void_p)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def hash_desc(n_levels=16, log2_T=19, base=16, pls=1.5, level3_res=None):
    """tcnn v1.6 level table.  level3_res / IA_TCNN_LEVEL3_RES in {54, 55}: the one resolution that depends on the last
    bit of exp2f (see instantavatar_amd/_lib.py: apply_level3_override); default = this host's libm (glibc: 54)."""
    hd = HashDesc()
    lib().orc_hash_desc_init(C.byref(hd), n_levels, log2_T, base, C.c_float(pls))
    if level3_res is None:
        level3_res = os.environ.get("IA_TCNN_LEVEL3_RES")
    if level3_res is not None and n_levels > 3:
        hd.res[3] = int(level3_res)
        off = 0
        for l in range(n_levels):
            r = int(hd.res[l])
            hd.offset[l] = off
            off += min((r * r * r + 7) // 8 * 8, 1 << log2_T)
        hd.offset[n_levels] = off
    return hd


def make_field(fp):
    """fp: dict from instantavatar_amd.synthetic.make_field (or a checkpoint).
    Returns (Field struct, keep-alive list)."""
    f = Field()
    keep = []
    f.center[:] = [float(v) for v in fp["center"]]
    f.scale[:] = [float(v) for v in fp["scale"]]
    f.hash = hash_desc(fp.get("n_levels", 16), fp.get("log2_T", 19))
    for k in ["table", "sig_w1", "sig_w2", "col_w1", "col_w2", "col_w3"]:
        a = np.ascontiguousarray(fp[k]).view(np.uint16)
        keep.append(a)
        setattr(f, k, a.ctypes.data)
    return f, keep


# --------------------------------------------------------------------------
# a1: LBS joint chain (lbs.py), numpy fp32
# --------------------------------------------------------------------------
def batch_rodrigues(rot_vecs):
    """lbs.py:295-329"""
    rot_vecs = _f32(rot_vecs)
    angle = np.linalg.norm(rot_vecs + np.float32(1e-8), axis=1, keepdims=True).astype(np.float32)
    rot_dir = rot_vecs / angle
    cos = np.cos(angle)[:, None].astype(np.float32)
    sin = np.sin(angle)[:, None].astype(np.float32)
    rx, ry, rz = rot_dir[:, 0:1], rot_dir[:, 1:2], rot_dir[:, 2:3]
    z = np.zeros_like(rx)
    K = np.concatenate([z, -rz, ry, rz, z, -rx, -ry, rx, z], axis=1).reshape(-1, 3, 3)
    ident = np.eye(3, dtype=np.float32)[None]
    return (ident + sin * K + (1 - cos) * (K @ K)).astype(np.float32)


def batch_rigid_transform(rot_mats, joints, parents):
    """lbs.py:345-401 (single batch element).  rot_mats [24,3,3], joints [24,3]."""
    joints = _f32(joints)[:, :, None]
    rel = joints.copy()
    rel[1:] -= joints[parents[1:]]
    tm = np.zeros((24, 4, 4), np.float32)
    tm[:, :3, :3] = rot_mats
    tm[:, :3, 3:] = rel
    tm[:, 3, 3] = 1
    chain = [tm[0]]
    for i in range(1, 24):
        chain.append((chain[parents[i]] @ tm[i]).astype(np.float32))
    transforms = np.stack(chain, 0)
    posed = transforms[:, :3, 3].copy()
    jh = np.concatenate([joints, np.zeros((24, 1, 1), np.float32)], 1)
    corr = np.zeros((24, 4, 4), np.float32)
    corr[:, :, 3:] = transforms @ jh
    return posed, (transforms - corr).astype(np.float32)


def smpl_forward(body, betas, body_pose, global_orient=None, transl=None, want_verts=True):
    """SMPL.forward (body_models.py:289-372) + lbs (lbs.py:152-250), B = 1."""
    betas = _f32(betas).reshape(-1)
    v_shaped = body["v_template"] + np.einsum("l,mkl->mk", betas, body["shapedirs"]).astype(np.float32)
    J = (body["J_regressor"] @ v_shaped).astype(np.float32)
    go = np.zeros(3, np.float32) if global_orient is None else _f32(global_orient).reshape(3)
    full = np.concatenate([go, _f32(body_pose).reshape(69)])
    rot = batch_rodrigues(full.reshape(-1, 3))
    Jt, A = batch_rigid_transform(rot, J, body["parents"])
    out = dict(A=A, joints=Jt, J_rest=J, rot=rot)
    if want_verts:
        pose_feature = (rot[1:] - np.eye(3, dtype=np.float32)).reshape(-1)
        v_posed = v_shaped + (pose_feature @ body["posedirs"]).reshape(-1, 3).astype(np.float32)
        T = (body["lbs_weights"] @ A.reshape(24, 16)).reshape(-1, 4, 4)
        vh = np.concatenate([v_posed, np.ones((len(v_posed), 1), np.float32)], 1)
        out["vertices"] = np.einsum("vij,vj->vi", T, vh)[:, :3].astype(np.float32)
        out["T"] = T.astype(np.float32)
        out["shape_offsets"] = (v_shaped - body["v_template"]).astype(np.float32)       # lbs.py:185-187
        out["pose_offsets"] = (v_posed - v_shaped).astype(np.float32)                    # lbs.py:211-222
    if transl is not None:  # body_models.py:353-360
        t = _f32(transl).reshape(3)
        out["A"] = A.copy()
        out["A"][:, :3, 3] += t
        out["joints"] = Jt + t
        if want_verts:
            out["vertices"] = out["vertices"] + t
            out["T"] = out["T"].copy()
            out["T"][:, :3, 3] += t
    return out


# --------------------------------------------------------------------------
# a20 + a2: deformer initialisation and per-frame preparation
# --------------------------------------------------------------------------
def deformer_initialize(body, betas, cano_pose, resolution=128, n_smooth=30, global_scale=1.2):
    """SNARFDeformer.initialize (snarf_deformer.py:41-69) +
    ForwardDeformer.switch_to_explicit (deformer_torch.py:130-186)."""
    so = smpl_forward(body, betas, cano_pose)
    tfs_inv_t = np.linalg.inv(so["A"].astype(np.float32)).astype(np.float32)
    vs = so["vertices"]
    d, h, w = resolution // 4, resolution, resolution
    ratio = h / d
    lin = lambda n: np.linspace(-1, 1, n, dtype=np.float32)
    gz, gy, gx = np.meshgrid(lin(d), lin(h), lin(w), indexing="ij")
    grid = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    gmin, gmax = vs.min(0), vs.max(0)
    offset = ((gmin + gmax) * np.float32(0.5)).astype(np.float32)
    scale = np.float32((gmax - gmin).max() / 2 * global_scale)
    offset_kernel = -offset
    scale_kernel = np.full(3, 1.0 / scale, np.float32)
    scale_kernel[2] *= ratio
    gd = grid.copy()                      # denormalize :164-169
    gd[:, 2] /= np.float32(ratio)
    gd *= scale
    gd += offset
    weights = np.empty((24, d, h, w), np.float32)
    lib().orc_query_weights_smpl(_p(_f32(gd)), C.c_long(len(gd)), _p(_f32(vs)), C.c_int(len(vs)),
                                 _p(_f32(body["lbs_weights"])), d, h, w, n_smooth, _p(weights))
    # get_bbox_from_smpl (snarf_deformer.py:20-31)
    c = (gmax + gmin) / 2
    s = ((gmax - gmin) / 2).max() * np.float32(1.2)
    bbox = np.stack([c - s, c + s]).astype(np.float32)
    return dict(tfs_inv_t=tfs_inv_t, vs_template=vs, lbs_voxel=weights, offset_kernel=offset_kernel,
                scale_kernel=scale_kernel, bbox=bbox, D=d, H=h, W=w, cano_joints=so["joints"],
                grid_denorm=gd)


def knn(pts, verts, K=30):
    """pytorch3d knn_points (deformer_torch.py:227): (squared distances ascending [N,K], indices [N,K])."""
    pts, verts = _f32(pts).reshape(-1, 3), _f32(verts).reshape(-1, 3)
    assert K <= 64
    d = np.empty((len(pts), K), np.float32)
    i = np.empty((len(pts), K), np.int64)
    lib().orc_knn(_p(pts), C.c_long(len(pts)), _p(verts), C.c_int(len(verts)), C.c_int(K), _p(d), _p(i))
    return d, i


def prepare_deformer(body, init, betas, body_pose, global_orient, transl):
    """snarf_deformer.py:71-93 -> tfs [24,4,4], w2s [4,4] and precompute."""
    so = smpl_forward(body, betas, body_pose, global_orient, transl, want_verts=False)
    s2w = so["A"][0].astype(np.float32)
    w2s = np.linalg.inv(s2w).astype(np.float32)
    tfs = (w2s[None] @ so["A"] @ init["tfs_inv_t"]).astype(np.float32)
    return tfs, w2s


def precompute(init, tfs):
    """ForwardDeformer.precompute (deformer_torch.py:77-83).  Returns voxel_J
    [12,D,H,W] and voxel_d [3,D,H,W] in the REFERENCE layout."""
    D, H, W = init["D"], init["H"], init["W"]
    vJ = np.empty((12, D, H, W), np.float32)
    vd = np.empty((3, D, H, W), np.float32)
    lib().orc_precompute(_p(init["lbs_voxel"]), _p(_f32(tfs)), _p(vJ), _p(vd), _p(init["offset_kernel"]),
                         _p(init["scale_kernel"]), D, H, W)
    return vJ, vd


def broyden(xd, voxel_J, tfs, init, bone_ids, cvg=1e-5, dvg=1e-1, want_iters=False):
    """broyden_cuda (deformer_torch.py:100-116) without the filter."""
    xd = _f32(xd).reshape(-1, 3)
    P, n = len(xd), len(bone_ids)
    x = np.empty((P, n, 3), np.float32)
    Ji = np.empty((P, n, 3, 3), np.float32)
    valid = np.empty((P, n), np.uint8)
    it = np.zeros((P, n), np.uint8)
    b = np.ascontiguousarray(bone_ids, np.int32)
    lib().orc_broyden(_p(xd), C.c_long(P), _p(voxel_J), init["D"], init["H"], init["W"], _p(_f32(tfs)),
                      _p(b), n, _p(init["offset_kernel"]), _p(init["scale_kernel"]), C.c_float(cvg),
                      C.c_float(dvg), _p(x), _p(Ji), _p(valid), _p(it))
    return (x, Ji, valid, it) if want_iters else (x, Ji, valid)


def filter_dup(x, mask):
    out = np.empty_like(mask)
    lib().orc_filter(_p(x), _p(mask), C.c_long(x.shape[0]), x.shape[1], _p(out))
    return out


def search(xd, voxel_J, tfs, init, bone_ids):
    x, Ji, valid = broyden(xd, voxel_J, tfs, init, bone_ids)
    return x, filter_dup(x, valid), Ji, valid


def field_fwd(field, x):
    x = _f32(x).reshape(-1, 3)
    rgb = np.empty((len(x), 3), np.float32)
    sig = np.empty(len(x), np.float32)
    lib().orc_field_fwd(C.byref(field), _p(x), C.c_long(len(x)), _p(rgb), _p(sig))
    return rgb, sig


def tcnn_encoder(field, xn):
    """ngp.py:78 `self.encoder(x)` on unit coordinates -> [n,16] (float values of the half outputs)."""
    xn = _f32(xn).reshape(-1, 3)
    out = np.empty((len(xn), 16), np.float32)
    lib().orc_tcnn_encoder(C.byref(field), _p(xn), C.c_long(len(xn)), _p(out))
    return out


def tcnn_color(field, in15):
    """ngp.py:81 `self.color_net(x[..., 1:])` -> [n,3]."""
    in15 = _f32(in15).reshape(-1, 15)
    out = np.empty((len(in15), 3), np.float32)
    lib().orc_tcnn_color(C.byref(field), _p(in15), C.c_long(len(in15)), _p(out))
    return out


def hashgrid(field, x):
    x = _f32(x).reshape(-1, 3)
    feat = np.zeros((len(x), 32), np.uint16)
    lib().orc_hashgrid(C.byref(field), _p(x), C.c_long(len(x)), _p(feat))
    return feat.view(np.float16)[:, :2 * field.hash.n_levels]


def deform_query(pts, world, eval_mode=True):
    """SNARFDeformer.__call__ (snarf_deformer.py:127-165): returns rgb, sigma."""
    pts = _f32(pts).reshape(-1, 3)
    x, valid, _, _ = search(pts, world["voxel_J"], world["tfs"], world["init"], world["bone_ids"])
    P, Cn = valid.shape
    rgb_c = np.zeros((P, Cn, 3), np.float32)
    sig_c = np.zeros((P, Cn), np.float32)
    m = valid.astype(bool)
    if m.any():
        r, s = field_fwd(world["field"], x[m])
        rgb_c[m], sig_c[m] = r, s
    rgb = np.empty((P, 3), np.float32)
    sig = np.empty(P, np.float32)
    lib().orc_candidate_max(_p(rgb_c), _p(sig_c), _p(valid), C.c_long(P), Cn,
                            C.c_float(0.0 if eval_mode else -1e5), 1 if eval_mode else 0, _p(rgb), _p(sig))
    return rgb, sig


def transform_rays_w2s(o, d, w2s):
    """snarf_deformer.py:95-103"""
    o2 = (_f32(o) @ w2s[:3, :3].T + w2s[:3, 3]).astype(np.float32)
    d2 = (_f32(d) @ w2s[:3, :3].T).astype(np.float32)
    dist = np.linalg.norm(o2, axis=-1).astype(np.float32)
    return o2, d2, dist - 1, dist + 1


def occupancy_from_density(density, G=64):
    occ = np.empty(G ** 3, np.uint8)
    lib().orc_occupancy_from_density(_p(_f32(density).reshape(-1)), G, _p(occ), None)
    return occ.reshape(G, G, G)


def density_grid_initialize(world, jitter, G=64):
    """DensityGrid.initialize (density_grid.py:95-110).  jitter [iters,G^3,3]
    replaces torch.rand_like (:100).  Returns (aabb [2,3], density, occ)."""
    vd = world["voxel_d"].reshape(3, -1)
    aabb = np.stack([vd.min(1), vd.max(1)]).astype(np.float32)   # get_bbox_deformed
    idx = np.arange(G, dtype=np.float32)
    cx, cy, cz = np.meshgrid(idx, idx, idx, indexing="ij")
    coords0 = (np.stack([cx, cy, cz], -1).reshape(-1, 3) / np.float32(G)).astype(np.float32)
    density = np.zeros(G ** 3, np.float32)
    for it in range(len(jitter)):
        coords = (coords0 + _f32(jitter[it]) / np.float32(G)) * (aabb[1] - aabb[0]) + aabb[0]
        _, d = deform_query(coords.astype(np.float32), world, eval_mode=True)
        density = np.maximum(density, d)
    return aabb, density, occupancy_from_density(density, G)


#: aabb of the training occupancy grid (Raymarcher.__init__, raymarcher_acc.py:56)
TRAIN_AABB = np.array([[-1.25, -1.55, -1.25], [1.25, 0.95, 1.25]], np.float32)


def density_grid_update(world, density_cached, density_field_old, jitter, step, G=64, aabb=TRAIN_AABB):
    """DensityGrid.update, the branch every non-`smpl_init` configuration takes (density_grid.py:46-51,77-92).
    jitter [G^3,3] replaces torch.rand_like (:47).  Returns dict(density_cached, density_field, density, valid):
    the new state plus the two values handed to DNeRFModel.update_density_grid."""
    idx = np.arange(G, dtype=np.float32)
    cx, cy, cz = np.meshgrid(idx, idx, idx, indexing="ij")
    coords0 = (np.stack([cx, cy, cz], -1).reshape(-1, 3) / np.float32(G)).astype(np.float32)
    coords = ((coords0 + _f32(jitter).reshape(-1, 3) / np.float32(G)) * (aabb[1] - aabb[0]) + aabb[0]).astype(np.float32)
    _, density = deform_query(coords, world, eval_mode=False)           # :48-49 (invalid candidates: -1e5)
    density = np.maximum(density, np.float32(0)).astype(np.float32)     # :50 clip(min=0)
    cached = np.maximum(_f32(density_cached).reshape(-1) * np.float32(0.8), density).astype(np.float32)  # :77
    field = occupancy_from_density(cached, G)                           # :78-85
    dens_out = (np.float32(1) - np.exp(np.float32(0.01) * -np.maximum(density, np.float32(0)))).astype(np.float32)  # :87
    valid = field.astype(bool) if step < 500 else np.asarray(density_field_old).astype(bool)             # :88-91
    return dict(density_cached=cached.reshape(G, G, G), density_field=field.reshape(G, G, G).astype(bool),
                density=dens_out.reshape(G, G, G), valid=valid.reshape(G, G, G))


def mesh_signed_distance(pts, verts, faces):
    """kaolin point_to_mesh_distance(...).sqrt() * (1 - 2 * check_sign(...)) as used at density_grid.py:62-70
    (restated from the definitions; kaolin itself is absent)."""
    pts, verts = _f32(pts).reshape(-1, 3), _f32(verts).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, np.int32).reshape(-1, 3)
    out = np.empty(len(pts), np.float32)
    lib().orc_mesh_sdf(_p(pts), C.c_long(len(pts)), _p(verts), _p(faces), C.c_int(len(faces)), _p(out))
    return out


def density_grid_smpl_init(verts, faces, density_cached, G=64, aabb=TRAIN_AABB):
    """DensityGrid.update, first call of the `smpl_init` branch (density_grid.py:53-75): cells whose centre lies inside the
    posed mesh or within 1 cm of it start occupied; their cached density becomes +inf (-log(1 - 1) * 100)."""
    idx = np.arange(G, dtype=np.float32)
    cx, cy, cz = np.meshgrid(idx, idx, idx, indexing="ij")
    coords0 = (np.stack([cx, cy, cz], -1).reshape(-1, 3) / np.float32(G)).astype(np.float32)
    coords = ((coords0 + np.float32(0.5) / np.float32(G)) * (aabb[1] - aabb[0]) + aabb[0]).astype(np.float32)
    sd = mesh_signed_distance(coords, verts, faces)
    field = sd < np.float32(0.01)
    with np.errstate(divide="ignore"):
        opacity = (-np.log(np.float32(1) - field.astype(np.float32)) * np.float32(100)).astype(np.float32)
    cached = np.maximum(_f32(density_cached).reshape(-1) * np.float32(0.8), opacity)
    return dict(signed_distance=sd.reshape(G, G, G), density_field=field.reshape(G, G, G), density_cached=cached.reshape(G, G, G))


def update_density_grid_reg(density, valid, step, N=20):
    """DNeRFModel.update_density_grid (DNeRF.py:99-110): reg = N * mean(density outside the grid)
    (+ 0.5 * mean(density) for the first 500 steps)."""
    d = _f32(density).reshape(-1).astype(np.float64)
    out = d[~np.asarray(valid).reshape(-1)]
    reg = N * (out.mean() if len(out) else float("nan"))
    if step < 500:
        reg += 0.5 * d.mean()
    return float(reg)


def render_test(o, d, near, far, occ, aabb, model, MAX_SAMPLES=256, MAX_BATCH_SIZE=291600, bg=None):
    """Raymarcher.render_test (raymarcher_acc.py:83-138).  model(pts)->(rgb,sigma)."""
    L = lib()
    o = _f32(o).reshape(-1, 3); d = _f32(d).reshape(-1, 3)
    near = _f32(near).reshape(-1).copy(); far = _f32(far).reshape(-1)
    N = len(o)
    color = np.zeros((N, 3), np.float32); depth = np.zeros(N, np.float32)
    no_hit = np.ones(N, np.float32); counter = np.zeros(N, np.float32)
    alive = np.arange(N, dtype=np.int64)
    step = ((far - near) / np.float32(MAX_SAMPLES)).astype(np.float32)
    offset = _f32(aabb[0]); scale = _f32(aabb[1] - aabb[0])
    G = occ.shape[0]
    occ8 = np.ascontiguousarray(occ, np.uint8)
    k = 0
    n_field = 0
    while k < MAX_SAMPLES:
        Na = len(alive)
        if Na == 0:
            break
        Ns = max(min(MAX_BATCH_SIZE // Na, MAX_SAMPLES), 1)
        pts = np.empty((Na, Ns, 3), np.float32); dn = np.empty((Na, Ns), np.float32); zn = np.empty((Na, Ns), np.float32)
        L.orc_raymarch_test(_p(o), _p(d), _p(near), _p(far), _p(alive), C.c_long(Na), _p(occ8), G, _p(scale),
                            _p(offset), _p(step), Ns, _p(pts), _p(dn), _p(zn))
        counter[alive] += (dn > 0).sum(-1)
        mask = dn > 0
        rgb = np.zeros_like(pts); sig = np.zeros((Na, Ns), np.float32)
        if mask.any():
            r, s = model(pts[mask])
            rgb[mask], sig[mask] = r, s
            n_field += int(mask.sum())
        L.orc_composite_test(_p(rgb), _p(sig), _p(dn), _p(zn), _p(alive), C.c_long(Na), Ns, _p(color), _p(depth),
                             _p(no_hit), C.c_float(0.01))
        alive = alive[(no_hit[alive] > 1e-4) & (zn[:, -1] > 0)]
        k += Ns
    color = color + no_hit[:, None] * (1.0 if bg is None else _f32(bg).reshape(-1, 3))
    return dict(rgb=color.astype(np.float32), depth=depth, alpha=1 - no_hit, counter=counter, n_field=n_field)


def make_world(body, init, field_params, betas, body_pose, global_orient, transl, bone_ids):
    tfs, w2s = prepare_deformer(body, init, betas, body_pose, global_orient, transl)
    vJ, vd = precompute(init, tfs)
    field, keep = make_field(field_params)
    return dict(init=init, tfs=tfs, w2s=w2s, voxel_J=vJ, voxel_d=vd, field=field, _keep=keep,
                bone_ids=np.asarray(bone_ids, np.int32))


def set_mlp_half_accumulate(on):
    """MLP accumulation mode of the field restatement: False (default) = fp32 accumulators (what the HIP kernels do), True = the
    running sum rounded to half after every 16-wide k block (tcnn v1.6 declares __half wmma accumulators).  Returns the old mode."""
    L = lib()
    old = bool(L.orc_get_mlp_half_accumulate())
    L.orc_set_mlp_half_accumulate(1 if on else 0)
    return old


def render_image_fast(world, rays_o, rays_d, jitter, G=64, **kw):
    """DNeRFModel.render_image_fast (models/DNeRF.py:72-97) for one frame."""
    aabb, density, occ = density_grid_initialize(world, jitter, G)
    o, d, near, far = transform_rays_w2s(rays_o, rays_d, world["w2s"])
    out = render_test(o, d, near, far, occ, aabb, lambda p: deform_query(p, world, True), **kw)
    out.update(occ=occ, aabb=aabb, density=density)
    return out


# ---- SMPLDeformer (deformers/smpl_deformer.py) -------------------------------------------------
def get_bbox_from_smpl(vs, factor=1.2):
    """smpl_deformer.py:7-18"""
    mn, mx = vs.min(0), vs.max(0)
    c = (mx + mn) / 2
    s = ((mx - mn) / 2).max() * np.float32(factor)
    return np.stack([c - s, c + s]).astype(np.float32)


def smpl_deformer_prepare(body, betas, body_pose, global_orient, transl):
    """SMPLDeformer.initialize + prepare_deformer (smpl_deformer.py:32-77): returns T_inv [Vn,4,4],
    vertices in the SMPL-root frame [Vn,3], w2s [4,4], canonical bbox."""
    pose_t = np.zeros((1, 69), np.float32)
    pose_t[:, 2] = np.pi / 6
    pose_t[:, 5] = -np.pi / 6
    st = smpl_forward(body, betas, pose_t)
    so = smpl_forward(body, betas, body_pose, global_orient, transl)
    s2w = so["A"][0].astype(np.float32)
    w2s = np.linalg.inv(s2w).astype(np.float32)
    T_inv = np.linalg.inv(so["T"].astype(np.float32)) @ s2w[None]
    T_inv[:, :3, 3] += st["pose_offsets"] - so["pose_offsets"]
    T_inv[:, :3, 3] += st["shape_offsets"] - so["shape_offsets"]
    T_inv = (st["T"].astype(np.float32) @ T_inv).astype(np.float32)
    verts = (so["vertices"] @ w2s[:3, :3].T + w2s[:3, 3]).astype(np.float32)
    return dict(T_inv=np.ascontiguousarray(T_inv), vertices=np.ascontiguousarray(verts), w2s=w2s,
                bbox=get_bbox_from_smpl(st["vertices"]))


def smpl_nn_deform(pts, verts, T_inv, threshold=0.05):
    """SMPLDeformer.deform -> (pts_cano [P,3], valid [P] bool, idx [P])."""
    pts = _f32(pts).reshape(-1, 3)
    P = len(pts)
    cano = np.empty((P, 3), np.float32); valid = np.empty(P, np.uint8); idx = np.empty(P, np.int32)
    lib().orc_smpl_nn_deform(_p(pts), C.c_long(P), _p(_f32(verts)), C.c_int(len(verts)), _p(_f32(T_inv)),
                             C.c_float(threshold), _p(cano), _p(valid), _p(idx))
    return cano, valid.astype(bool), idx


def smpl_deform_query(pts, prep, field, eval_mode=True, threshold=0.05):
    """SMPLDeformer.deform_test / deform_train (smpl_deformer.py:112-131)."""
    cano, valid, _ = smpl_nn_deform(pts, prep["vertices"], prep["T_inv"], threshold)
    rgb = np.zeros((len(cano), 3), np.float32)
    sigma = np.zeros(len(cano), np.float32) if eval_mode else np.full(len(cano), -1e5, np.float32)
    if valid.any():
        r, s = field_fwd(field, cano[valid])
        rgb[valid], sigma[valid] = r, s
        if not eval_mode:
            bad = ~(np.isfinite(rgb).all(-1) & np.isfinite(sigma))
            rgb[bad] = 0
            sigma[bad] = -1e5
    return rgb, sigma


# ---- a7: implicit differentiation of the roots (deformer_torch.py:50-67,118-128,190-202) ----------
def query_weights(init, xc):
    """ForwardDeformer.query_weights: trilinear sample of lbs_voxel_final [24,D,H,W] at xc [n,3] with
    grid_sample(align_corners=True, padding_mode="border") semantics -> [n,24]."""
    vol = init["lbs_voxel"]
    _, D, H, W = vol.shape
    g = (_f32(xc) + init["offset_kernel"]) * init["scale_kernel"]          # normalised coordinates
    out = np.zeros((len(g), 24), np.float32)
    idx = [np.clip((g[:, a] + 1) / 2 * (n - 1), 0, n - 1) for a, n in ((0, W), (1, H), (2, D))]
    base = [np.floor(i).astype(np.int64) for i in idx]
    frac = [(i - b).astype(np.float32) for i, b in zip(idx, base)]
    for cz in (0, 1):
        for cy in (0, 1):
            for cx in (0, 1):
                xx, yy, zz = base[0] + cx, base[1] + cy, base[2] + cz
                ok = (xx < W) & (yy < H) & (zz < D)
                wt = ((frac[0] if cx else 1 - frac[0]) * (frac[1] if cy else 1 - frac[1]) * (frac[2] if cz else 1 - frac[2]))
                xx, yy, zz = np.minimum(xx, W - 1), np.minimum(yy, H - 1), np.minimum(zz, D - 1)
                out += (wt * ok)[:, None] * vol[:, zz, yy, xx].T
    return out


def implicit_diff_grad(init, xc, J_inv, valid, grad_xc):
    """dL/dtfs [24,4,4] of  x_c* - J_inv (d(x_c*) - sg[d(x_c*)]),  d(x) = sum_n w_n(x) (R_n x + t_n):
    dL/dT_n[c][k] = sum_valid w_n v_c h_k,  v = -J_inv^T dL/dx_c,  h = (x_c*, 1)."""
    xc, g = _f32(xc).reshape(-1, 3), _f32(grad_xc).reshape(-1, 3)
    J = _f32(J_inv).reshape(-1, 3, 3)
    m = np.asarray(valid).reshape(-1).astype(bool)
    w = query_weights(init, xc[m]).astype(np.float64)
    v = -np.einsum("prc,pr->pc", J[m].astype(np.float64), g[m].astype(np.float64))
    h = np.concatenate([xc[m], np.ones((m.sum(), 1), np.float32)], 1).astype(np.float64)
    out = np.zeros((24, 4, 4), np.float64)
    out[:, :3, :] = np.einsum("pn,pc,pk->nck", w, v, h)
    return out.astype(np.float32)


def inverse_skinning(init, xc, xd, valid, tfs, grad_out=None):
    """ForwardDeformer.forward, `version != 1` (deformer_torch.py:68-75): for every valid root x_c* the blended transform
    T = sum_n w_n(x_c*) tfs_n of its skinning weights (query_weights, no gradient: the roots are detached) and
    x_c = (x_d - t)^T R   (row vector times matrix: R^T (x_d - t)); invalid slots 0.
    xc [P,I,3], xd [P,3], valid [P,I], tfs [24,4,4] -> value [P,I,3]; with grad_out [P,I,3] also dL/dtfs [24,4,4]:
    dL/dR[i][j] = (x_d - t)_i g_j, dL/dt[i] = -sum_j R[i][j] g_j, dL/dtfs_n = w_n dL/dT (rows 0..2)."""
    xc = _f32(xc)
    P, I = xc.shape[0], xc.shape[1]
    m = np.asarray(valid).reshape(P, I).astype(bool)
    w = query_weights(init, xc[m]).astype(np.float64)                        # [V,24]
    T = np.einsum("pn,nij->pij", w, _f32(tfs).astype(np.float64))           # [V,4,4]
    a = np.repeat(_f32(xd)[:, None, :], I, axis=1)[m].astype(np.float64) - T[:, :3, 3]
    val = np.zeros((P, I, 3), np.float32)
    val[m] = np.einsum("pi,pij->pj", a, T[:, :3, :3]).astype(np.float32)
    if grad_out is None:
        return val
    g = _f32(grad_out)[m].astype(np.float64)
    dT = np.zeros((len(w), 3, 4), np.float64)
    dT[:, :, :3] = a[:, :, None] * g[:, None, :]
    dT[:, :, 3] = -np.einsum("pij,pj->pi", T[:, :3, :3], g)
    d_tfs = np.zeros((24, 4, 4), np.float64)
    d_tfs[:, :3, :] = np.einsum("pn,pik->nik", w, dT)
    return val, d_tfs.astype(np.float32)


# ---- a15: Raymarcher.render_train (raymarcher_acc.py:140-186) + composite (:25-36) -----------------
def render_train(o, d, near, far, occ, aabb, model, jitter, MAX_SAMPLES=256, bg=None, noise=None):
    """One training render: fixed MAX_SAMPLES slots per ray (raymarch_train), z += jitter * step
    (jitter [N,S] replaces torch.rand_like, :156), masked field evaluation (empty slots sigma = -1e3),
    optional sigma noise [N,S], relu / cumprod compositing.  model(pts) -> (rgb, sigma) in training
    mode.  Returns rgb [N,3], depth [N], alpha [N] (= sum of weights), weights [N,S]."""
    L = lib()
    o = _f32(o).reshape(-1, 3); d = _f32(d).reshape(-1, 3)
    near = _f32(near).reshape(-1); far = _f32(far).reshape(-1)
    N, S = len(o), MAX_SAMPLES
    step = ((far - near) / np.float32(S)).astype(np.float32)
    offset = _f32(aabb[0]); scale = _f32(aabb[1] - aabb[0])
    G = occ.shape[0]
    occ8 = np.ascontiguousarray(occ, np.uint8)
    z = np.zeros((N, S), np.float32)
    L.orc_raymarch_train(_p(o), _p(d), _p(near), _p(far), C.c_long(N), _p(occ8), G, _p(scale), _p(offset), _p(step), S, _p(z))
    mask = z > 0
    z = (z + _f32(jitter).reshape(N, S) * step[:, None]).astype(np.float32)
    pts = (z[..., None] * d[:, None] + o[:, None]).astype(np.float32)
    rgb = np.zeros((N, S, 3), np.float32)
    sig = np.full((N, S), -1e3, np.float32)
    if mask.any():
        r, s_ = model(pts[mask])
        rgb[mask], sig[mask] = r, s_
    if noise is not None:
        sig = sig + _f32(noise).reshape(N, S)
    tau = np.maximum(sig, 0) * step[:, None]
    alpha = (1.0 - np.exp(-tau)).astype(np.float32)
    trans = np.cumprod(np.concatenate([np.ones((N, 1), np.float32), (1 - alpha + np.float32(1e-10))], 1), 1).astype(np.float32)
    w = alpha * trans[:, :-1]
    bgc = 1.0 if bg is None else _f32(bg).reshape(-1, 3)
    color = (w[..., None] * rgb).sum(1) + trans[:, -1:] * bgc
    return dict(rgb=color.astype(np.float32), depth=(w * z).sum(1), alpha=w.sum(1), weights=w, n_field=int(mask.sum()))


def pack_rgba8(rgb, alpha):
    """animate.py:107-113 / novel_view.py:120-125: `img = cat([rgb, alpha[..., None]], -1)`, `(img * 255).astype(np.uint8)` --
    the product in fp32, truncated; values outside [0, 1] clamped first (numpy's out-of-range float -> uint8 cast is
    undefined; the reference's images stay inside but for an ulp).  rgb [...,3], alpha [...] -> uint8 [...,4]."""
    img = np.concatenate([_f32(rgb), _f32(alpha)[..., None]], axis=-1)
    return (np.clip(img, np.float32(0), np.float32(1)) * np.float32(255)).astype(np.uint8)


# ---- optimiser step (DNeRF.py:46-50, :151-159) -----------------------------------------------------------------------
def _fma32(a, b, c):
    """fused multiply-add of float32 arrays: the product of two float32 values is exact in float64; the sum is rounded to
    float64 and then to float32 (a double rounding that can differ from a true FMA only when the float64 sum lands exactly on a
    float32 tie: not observed on the test vectors)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def adam_step(params, grads, state, lrs, betas=(0.9, 0.99), eps=1e-15, skip=False):
    """`self.scaler.step(optimizer)` of DNeRFModel.training_step (DNeRF.py:151-159) for `torch.optim.Adam` (DNeRF.py:46-50),
    restated from torch/optim/adam.py `_single_tensor_adam` (no weight decay, no amsgrad) in float32 with IEEE sqrt / division
    and the two fused multiply-adds of torch's vectorised CPU kernels (lerp, addcmul):
        any gradient element inf / NaN (GradScaler's check)  or  skip  ->  nothing changes, returns True
        step += 1;  m = fma(g - m, 1 - b1, m);  v = fma((1 - b2) g, g, b2 v)
        denom = sqrt(v) / sqrt(1 - b2^step) + eps;   p = p + (-(lr / (1 - b1^step)) m) / denom
    params / grads: lists of float32 arrays (updated in place); state: list of dicts {step, exp_avg, exp_avg_sq}; lrs: one
    learning rate per tensor (the parameter groups).  Returns found_inf."""
    f32 = np.float32
    found = bool(skip) or any(not np.isfinite(g).all() for g in grads)
    if found:
        return True
    b1, b2 = betas
    for p, g, st, lr in zip(params, grads, state, lrs):
        st["step"] = float(st["step"]) + 1.0
        t = st["step"]
        m, v = st["exp_avg"], st["exp_avg_sq"]
        m[...] = _fma32(g - m, f32(1 - b1), m)
        v[...] = _fma32((f32(1 - b2) * g).astype(f32), g, (f32(b2) * v).astype(f32))
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        neg_step = f32(-(float(lr) / bc1))
        denom = ((np.sqrt(v) / f32(bc2 ** 0.5)).astype(f32) + f32(eps)).astype(f32)
        p[...] = (p + ((neg_step * m).astype(f32) / denom).astype(f32)).astype(f32)
    return False

"""Dev tool (CPU, oracle): what 16-bit transform records would do to the image -- voxel_J rounded to fp16 / to 16-bit fixed point, whole
frames re-rendered by the oracle, rays off by more than 1e-3 counted (k_search round 3 (1))."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantavatar_amd import synthetic as syn
from oracle import oracle as orc
body=syn.make_body()
init=orc.deformer_initialize(body, np.zeros(10,np.float32), syn.cano_pose("A_pose"), resolution=128, n_smooth=30)
fp=syn.make_field(init["cano_joints"], init["bbox"])
poses,tr=syn.procedural_pose_track(200)
res=128
ro,rd=syn.make_camera_rays(res)
for f in (37, 120):
    world=orc.make_world(body, init, fp, np.zeros(10,np.float32), poses[f,3:], poses[f,:3], tr[f], syn.INIT_BONES)
    jit=np.random.RandomState(0).rand(5,64**3,3).astype(np.float32)
    ref=orc.render_image_fast(world, ro, rd, jit)
    vJ=world["voxel_J"].copy()
    for name,q in (("fp16", lambda a: a.astype(np.float16).astype(np.float32)),
                   ("fixed16 (rot 2^-15, transl 2.4 m / 2^16)", None)):
        w2=dict(world)
        if q is None:
            J=vJ.copy().reshape(12,-1)   # channel-major [12, D*H*W]?
            shp=vJ.shape
            Jq=vJ.copy()
            # channels: index c in 0..11; translation = c%4==3
            if shp[0]==12:
                for c in range(12):
                    s=(2.4/65536) if c%4==3 else (2.0/65536)
                    Jq[c]=np.round(vJ[c]/s)*s
            else:
                for c in range(12):
                    s=(2.4/65536) if c%4==3 else (2.0/65536)
                    Jq[...,c]=np.round(vJ[...,c]/s)*s
            w2["voxel_J"]=Jq.astype(np.float32)
        else:
            w2["voxel_J"]=q(vJ)
        out=orc.render_image_fast(w2, ro, rd, jit)
        e=np.abs(out["rgb"]-ref["rgb"]).max(1); ea=np.abs(out["alpha"]-ref["alpha"])
        hit=ref["alpha"]>0.01
        print("frame %d %-45s rays |drgb|>1e-3: %.4f of all, %.4f of hit rays; max %.3e; alpha>1e-3: %.4f; occ cells differ %d" % (
            f, name, (e>1e-3).mean(), (e[hit]>1e-3).mean(), e.max(), (ea>1e-3).mean(), int((out["occ"]!=ref["occ"]).sum())))
print("voxel_J shape", vJ.shape)

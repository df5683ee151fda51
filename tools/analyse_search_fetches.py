"""Dev tool (CPU, oracle): distribution of trilinear fetches per Broyden solve for the render and the probe points of one frame --
how many solves are trivial, valid, how many fetches each kind takes (k_search round 3)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantavatar_amd import synthetic as syn
from oracle import oracle as orc
body=syn.make_body()
t0=time.time()
init=orc.deformer_initialize(body, np.zeros(10,np.float32), syn.cano_pose("A_pose"), resolution=128, n_smooth=30)
print("init", time.time()-t0)
fp=syn.make_field(init["cano_joints"], init["bbox"])
poses,tr=syn.procedural_pose_track(200)
f=37
world=orc.make_world(body, init, fp, np.zeros(10,np.float32), poses[f,3:], poses[f,:3], tr[f], syn.INIT_BONES)
res=192
ro,rd=syn.make_camera_rays(res)
jit=np.random.RandomState(0).rand(2,64**3,3).astype(np.float32)
pts_log=[]
aabb,density,occ=orc.density_grid_initialize(world, jit, 64)
o,d,near,far=orc.transform_rays_w2s(ro,rd,world["w2s"])
def model(p):
    pts_log.append(p.copy())
    return orc.deform_query(p, world, True)
out=orc.render_test(o,d,near,far,occ,aabb,model)
pts=np.concatenate(pts_log)
print("render pts", pts.shape, [len(p) for p in pts_log][:12], "alpha cov", (out["alpha"]>0.5).mean())
def stats(P, name):
    x,Ji,valid,iters=orc.broyden(P, world["voxel_J"], world["tfs"], world["init"], world["bone_ids"], want_iters=True)
    it=iters.reshape(-1); v=valid.reshape(-1).astype(bool)
    # trivial solves: initial fetch all OOB -> emulate classification: x0 normalised coords
    T=world["tfs"][list(world["bone_ids"])]  # [13,4,4]
    x0=np.einsum('nji,pnj->pni', T[:,:3,:3], P[:,None,:]-T[None,:,:3,3])
    g=(x0+init["offset_kernel"])*init["scale_kernel"]
    dims=np.array([init["W"],init["H"],init["D"]])
    idx=(g+1)/2*(dims-1)
    f0=np.floor(idx)
    triv=((f0<-1)|(f0>=dims)).any(-1).reshape(-1)
    print("==",name,"points",len(P),"pairs",len(it),"trivial %.3f"%triv.mean(),"valid %.4f"%v.mean())
    nt=~triv
    h=np.bincount(it[nt],minlength=12)
    print(" fetches/solve (non-trivial): mean %.2f"%it[nt].mean(), " hist", (h/h.sum()).round(3)[1:])
    hv=np.bincount(it[nt&v],minlength=12); hi=np.bincount(it[nt&~v],minlength=12)
    print(" valid: mean %.2f share of fetches %.3f"%(it[nt&v].mean(), it[nt&v].sum()/it[nt].sum()), (hv/hv.sum()).round(3)[1:])
    print(" invalid: mean %.2f"%it[nt&~v].mean(), (hi/hi.sum()).round(3)[1:])
    # root displacement in voxels for valid
    xr=x.reshape(-1,3)[v]; x0r=x0.reshape(-1,3)[v]
    dv=np.abs((xr-x0r)*init["scale_kernel"]/2*(dims-1))
    print(" valid roots: |x*-x0| in voxels median", np.median(dv.max(1)).round(3), "90%", np.quantile(dv.max(1),0.9).round(3), "same cell as x0: %.3f"%((np.floor((xr+init["offset_kernel"])*init["scale_kernel"]*0.5*(dims-1)+ (dims-1)/2)==np.floor((x0r+init["offset_kernel"])*init["scale_kernel"]*0.5*(dims-1)+(dims-1)/2)).all(1).mean()))
stats(pts,"render")
G=64
c=np.stack(np.meshgrid(np.arange(G),np.arange(G),np.arange(G),indexing="ij"),-1).reshape(-1,3).astype(np.float32)
probe=((c+jit[0])/G)*(aabb[1]-aabb[0])+aabb[0]
stats(probe.astype(np.float32),"probe")

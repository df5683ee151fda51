"""Dev tool (CPU, numpy re-implementation of the Broyden iteration): cell traces of every solve of a frame's render points and what an LDS
record cache could catch -- same cell as the solve's previous fetch, per-point 1- and 2-entry caches, same-iteration duplicates among a
point's 13 inits (k_search round 3 (3))."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantavatar_amd import synthetic as syn
from oracle import oracle as orc
body=syn.make_body()
init=orc.deformer_initialize(body, np.zeros(10,np.float32), syn.cano_pose("A_pose"), resolution=128, n_smooth=30)
fp=syn.make_field(init["cano_joints"], init["bbox"])
poses,tr=syn.load_animation_track(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'aist_demo_200.npz'))
f=60
world=orc.make_world(body, init, fp, np.zeros(10,np.float32), poses[f,3:], poses[f,:3], tr[f], syn.INIT_BONES)
res=128
ro,rd=syn.make_camera_rays(res)
jit=np.random.RandomState(0).rand(2,64**3,3).astype(np.float32)
pts_log=[]
aabb,density,occ=orc.density_grid_initialize(world, jit, 64)
o,d,near,far=orc.transform_rays_w2s(ro,rd,world["w2s"])
def model(p):
    pts_log.append(p.copy()); return orc.deform_query(p, world, True)
out=orc.render_test(o,d,near,far,occ,aabb,model)
pts=np.concatenate(pts_log)[:60000]
print("pts", pts.shape)
vJ=world["voxel_J"]  # [12,D,H,W]
D,H,W=vJ.shape[1:]
vJc=np.ascontiguousarray(vJ.reshape(12,-1).T)  # [N,12]
off=init["offset_kernel"].astype(np.float64); scl=init["scale_kernel"].astype(np.float64)
T=world["tfs"][list(world["bone_ids"])].astype(np.float64)
P=len(pts); K=13
xd=np.repeat(pts.astype(np.float64)[:,None,:],K,1).reshape(-1,3)
Tb=np.tile(T,(P,1,1))
def fetch(x):
    g=(x+off)*scl
    idx=(g+1)/2*(np.array([W,H,D])-1)
    idx=np.where(np.isfinite(idx), idx, -100.0)
    f0=np.floor(idx).astype(np.int64)
    fr=idx-f0
    J=np.zeros((len(x),12))
    for c in range(8):
        cx=f0[:,0]+(c&1); cy=f0[:,1]+((c>>1)&1); cz=f0[:,2]+(c>>2)
        ok=(cx>=0)&(cx<W)&(cy>=0)&(cy<H)&(cz>=0)&(cz<D)
        w=(fr[:,0] if c&1 else 1-fr[:,0])*(fr[:,1] if c&2 else 1-fr[:,1])*(fr[:,2] if c&4 else 1-fr[:,2])
        lin=(np.clip(cz,0,D-1)*H+np.clip(cy,0,H-1))*W+np.clip(cx,0,W-1)
        J+= (w*ok)[:,None]*vJc[lin]
    cell=(f0[:,2]*4096+f0[:,1])*4096+f0[:,0]
    anyin=((f0[:,0]>=-1)&(f0[:,0]<W)&(f0[:,1]>=-1)&(f0[:,1]<H)&(f0[:,2]>=-1)&(f0[:,2]<D))
    return J, cell, anyin, g
x=np.einsum('nji,nj->ni', Tb[:,:3,:3], xd-Tb[:,:3,3])
J,cell,anyin,g=fetch(x)
active=anyin.copy()   # trivial solves excluded
Ji=np.transpose(J[:,[0,1,2,4,5,6,8,9,10]].reshape(-1,3,3),(0,2,1)).copy()
gx=np.einsum('nij,nj->ni', J[:,[0,1,2,4,5,6,8,9,10]].reshape(-1,3,3), x)+J[:,[3,7,11]]-xd
cells=[np.where(active,cell,-1)]
loaded=[active.copy()]
for it in range(10):
    u=-np.einsum('nij,nj->ni',Ji,gx)
    xn=x+u
    J,cell,anyin,g=fetch(xn)
    gn=np.einsum('nij,nj->ni', J[:,[0,1,2,4,5,6,8,9,10]].reshape(-1,3,3), xn)+J[:,[3,7,11]]-xd
    cells.append(np.where(active,cell,-1)); loaded.append(active&anyin)
    nrm=(gn**2).sum(1)
    done=(nrm<1e-10)|(nrm>1e-2)
    dg=gn-gx
    c=np.einsum('nji,nj->ni',Ji,u)
    s=(c*dg).sum(1)
    r=-np.einsum('nij,nj->ni',Ji,dg)
    with np.errstate(all='ignore'):
        Ji=Ji+((r+u)[:,:,None]*c[:,None,:])/s[:,None,None]
    x=np.where(active[:,None],xn,x); gx=gn
    active=active&~done
cells=np.stack(cells,1)  # [P*K, 11]
loaded=np.stack(loaded,1)
nf=(cells>=0).sum()
print("fetches", nf, "per nontrivial solve", nf/ (cells[:,0]>=0).sum(), "loaded", loaded.sum())
# (a) same cell as the lane's previous fetch
same_prev=((cells[:,1:]==cells[:,:-1])&(cells[:,1:]>=0)).sum()
print("same cell as own previous fetch: %.3f of fetches"%(same_prev/nf))
# (b) per-point 1-entry cache, lockstep over iterations, all 13 inits of a point together
cp=cells.reshape(P,K,11)
tag=np.full(P,-2,np.int64); hits=0
for it in range(11):
    c=cp[:,:,it]
    hit=(c==tag[:,None])&(c>=0)
    hits+=hit.sum()
    miss=(c>=0)&~hit
    # winner: first missing init
    first=np.argmax(miss,1); has=miss.any(1)
    tag=np.where(has, c[np.arange(P),first], tag)
print("per-point 1-entry cache (lockstep): hit %.3f of fetches"%(hits/nf))
# (c) per-point 2-entry cache (LRU-ish: replace oldest)
tags=np.full((P,2),-2,np.int64); age=np.zeros((P,2),np.int64); hits=0
for it in range(11):
    c=cp[:,:,it]
    hit=((c[:,:,None]==tags[:,None,:]).any(2))&(c>=0)
    hits+=hit.sum()
    miss=(c>=0)&~hit
    first=np.argmax(miss,1); has=miss.any(1)
    slot=np.argmin(age,1)
    newtag=c[np.arange(P),first]
    tags[np.arange(P)[has],slot[has]]=newtag[has]; age[np.arange(P)[has],slot[has]]=it+1
print("per-point 2-entry cache: hit %.3f"%(hits/nf))
# (d) within-iteration sharing: fetches whose cell equals another init's cell of the same point in the same iteration (dedupe potential)
dup=0
for it in range(11):
    c=cp[:,:,it]
    for p0 in range(0,P,20000):
        cc=np.sort(c[p0:p0+20000],1)
        dup+=((cc[:,1:]==cc[:,:-1])&(cc[:,1:]>=0)).sum()
print("same-iteration duplicates among a point's inits: %.3f of fetches"%(dup/nf))

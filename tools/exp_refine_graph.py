"""Dev tool: which part of the refine step survives HIP-graph capture?  Every stage runs in its own process (a failed
capture can take the process down)."""
import subprocess
import sys

STAGES = {
    "smpl_fwd_bwd": """
import torch, numpy as np
from instantavatar_amd import synthetic as syn
from instantavatar_amd.deformers.smplx import SMPL
from instantavatar_amd.deformers.snarf_deformer import affine_inverse
dev='cuda:0'
smpl=SMPL.from_dict(syn.make_body(42)).to(dev)
emb=torch.nn.Embedding.from_pretrained(torch.randn(4,69,device=dev)*0.2, freeze=False)
idx=torch.tensor([1],device=dev)
def step():
    bp=emb(idx)
    out=smpl(betas=torch.zeros(1,10,device=dev), body_pose=bp, global_orient=torch.zeros(1,3,device=dev)+0.1, transl=torch.ones(1,3,device=dev), return_verts=False)
    w2s=affine_inverse(out.A[:,0].float())
    tfs=w2s[:,None]@out.A
    emb.weight.grad=None
    (tfs*tfs).sum().backward()
    return tfs
s=torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t=step()
g.replay(); torch.cuda.synchronize()
print('OK', float(t.sum()), float(emb.weight.grad.abs().sum()))
""",
    "refine_step": """
import sys
sys.path.insert(0, 'tests')
import torch, numpy as np
import test_gpu_refine as T
from instantavatar_amd.training import GraphedTrainStep, training_step
model, opt, loss_fn = T._setup()
training_step(model, T._batch(0), opt, loss_fn, is_refine=True)
st = GraphedTrainStep(model, opt, loss_fn, is_refine=True)
for it in range(6):
    b = T._batch(it % 3); b['idx_dev'] = torch.tensor([it % 3], device='cuda:0')
    out = st(b)
torch.cuda.synchronize()
print('OK replays', st.replays, 'eager', st.eager_steps, 'err', st.capture_error, float(out['loss']))
""",
}
for name in (sys.argv[1:] or list(STAGES)):
    r = subprocess.run([sys.executable, "-c", STAGES[name]], capture_output=True, text=True)
    print("==", name, "rc", r.returncode)
    print(r.stdout[-600:])
    print("\n".join(l for l in r.stderr.splitlines() if "Warning" not in l and "warn" not in l)[-1500:])

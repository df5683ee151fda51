"""SMPLParamEmbedding (drop-in for instant_avatar/models/structures/body_model_param.py): per-frame SMPL parameters as
trainable embeddings, optimised together with the field in the `fit.py` stage (SNARF_NGP_fitting.yaml:
optimize_SMPL.enable) and in pose refinement (SNARF_NGP_refine.yaml)."""
import torch
import torch.nn as nn


class SMPLParamEmbedding(nn.Module):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        for k, v in kwargs.items():  # betas [1,10], global_orient [N,3], transl [N,3], body_pose [N,69]
            setattr(self, k, nn.Embedding.from_pretrained(torch.as_tensor(v).float(), freeze=False))
        self.keys = ["betas", "global_orient", "transl", "body_pose"]

    def forward(self, idx):
        return {
            "betas": self.betas(torch.zeros_like(idx)),
            "body_pose": self.body_pose(idx),
            "global_orient": self.global_orient(idx),
            "transl": self.transl(idx),
        }

    def tv_loss(self, idx):
        """temporal smoothness of the per-frame parameters (body_model_param.py:23-35; the reference iterates
        `self.items()`, which nn.Module does not have -- the intent, a sum over the three per-frame tables, is kept)"""
        loss = 0
        N = len(self.global_orient.weight)
        idx_p = (idx - 1).clip(min=0)
        idx_n = (idx + 1).clip(max=N - 1)
        for k in ("global_orient", "transl", "body_pose"):
            v = getattr(self, k)
            loss = loss + (v(idx) - v(idx_p)).square().mean()
            loss = loss + (v(idx_n) - v(idx)).square().mean()
        return loss

    def export(self):
        """{key: numpy array} of the optimised tables, what fit.py writes to poses/train.npz (fit.py:49-52)"""
        return {k: getattr(self, k).weight.detach().cpu().numpy().copy() for k in self.keys}

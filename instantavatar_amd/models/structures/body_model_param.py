"""SMPLParamEmbedding (drop-in for instant_avatar/models/structures/body_model_param.py): per-frame SMPL parameters as
trainable embedding tables, optimised together with the field in the `fit.py` stage (SNARF_NGP_fitting.yaml:
optimize_SMPL.enable) and in pose refinement (SNARF_NGP_refine.yaml).

Surface kept from the reference: construction from keyword tables, attribute names `betas / global_orient / transl /
body_pose` (their `.weight` is what checkpoints and `configure_optimizers` see), `forward(idx)` returning the four
entries of the batch, `tv_loss(idx)`, plus `export()` for what fit.py writes to `poses/train.npz`."""
import torch
import torch.nn as nn

_PER_FRAME = ("global_orient", "transl", "body_pose")     # one row per frame; `betas` has a single row shared by all frames


class SMPLParamEmbedding(nn.Module):
    keys = ["betas", "global_orient", "transl", "body_pose"]

    def __init__(self, **tables) -> None:
        super().__init__()
        for name, init in tables.items():  # betas [1,10], global_orient [N,3], transl [N,3], body_pose [N,69]
            table = nn.Embedding.from_pretrained(torch.as_tensor(init).float().contiguous().clone(), freeze=False)   # (its own dense storage: callers hand over views of larger tables)
            self.add_module(name, table)

    def forward(self, idx):
        rows = {name: getattr(self, name)(idx) for name in _PER_FRAME}
        rows["betas"] = self.betas(idx * 0)          # every frame reads row 0 (body_model_param.py:17)
        return rows

    def tv_loss(self, idx):
        """Temporal smoothness of the per-frame tables: squared difference to the previous and to the next frame, clamped
        at the ends of the sequence (body_model_param.py:23-35; the reference iterates `self.items()`, which nn.Module does
        not have -- the intent, a sum over the three per-frame tables, is kept)."""
        last = self.global_orient.weight.shape[0] - 1
        before, after = (idx - 1).clamp(min=0), (idx + 1).clamp(max=last)
        total = 0
        for name in _PER_FRAME:
            table = getattr(self, name)
            here = table(idx)
            total = total + (here - table(before)).square().mean() + (table(after) - here).square().mean()
        return total

    def export(self):
        """{key: numpy array} of the optimised tables, what fit.py writes to poses/train.npz (fit.py:49-52)"""
        return {name: getattr(self, name).weight.detach().cpu().numpy().copy() for name in self.keys}

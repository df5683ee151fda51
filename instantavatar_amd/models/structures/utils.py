"""Rays container -- same fields as instant_avatar/models/structures/utils.py:5-10."""
from dataclasses import dataclass

import torch


@dataclass
class Rays:
    o: torch.Tensor  # (N, 3)
    d: torch.Tensor  # (N, 3)

    near: torch.Tensor = None  # (N, )
    far: torch.Tensor = None  # (N, )

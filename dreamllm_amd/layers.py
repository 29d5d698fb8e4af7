"""nn.Module shells over the HIP operators that keep torch.nn's parameter names (state_dict compatibility)."""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps)

"""Projector registry -- API mirror of omni/models/projector/{builder,base_projector,mlp_projector}.py on HIP kernels.

`build_projector(cfg_dict, in_hidden_size, out_hidden_size, bias)` returns a module whose forward takes a tensor or a
list and RETURNS A LIST (callers take `[-1]`), honours `freeze_projector` through `torch.set_grad_enabled`, and keeps the
reference's state_dict keys (`projector.weight/bias`, `projector.{0,2,..}.weight`).  conv/sam projectors are not used by
any dreamllm config (projects/dreamllm/configs/common.py:28,46) and are out of scope.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .utils import check_path_and_file, logger


class HipLinear(nn.Linear):
    """nn.Linear parameters, HIP MFMA GEMM arithmetic (fwd, dgrad, wgrad).  Same state_dict keys as nn.Linear."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class HipGELU(nn.GELU):
    def forward(self, x):
        return ops.gelu(x)


class BaseProjector(nn.Module):
    """omni/models/projector/base_projector.py:8-36."""

    def load_model(self, model_name_or_path=None):
        if model_name_or_path is not None:
            if check_path_and_file(model_name_or_path, f"{self.save_model_name}_projector.bin"):
                logger.info(f"loading `BaseProjector` from {model_name_or_path}...")
                self.load_state_dict(torch.load(model_name_or_path, map_location="cpu"))
                return True
            # older checkpoints: the bare projector's state dict under a `.pt` name
            if check_path_and_file(model_name_or_path, f"{self.save_model_name}_projector.pt"):
                logger.info(f"loading `BaseProjector` from {model_name_or_path}...")
                self.projector.load_state_dict(torch.load(model_name_or_path, map_location="cpu"))
                return True
        return False

    def forward(self, features) -> list:
        # subclasses return a LIST of feature tensors (one per path), as their callers index it
        pass

    @property
    def save_model_name(self):
        return self.args.save_model_name + "_projector"

    @property
    def dtype(self):
        return next(self.projector.parameters()).dtype

    @property
    def device(self):
        return next(self.projector.parameters()).device


class LinearProjector(BaseProjector):
    """omni/models/projector/mlp_projector.py:11-27."""

    def __init__(self, args, in_hidden_size, out_hidden_size, bias=True):
        super().__init__()
        self.args = args
        self.freeze_projector = args.freeze_projector
        self.depth = args.depth
        assert self.depth == 1, "LinearProjector now only supports depth=1"
        assert bias is not None, "bias should be set as True or False"
        self.projector = HipLinear(in_hidden_size, out_hidden_size, bias=bias)

    def forward(self, features):
        if not isinstance(features, list):
            features = [features]
        with torch.set_grad_enabled(not self.freeze_projector and torch.is_grad_enabled()):
            return [self.projector(feature) for feature in features]


class MLPProjector(BaseProjector):
    """omni/models/projector/mlp_projector.py:30-50: Linear (GELU Linear) x (depth-1)."""

    def __init__(self, args, in_hidden_size, out_hidden_size, bias=False):
        super().__init__()
        self.args = args
        self.freeze_projector = args.freeze_projector
        self.depth = args.depth
        assert self.depth > 1, "MLPProjector now only supports depth > 1, use linear if depth is 1"
        assert bias is not None, "bias should be set as True or False"
        modules = [HipLinear(in_hidden_size, out_hidden_size, bias=bias)]
        for _ in range(1, self.depth):
            modules.append(HipGELU())
            modules.append(HipLinear(out_hidden_size, out_hidden_size, bias=bias))
        self.projector = nn.Sequential(*modules)

    def forward(self, features):
        if not isinstance(features, list):
            features = [features]
        with torch.set_grad_enabled(not self.freeze_projector and torch.is_grad_enabled()):
            return [self.projector(feature) for feature in features]


def build_projector(projector_cfg, in_hidden_size, out_hidden_size, bias=None):
    """omni/models/projector/builder.py:9-22."""
    projector_cfg = SimpleNamespace(**projector_cfg)
    projector = getattr(projector_cfg, "projector", None)
    logger.info(f"Building projector ({projector_cfg.save_model_name}): {projector}")
    if projector == "linear":
        return LinearProjector(args=projector_cfg, in_hidden_size=in_hidden_size, out_hidden_size=out_hidden_size, bias=bias)
    if projector == "mlp":
        return MLPProjector(args=projector_cfg, in_hidden_size=in_hidden_size, out_hidden_size=out_hidden_size, bias=bias)
    if projector in ("conv", "sam"):
        raise NotImplementedError(
            f"projector `{projector}` is outside the accelerated hot path (unused by projects/dreamllm configs)")
    raise ValueError(f"Unknown projector: {projector} (supported: linear, mlp, conv, sam)")

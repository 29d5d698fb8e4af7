"""Small host-side helpers the plugins rely on (mirrors of omni/utils/{fsdp_utils,modeling_utils,misc,torch_utils}.py)."""
from __future__ import annotations

import importlib
import itertools
import logging
import os
from collections.abc import Mapping

import torch

logger = logging.getLogger("dreamllm_amd")


class FSDPMixin:
    """omni/utils/fsdp_utils.py:18-20."""

    def fsdp_ignored_modules(self) -> list:
        return []


def get_model_device(model: torch.nn.Module):
    """omni/utils/modeling_utils.py:76."""
    return next(itertools.chain(model.parameters(), model.buffers())).device


def get_model_dtype(model: torch.nn.Module):
    """omni/utils/modeling_utils.py:92: dtype of the first floating parameter/buffer."""
    for t in itertools.chain(model.parameters(), model.buffers()):
        if t.is_floating_point():
            return t.dtype
    return torch.float32


def check_path_and_file(path, file):
    """omni/utils/misc.py:226."""
    if path is not None and os.path.isdir(path):
        return os.path.isfile(os.path.join(path, file))
    return False


class _TorchDraws:
    """Where the plugin heads get their random numbers (`torch.randn / randn_like / randint / bernoulli`, called exactly
    where the reference calls them: modeling_plugins.py:511,520-528,541).  One indirection so that tests can replay the
    draws recorded from an execution of the reference wrapper (`replay_draws`)."""

    def randn(self, shape, generator=None, device=None, dtype=None):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    def randn_like(self, x):
        return torch.randn_like(x)

    def randint(self, low, high, shape, device=None):
        return torch.randint(low, high, shape, device=device)

    def bernoulli(self, p):
        return torch.bernoulli(p)


class _ReplayDraws:
    """Returns pre-recorded draws in order (kind and shape are checked); see oracle/duck_diffusers.DrawLog."""

    def __init__(self, draws):
        self._q = list(draws)

    def _next(self, kind, shape, device, dtype):
        if not self._q:
            raise AssertionError(f"replay_draws: no recorded draw left for {kind}{tuple(shape)}")
        k, t = self._q.pop(0)
        if k != kind or tuple(t.shape) != tuple(shape):
            raise AssertionError(f"replay_draws: expected {k}{tuple(t.shape)}, the head asked for {kind}{tuple(shape)}")
        return t.to(device=device, dtype=dtype if dtype is not None else t.dtype)

    def randn(self, shape, generator=None, device=None, dtype=None):
        return self._next("randn", shape, device, dtype)

    def randn_like(self, x):
        return self._next("randn_like", x.shape, x.device, x.dtype)

    def randint(self, low, high, shape, device=None):
        return self._next("randint", shape, device, torch.int64)

    def bernoulli(self, p):
        return self._next("bernoulli", p.shape, p.device, p.dtype)

    def remaining(self):
        return len(self._q)


draws = _TorchDraws()


class replay_draws:
    """`with replay_draws(recorded): head(...)` -- the heads consume `recorded` instead of fresh random numbers."""

    def __init__(self, recorded):
        self._src = _ReplayDraws(recorded)

    def __enter__(self):
        global draws
        self._prev = draws
        draws = self._src
        return self._src

    def __exit__(self, *exc):
        global draws
        draws = self._prev
        if exc[0] is None and self._src.remaining():
            raise AssertionError(f"replay_draws: {self._src.remaining()} recorded draws were not consumed")
        return False


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """omni/utils/torch_utils.py:7-52: sample on the generator's device (CPU generator => CPU sample, then moved) so a
    seed reproduces the same latents on every backend."""
    device = device or torch.device("cpu")
    if isinstance(generator, list):
        shape1 = (1,) + tuple(shape[1:])
        lat = [randn_tensor(shape1, g, device, dtype) for g in generator]
        return torch.cat(lat, 0)
    gdev = generator.device.type if generator is not None else torch.device(device).type
    if gdev == "cpu":
        return draws.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
    return draws.randn(shape, generator=generator, device=device, dtype=dtype)


def locate(name: str):
    """omni/config/registry.py:30: resolve a dotted path to an object."""
    parts = name.split(".")
    for i in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ImportError:
            continue
        for p in parts[i:]:
            obj = getattr(obj, p)
        return obj
    raise ImportError(f"cannot locate {name}")


def target_to_string(t) -> str:
    return t if isinstance(t, str) else f"{t.__module__}.{t.__qualname__}"


def deep_instantiate(cfg):
    """omni/config/instantiate.py:86-136 on plain dicts/lists: build objects from `_target_` dotted paths."""
    if isinstance(cfg, list):
        return [deep_instantiate(x) for x in cfg]
    if isinstance(cfg, Mapping):
        if "_target_" in cfg:
            kw = {k: deep_instantiate(v) for k, v in cfg.items()}
            cls = kw.pop("_target_")
            if isinstance(cls, str):
                cls = locate(cls)
            assert callable(cls), f"_target_ {cls} does not define a callable object"
            return cls(**kw)
        return {k: deep_instantiate(v) for k, v in cfg.items()}
    return cfg


def average_init_token_embeddings(model, num_added_tokens: int):
    """omni/utils/tokenizer_utils.py:70-80: newly added token rows of the input and output embeddings start at the mean of the
    existing rows (train.py:142-146)."""
    assert num_added_tokens > 0, "`num_added_tokens` should be positive"
    input_embeddings = model.get_input_embeddings().weight.data
    output_embeddings = model.get_output_embeddings().weight.data
    input_embeddings[-num_added_tokens:] = input_embeddings[:-num_added_tokens].mean(dim=0, keepdim=True)
    output_embeddings[-num_added_tokens:] = output_embeddings[:-num_added_tokens].mean(dim=0, keepdim=True)

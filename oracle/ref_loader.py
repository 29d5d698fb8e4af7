"""TEST INFRASTRUCTURE -- imports the *real* reference (RunpeiDong/DreamLLM, /root/reference) on CPU.

Only usable in the authoring container (the GPU box has no /root/reference).  Used by `oracle/make_golden.py` to pin the
restated oracle (`oracle/llm_ref.py`, ...) and to generate the golden vectors committed under `tests/golden/`.
Nothing under dreamllm_amd/ may import this module.

The reference is pure Python but depends on packages that are not installed here (loguru, omegaconf, diffusers, ...).
Those are stubbed through a meta-path finder; `omni.models.dreamllm.modeling_plugins` (diffusers/torchvision at import
time, modeling_plugins.py:10-14) is replaced by a stub exposing the two names modeling_dreamllm.py needs.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DREAMLLM_REFERENCE", "/root/reference")

_STUB_ROOTS = {
    "loguru", "omegaconf", "pendulum", "black", "isort", "pyinstrument", "dacite", "colorama", "hydra", "megfile",
    "boto3", "smart_open", "diffusers", "torchvision", "wandb", "webdataset", "deepspeed", "peft", "xformers",
    "flash_attn", "cv2", "decord", "imageio", "termcolor", "tabulate_stub",
}


class _AnyAttr(types.ModuleType):
    """Module whose every attribute is a permissive dummy (callable, subscriptable, usable as a decorator/base)."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        d = _Dummy(f"{self.__name__}.{name}")
        setattr(self, name, d)
        return d


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy(f"{cls.__name__}.{name}")


class _Dummy(metaclass=_DummyMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # decorator use
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __getitem__(self, k):
        return _Dummy()

    def __or__(self, o):
        return self

    def __ror__(self, o):
        return self

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _AnyAttr(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


_installed = False


def install():
    """Make `import omni.models.dreamllm.modeling_dreamllm` work on CPU in this container."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise FileNotFoundError(f"{REFERENCE_ROOT} not present (only the authoring container has the reference)")
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # loguru.logger used at import time
    loguru = importlib.import_module("loguru")
    loguru.logger = _Logger()
    # the reference's own loguru wrapper subclasses loguru internals (omni/utils/loguru.py:13-41): stub it whole
    lg = types.ModuleType("omni.utils.loguru")
    lg.logger = _Logger()
    sys.modules["omni.utils.loguru"] = lg
    # plugin module stub (PluginType / PipelineImageType only)
    stub = types.ModuleType("omni.models.dreamllm.modeling_plugins")
    stub.PluginType = str
    stub.PipelineImageType = object
    sys.modules["omni.models.dreamllm.modeling_plugins"] = stub
    _installed = True


def load_modeling():
    install()
    mod = importlib.import_module("omni.models.dreamllm.modeling_dreamllm")
    return mod


def load_projector():
    install()
    return importlib.import_module("omni.models.projector.builder")


def make_config(**kw):
    install()
    cfgmod = importlib.import_module("omni.models.dreamllm.configuration_dreamllm")
    cfg = cfgmod.DreamLLMConfig(**kw)
    cfg.rope_scaling = None  # transformers 5.x auto-fills it; modeling_dreamllm.py:287 expects None or {"type","factor"}
    return cfg

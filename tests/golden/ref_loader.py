"""Import the *reference* implementation on CPU (build container only).

Used by ``gen_golden.py`` and by the optional ``-m "not gpu"`` cross-checks that
skip when ``/root/reference`` is absent (it never exists on the GPU box).  The
reference source is imported from where it lies; nothing is copied.

Third-party packages the image lacks are replaced by minimal build-owned
stand-ins *for import purposes only*:
  natten   -> NeighborhoodAttention1D running oracle.pluto_ref.neighborhood_attention_1d
              (published natten 0.14.6 algorithm; parity unpinned, see oracle/__init__.py)
  timm     -> DropPath (identity when p == 0 or eval)
  lightning-> LightningModule = nn.Module with no-op log/save_hyperparameters
  cv2, carla, shapely, numba, omegaconf ... -> empty modules (never called on this path)
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("RIFT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "rift", "cbv", "planning"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _NA1D(nn.Module):
    """Parameter layout of natten 0.14.6 NeighborhoodAttention1D: qkv, rpb, proj."""

    def __init__(self, dim, num_heads, kernel_size, dilation=None, bias=True, qkv_bias=True,
                 qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert dilation in (None, 1) and qk_scale is None and attn_drop == 0.0 and proj_drop == 0.0
        self.num_heads, self.kernel_size = num_heads, kernel_size
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.rpb = nn.Parameter(torch.zeros(num_heads, 2 * kernel_size - 1))
        nn.init.trunc_normal_(self.rpb, std=0.02, mean=0.0, a=-2.0, b=2.0)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        from oracle.pluto_ref import SD, neighborhood_attention_1d
        sd = SD({"qkv.weight": self.qkv.weight, "qkv.bias": self.qkv.bias, "rpb": self.rpb,
                 "proj.weight": self.proj.weight, "proj.bias": self.proj.bias})
        return neighborhood_attention_1d(x, sd, self.num_heads, self.kernel_size)


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class _LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    assert available(), f"reference checkout not found at {REF_ROOT}"
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for p in (repo, REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    _stub("natten", NeighborhoodAttention1D=_NA1D)
    timm = _stub("timm")
    timm.layers = _stub("timm.layers", DropPath=_DropPath)
    L = _stub("lightning", LightningModule=_LightningModule, LightningDataModule=object)
    L.pytorch = _stub("lightning.pytorch")
    L.pytorch.utilities = _stub("lightning.pytorch.utilities")
    L.pytorch.utilities.types = _stub("lightning.pytorch.utilities.types",
                                      EVAL_DATALOADERS=object, TRAIN_DATALOADERS=object)
    _stub("omegaconf", DictConfig=dict)
    for name in ("cv2",):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                _stub(name)
    # the package __init__ of rift.cbv.planning imports carla-bound policies: register
    # the package shells ourselves so that only the sub-modules we ask for are executed
    for pkg in ("rift", "rift.cbv", "rift.cbv.planning"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    _installed = True


def planning_model(**kw):
    install()
    from rift.cbv.planning.pluto.model.pluto_model import PlanningModel
    return PlanningModel(radius=120, **kw)


def rift_trainer_module():
    install()
    return importlib.import_module("rift.cbv.planning.fine_tuner.rlft.rift_pluto.rift_trainer")

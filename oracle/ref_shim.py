"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference modules.

Imports ``/root/reference/repo`` on CPU behind thin import shims so that the
reference's own ``TargetDiff`` / ``UniTransformer`` code can be executed in the
build container.  It exists to (a) validate ``oracle/`` (the restatement) and
(b) generate the golden vectors committed under ``tests/golden/``
(``oracle/make_golden.py``).  ``/root/reference`` does not exist on the GPU box,
so nothing under ``tests/ -m gpu``, ``bench.py`` or ``smoke()`` imports this.

Shims:
  * MagicMock modules for packages that are not installed and are not on the
    hot path: rdkit, torch_geometric, easydict, lmdb, Bio, EFGs, openbabel,
    vina, meeko, torch_cluster, tensorboard.
  * ``torch_scatter``: pure-torch scatter_sum/add/mean/softmax with the
    published torch_scatter semantics (README.MD:57-58 installs it unpinned).
  * ``knn_graph`` (torch_cluster via torch_geometric.nn, unitransformer.py:11,80):
    replaced by ``oracle.unitransformer.knn_graph`` (exact fp32 squared
    distances, ties by index).
  * ``torch.Any`` alias (molecule_featurizer.py:155 uses it as an annotation).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
import typing
from unittest import mock

import torch

REFERENCE_ROOT = os.environ.get("CBGX_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = (
    "rdkit", "torch_geometric", "easydict", "lmdb", "Bio", "EFGs", "openbabel",
    "vina", "meeko", "torch_cluster", "tensorboard", "plip", "AutoDockTools",
    "pdb2pqr", "prody",
)


class AttrDict(dict):
    """Minimal stand-in for easydict.EasyDict (attribute access + nested)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


def _make_torch_scatter():
    m = types.ModuleType("torch_scatter")

    def _expand(index, src, dim):
        if dim < 0:
            dim = src.dim() + dim
        shape = [1] * src.dim()
        shape[dim] = -1
        return index.view(shape).expand_as(src), dim

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        idx, dim = _expand(index, src, dim)
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        shape = list(src.shape)
        shape[dim] = dim_size
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        return res.scatter_add_(dim, idx, src)

    def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
        s = scatter_sum(src, index, dim, None, dim_size)
        ones = torch.ones(index.shape, dtype=src.dtype, device=src.device)
        cnt = scatter_sum(ones, index, 0, None, s.shape[dim if dim >= 0 else src.dim() + dim])
        cnt = cnt.clamp(min=1)
        shape = [1] * s.dim()
        shape[dim if dim >= 0 else src.dim() + dim] = -1
        return s / cnt.view(shape)

    def scatter_softmax(src, index, dim=-1, dim_size=None):
        idx, dim = _expand(index, src, dim)
        if dim_size is None:
            dim_size = int(index.max()) + 1
        shape = list(src.shape)
        shape[dim] = dim_size
        mx = torch.full(shape, float("-inf"), dtype=src.dtype)
        mx = mx.scatter_reduce(dim, idx, src, reduce="amax", include_self=True)
        ex = (src - mx.gather(dim, idx)).exp()
        den = torch.zeros(shape, dtype=src.dtype).scatter_add_(dim, idx, ex)
        return ex / den.gather(dim, idx)

    def scatter_max(*a, **k):
        raise NotImplementedError("scatter_max is not on the diffusion hot path")

    m.scatter_sum = scatter_sum
    m.scatter_add = scatter_sum
    m.scatter_mean = scatter_mean
    m.scatter_softmax = scatter_softmax
    m.scatter_max = scatter_max
    m.scatter = scatter_sum
    return m


_LOADED = None


def load_reference():
    """Import the reference package; returns the ``repo.models`` module."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "repo")):
        raise FileNotFoundError(f"reference tree not found at {REFERENCE_ROOT}")
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    sys.modules.setdefault("torch_scatter", _make_torch_scatter())
    ed = types.ModuleType("easydict")
    ed.EasyDict = AttrDict
    sys.modules["easydict"] = ed
    if not hasattr(torch, "Any"):
        torch.Any = typing.Any
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import repo.models as M  # must enter via repo.models (circular import otherwise)
    import repo.modules.e3nn.unitransformer as U
    import repo.models.diffusion.diffbp as DB
    from oracle.unitransformer import knn_graph as _knn

    def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", **kw):
        assert flow == "source_to_target" and not loop
        return _knn(x, batch, k)

    from oracle.diffbp import knn_cross as _knn_cross

    def knn(x, y, k, batch_x=None, batch_y=None, **kw):
        """torch_cluster.knn as called by interior_loss (diffbp.py:19): for every y the k nearest x of the same graph;
        row 0 = y index, row 1 = x index"""
        return _knn_cross(x, y, batch_x, batch_y, k)

    U.knn_graph = knn_graph
    DB.knn_graph = knn_graph
    DB.knn = knn
    _LOADED = M
    return M


def targetdiff_config(num_atomtype=13, num_layers=9, **encoder_overrides):
    """Model block of configs/denovo/train/targetdiff.yml:1-23 (+ num_atomtype)."""
    enc = dict(type="unitransformer", node_feat_dim=128, n_heads=16, num_layers=num_layers)
    enc.update(encoder_overrides)
    return AttrDict(
        type="targetdiff",
        num_atomtype=num_atomtype,
        encoder=enc,
        generator=dict(
            pos_schedule=dict(type="sigmoid", beta_start=1.0e-7, beta_end=2.0e-3),
            atom_schedule=dict(type="cosine", cosine_s=0.01),
            num_diffusion_timesteps=1000,
            time_sampler="symmetric",
        ),
        embedder=dict(emb_dim=128, atom=dict(type="linear"), residue=dict(type="linear")),
    )


def build_reference_targetdiff(num_atomtype=13, num_layers=9, seed=0):
    M = load_reference()
    torch.manual_seed(seed)
    model = M.get_model(targetdiff_config(num_atomtype, num_layers)).eval()
    return model

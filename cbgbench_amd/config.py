"""Config surface of the reference: YAML with ``!include`` -> attribute dict, plus the derived
``num_atomtype`` / ``mode`` fields.

Mirrors ``load_config`` (repo/utils/misc.py:109-145) and ``set_num_atom_type``
(repo/utils/configuration.py:13-38) so the reference's ``configs/{task}/{train,test}/{method}.yml``
files load unchanged.  ``easydict`` is not installed here; ``Config`` provides the same attribute
access and ``.get``.
"""
import json
import os

import yaml


class Config(dict):
    """dict with attribute access, recursively (stand-in for easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


def to_plain(v):
    """nested plain dict / list copy of a Config (what goes into checkpoints: unpicklable without this package otherwise)"""
    if isinstance(v, dict):
        return {k: to_plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(to_plain(x) for x in v)
    return v


def checkpoint_config(config):
    """the `config` entry of a checkpoint in a form the reference's scripts can unpickle: an easydict.EasyDict when
    easydict is importable (what the reference's train.py:266-273 stores), else plain nested dicts"""
    plain = to_plain(config)
    try:
        from easydict import EasyDict
    except ImportError:
        return plain
    return EasyDict(plain)


def load_checkpoint_file(path, map_location="cpu"):
    """torch.load of a checkpoint written by this package OR by the reference (whose `config` is a pickled
    easydict.EasyDict: when easydict is not installed a stand-in module mapping EasyDict -> Config is registered for the
    duration of the load).  `config`, when present, comes back as a Config."""
    import sys
    import types

    import torch
    shim = None
    if "easydict" not in sys.modules:
        try:
            import easydict  # noqa: F401
        except ImportError:
            shim = types.ModuleType("easydict")
            shim.EasyDict = Config
            sys.modules["easydict"] = shim
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    finally:
        if shim is not None and sys.modules.get("easydict") is shim:
            del sys.modules["easydict"]
    if isinstance(ckpt, dict) and isinstance(ckpt.get("config"), dict):
        ckpt["config"] = Config(to_plain(ckpt["config"]))
    return ckpt


class _Loader(yaml.SafeLoader):
    def __init__(self, stream):
        self._root = os.path.split(getattr(stream, "name", os.path.curdir + os.sep))[0]
        super().__init__(stream)


def _include(loader, node):
    filename = os.path.abspath(os.path.join(loader._root, loader.construct_scalar(node)))
    ext = os.path.splitext(filename)[1].lstrip(".")
    with open(filename, "r") as f:
        if ext in ("yaml", "yml"):
            return yaml.load(f, _Loader)
        if ext == "json":
            return json.load(f)
        return f.read()


yaml.add_constructor("!include", _include, _Loader)


def load_config(config_path):
    """-> (Config, config_name); same contract as repo/utils/misc.py:141-145."""
    with open(config_path, "r") as f:
        config = Config(yaml.load(f, _Loader))
    name = os.path.basename(config_path)
    return config, name[: name.rfind(".")]


# atom-type vocabulary sizes: len(map_atom_type_only_to_index) = 8 (repo/utils/molecule/constants.py:54-63),
# len(map_atom_type_aromatic_to_index) = 13 (:65-79).
NUM_ATOM_TYPES = {"basic": 8, "add_aromatic": 13}
# what an atom-type index stands for (the inverse of those two maps): atomic number, and for 'add_aromatic' the aromatic flag
# -- the ``atom`` / ``aromatic`` fields sample.py:29-30 hands to the reconstruction step
_ATOMIC_NUMBER = {"basic": (1, 6, 7, 8, 9, 15, 16, 17),
                  "add_aromatic": (1, 6, 6, 7, 7, 8, 8, 9, 15, 15, 16, 16, 17)}
_AROMATIC = {"add_aromatic": (False, False, True, False, True, False, True, False, False, True, False, True, False)}


def get_atomic_number_from_index(index, mode):
    """repo/utils/molecule/constants.py:85-94 for the two vocabularies the diffusion configs use"""
    if mode not in _ATOMIC_NUMBER:
        raise ValueError(mode)
    return [_ATOMIC_NUMBER[mode][int(i)] for i in index]


def is_aromatic_from_index(index, mode):
    """constants.py:97-106: per-atom flags for 'add_aromatic', None for 'basic'"""
    if mode == "basic":
        return None
    if mode not in _AROMATIC:
        raise ValueError(mode)
    return [_AROMATIC[mode][int(i)] for i in index]


# transforms that carry a ``mode`` (repo/datasets/transforms: featurize_ligand*, assign_atomtype, ...)
_MODE_KEYS = ("mode",)


def set_num_atom_type(config, num_type=None):
    """Inject ``config.model.num_atomtype`` (and ``config.mode``) from the transform list, like
    repo/utils/configuration.py:13-38."""
    if num_type is not None:
        config.model.num_atomtype = num_type
        return config
    if "test" in config.data:
        tsfm = config.data.test.transform
    elif "train" in config.data:
        tsfm = config.data.train.transform
    else:
        raise ValueError("no mode can be detected, please specific it.")
    mode = None
    for t in tsfm:
        if "mode" in t and t.mode in NUM_ATOM_TYPES:
            mode = t.mode
    if mode is None:
        raise ValueError("the mode cannot be inferred automatically, please specific it.")
    config.model.num_atomtype = NUM_ATOM_TYPES[mode]
    config.mode = mode
    return config

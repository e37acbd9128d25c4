"""The two plug points the reference's callers use:

* model registry  ``@register_model(name)`` / ``get_model(config)``   (repo/models/_base.py:5-14;
  callers sample.py:155, train.py:147)
* encoder factory ``get_e3_gnn(cfg, num_classes)`` keyed by ``cfg.type`` (repo/modules/e3nn/__init__.py:5-18)
"""
_MODEL_DICT = {}


def register_model(name):
    def decorator(cls):
        _MODEL_DICT[name] = cls
        return cls
    return decorator


def get_model(config):
    if config.type not in _MODEL_DICT:
        raise KeyError(f"Unknown model type: {config.type} (registered: {sorted(_MODEL_DICT)})")
    return _MODEL_DICT[config.type](config)


def registered_models():
    return sorted(_MODEL_DICT)


def get_e3_gnn(cfg, num_classes=None, num_edge_classes=None):
    if num_classes is not None:
        cfg.num_classes = num_classes
    if num_edge_classes is not None:
        cfg.num_edge_classes = num_edge_classes
    if cfg.type == "unitransformer":
        from .unitransformer import UniTransformer
        return UniTransformer(cfg)
    # gvptransformer / ipatransformer serve the autoregressive and D3FG models (out of scope)
    raise ValueError(f"Unknown model type: {cfg.type}")

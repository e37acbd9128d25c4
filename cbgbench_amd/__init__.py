"""cbgbench_amd -- MI355X-native denoising hot path of CBGBench.

Host side (Python, PyTorch-ROCm for device memory / streams) mirrors the reference's plug points:
``get_model`` / ``register_model`` (model registry), ``get_e3_gnn`` (encoder factory), ``load_config``.
The per-step equivariant message passing itself is libcbgx.so (hand-written gfx950 kernels behind the
C ABI of include/cbgx.h); there is no CPU fallback.
"""
from .config import Config, load_config, set_num_atom_type  # noqa: F401
from .registry import get_e3_gnn, get_model, register_model, registered_models  # noqa: F401
from . import targetdiff as _targetdiff  # noqa: F401  (registers 'targetdiff')
from .targetdiff import TargetDiff  # noqa: F401
from . import diffsbdd as _diffsbdd  # noqa: F401  (registers 'diffsbdd')
from .diffsbdd import DiffSBDD  # noqa: F401
from . import diffbp as _diffbp  # noqa: F401  (registers 'diffbp')
from .diffbp import DiffBP  # noqa: F401
from .unitransformer import UniTransformer  # noqa: F401

__all__ = ["Config", "load_config", "set_num_atom_type", "get_model", "register_model", "get_e3_gnn",
           "registered_models", "TargetDiff", "DiffSBDD", "DiffBP", "UniTransformer", "default_targetdiff_config",
           "default_diffsbdd_config", "default_diffbp_config"]


def default_targetdiff_config(num_atomtype=13, num_layers=9, num_diffusion_timesteps=1000):
    """The ``model:`` block every shipped targetdiff config uses (configs/denovo/train/targetdiff.yml:1-23),
    plus the ``num_atomtype`` that set_num_atom_type injects."""
    return Config(
        type="targetdiff",
        num_atomtype=num_atomtype,
        encoder=dict(type="unitransformer", node_feat_dim=128, n_heads=16, num_layers=num_layers),
        generator=dict(pos_schedule=dict(type="sigmoid", beta_start=1.0e-7, beta_end=2.0e-3),
                       atom_schedule=dict(type="cosine", cosine_s=0.01),
                       num_diffusion_timesteps=num_diffusion_timesteps, time_sampler="symmetric"),
        embedder=dict(emb_dim=128, atom=dict(type="linear"), residue=dict(type="linear")),
    )


def default_diffsbdd_config(num_atomtype=8, num_layers=9, num_diffusion_timesteps=1000, eval_interval=10):
    """The ``model:`` block of configs/denovo/train/diffsbdd.yml:1-19 (+ num_atomtype; the shipped test config uses
    the 'basic' 8-type vocabulary, configs/denovo/test/diffsbdd.yml:17)."""
    return Config(
        type="diffsbdd",
        num_atomtype=num_atomtype,
        eval_interval=eval_interval,     # evaluation times of the eval-mode loss (diffsbdd.py:74-76, default 10)
        encoder=dict(type="unitransformer", node_feat_dim=128, n_heads=16, num_layers=num_layers),
        generator=dict(pos_schedule=dict(type="polynomial_2"), atom_schedule=dict(type="polynomial_2"),
                       num_diffusion_timesteps=num_diffusion_timesteps, time_sampler="random"),
        embedder=dict(emb_dim=128, atom=dict(type="linear"), residue=dict(type="linear")),
    )


def default_diffbp_config(num_atomtype=13, num_layers=9, num_diffusion_timesteps=1000):
    """The ``model:`` block of configs/denovo/train/diffbp.yml:1-26 (+ num_atomtype)."""
    return Config(
        type="diffbp",
        num_atomtype=num_atomtype,
        encoder=dict(type="unitransformer", node_feat_dim=128, n_heads=16, num_layers=num_layers),
        generator=dict(pos_schedule=dict(type="sigmoid", beta_start=1.0e-7, beta_end=2.0e-3),
                       atom_schedule=dict(type="uniform"), num_diffusion_timesteps=num_diffusion_timesteps,
                       time_sampler="symmetric", com_schedule=dict(type="log", sigma_min=1.0e-7, sigma_max=5.0)),
        embedder=dict(emb_dim=128, atom=dict(type="linear"), residue=dict(type="linear")),
    )

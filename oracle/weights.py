"""TEST INFRASTRUCTURE ONLY -- deterministic synthetic TargetDiff weights.

No checkpoints ship with the reference (README.MD:100,241 point to Google
Drive) and there is no network, so parity tests and the bench use synthetic
weights.  They are generated from a numpy PCG64 stream keyed by the state-dict
key name, so the same ``state_dict`` can be rebuilt here, in the golden-vector
generator (which loads it into the *reference* ``TargetDiff`` with
``strict=True``) and on the GPU box, without shipping 10.7 MB of floats.

LayerNorm affines and biases are deliberately non-trivial (the reference's
default init has gamma=1, beta=0, which would hide indexing mistakes).

Key names / shapes follow SURVEY.md Appendix A.2; ``tests/golden/state_dict_keys_*.json``
(dumped from the reference model) pins them.
"""
import zlib

import numpy as np
import torch

from . import targetdiff as T


def _mlp_spec(prefix, d_in, d_hidden, d_out):
    return [
        (f"{prefix}.net.0.weight", (d_hidden, d_in), "w"),
        (f"{prefix}.net.0.bias", (d_hidden,), "b"),
        (f"{prefix}.net.1.weight", (d_hidden,), "g"),
        (f"{prefix}.net.1.bias", (d_hidden,), "b"),
        (f"{prefix}.net.3.weight", (d_out, d_hidden), "w"),
        (f"{prefix}.net.3.bias", (d_out,), "b"),
    ]


def denoiser_spec(num_classes, num_layers, H=128, heads=16, G=20, prefix="denoiser"):
    kv_in = 2 * H + 4 + 4 * G
    spec = [(f"{prefix}.dist_emb.0.offset", (G,), "offset")]
    spec += _mlp_spec(f"{prefix}.dist_emb.1", G, 8 * G, 1)
    for l in range(num_layers):
        p = f"{prefix}.blocks.{l}.x2h_layers.0"
        spec.append((f"{p}.distance_expansion.offset", (G,), "offset"))
        spec += _mlp_spec(f"{p}.hk_func", kv_in, H, H)
        spec += _mlp_spec(f"{p}.hv_func", kv_in, H, H)
        spec += _mlp_spec(f"{p}.hq_func", H, H, H)
        p = f"{prefix}.blocks.{l}.h2x_layers.0"
        spec.append((f"{p}.distance_expansion.offset", (G,), "offset"))
        spec += _mlp_spec(f"{p}.xk_func", kv_in, H, H)
        spec += _mlp_spec(f"{p}.xv_func", kv_in, H, heads)
        spec += _mlp_spec(f"{p}.xq_func", H, H, H)
    spec += [
        (f"{prefix}.classifier.0.weight", (H, H), "w"),
        (f"{prefix}.classifier.0.bias", (H,), "b"),
        (f"{prefix}.classifier.2.weight", (num_classes, H), "w"),
        (f"{prefix}.classifier.2.bias", (num_classes,), "b"),
    ]
    return spec


def embedder_spec(num_classes, H=128, prefix="context_embedder"):
    return [
        (f"{prefix}.ligand_atom_emb.weight", (H, num_classes), "w"),
        (f"{prefix}.ligand_atom_emb.bias", (H,), "b"),
        (f"{prefix}.protein_atom_emb.weight", (H, 7), "w"),
        (f"{prefix}.protein_atom_emb.bias", (H,), "b"),
        (f"{prefix}.residue_emb.weight", (H, 20), "w"),
        (f"{prefix}.residue_emb.bias", (H,), "b"),
        (f"{prefix}.ligand_indicator.weight", (H, 1), "w"),
        (f"{prefix}.ligand_indicator.bias", (H,), "b"),
    ]


def _gen(key, shape, kind, seed):
    if kind == "offset":
        from .unitransformer import RBF_OFFSETS
        return torch.tensor(RBF_OFFSETS, dtype=torch.float32)
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    a = rng.standard_normal(shape)
    if kind == "w":
        a = a / np.sqrt(shape[-1])
    elif kind == "b":
        a = 0.1 * a
    elif kind == "g":
        a = 1.0 + 0.1 * a
    return torch.from_numpy(a.astype(np.float32))


def schedule_state(num_timesteps=1000, pos=("sigmoid", 1e-7, 2e-3), atom=("cosine", 0.01)):
    """Frozen schedule tables that live in the reference state dict
    (pos_scheduler.* / type_scheduler.*; diffusion_scheduler.py:34-54,331-337)."""
    sd = {}
    pb = T.vp_betas(num_timesteps, pos[1], pos[2], pos[0])
    for k, v in T.vp_tables(pb).items():
        sd[f"pos_scheduler.{k}"] = v
    # TypeVPScheduler keeps the base-class defaults beta_start=1e-7, beta_end=2e-3 (unused for cosine)
    tb = T.vp_betas(num_timesteps, 1e-7, 2e-3, atom[0], cosine_s=atom[1])
    for k, v in T.type_tables(tb).items():
        sd[f"type_scheduler.{k}"] = v
    return sd


def synthetic_state_dict(num_classes=13, num_layers=9, seed=0, num_timesteps=1000):
    sd = {}
    sd.update(schedule_state(num_timesteps))
    for key, shape, kind in embedder_spec(num_classes) + denoiser_spec(num_classes, num_layers):
        sd[key] = _gen(key, shape, kind, seed)
    return sd


def synthetic_state_dict_diffsbdd(num_classes=8, num_layers=9, seed=0, num_timesteps=1000):
    """DiffSBDD: same denoiser / embedder keys; the schedules are two gamma tables (schedule_utils.py:60-90)."""
    from . import diffsbdd as D
    sd = {}
    g = D.polynomial_gamma(num_timesteps, 2.0, 5e-4)
    sd["pos_scheduler.gamma.gamma"] = g.clone()
    sd["type_scheduler.gamma.gamma"] = g.clone()
    for key, shape, kind in embedder_spec(num_classes) + denoiser_spec(num_classes, num_layers):
        sd[key] = _gen(key, shape, kind, seed)
    return sd


def com_head_spec(num_layers_com=3, H=128, heads=16, G=20, prefix="com_head"):
    kv_in = 2 * H + 4 + 4 * G
    spec = []
    for l in range(num_layers_com):
        p = f"{prefix}.h2xattentions.{l}"
        spec.append((f"{p}.distance_expansion.offset", (G,), "offset"))
        spec += _mlp_spec(f"{p}.xk_func", kv_in, H, H)
        spec += _mlp_spec(f"{p}.xv_func", kv_in, H, heads)
        spec += _mlp_spec(f"{p}.xq_func", H, H, H)
    spec.append((f"{prefix}.dist_emb.0.offset", (G,), "offset"))
    spec += _mlp_spec(f"{prefix}.dist_emb.1", G, 8 * G, 1)
    return spec


def synthetic_state_dict_diffbp(num_classes=13, num_layers=9, seed=0, num_timesteps=1000):
    """DiffBP: VP position tables only (MaskTypeSchedule has no parameters) + denoiser + embedder + com_head."""
    sd = {}
    pb = T.vp_betas(num_timesteps, 1e-7, 2e-3, "sigmoid")
    for k, v in T.vp_tables(pb).items():
        sd[f"pos_scheduler.{k}"] = v
    for key, shape, kind in embedder_spec(num_classes) + denoiser_spec(num_classes, num_layers) + com_head_spec():
        sd[key] = _gen(key, shape, kind, seed)
    return sd

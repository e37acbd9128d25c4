"""Deterministic synthetic parameters for benchmarks and smoke runs (no checkpoints ship with the reference and there
is no network).  Every trainable tensor is drawn from a numpy PCG64 stream keyed by (seed, crc32(state-dict key)):
weights ~ N(0, 1/fan_in), biases ~ 0.1 N(0,1), LayerNorm gains ~ 1 + 0.1 N(0,1) -- deliberately non-trivial affines so
that indexing mistakes cannot hide.  The same rule is implemented independently in oracle/weights.py (the test
infrastructure); tests/test_host.py checks that the two agree bit for bit."""
import zlib

import numpy as np
import torch


def draw(key, shape, seed=0):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    a = rng.standard_normal(tuple(shape))
    if key.endswith("net.1.weight"):                       # LayerNorm gain
        a = 1.0 + 0.1 * a
    elif key.endswith("bias"):
        a = 0.1 * a
    else:                                                   # Linear weight
        a = a / np.sqrt(shape[-1])
    return torch.from_numpy(a.astype(np.float32))


@torch.no_grad()
def fill_(model, seed=0):
    """overwrite every trainable parameter of ``model`` in place (frozen schedule tables and buffers are left alone)"""
    for key, p in model.named_parameters():
        if p.requires_grad:
            p.copy_(draw(key, p.shape, seed).to(p.device))
    return model

"""The VERIFIABLE ReLU-flip exception of the gradient tests (tests/test_gpu_training.py on the fixtures, tests/test_gpu_config_sized.py
at the configs[4] shape): test infrastructure around the CPU oracle, imported by both."""
import contextlib

import torch
import torch.nn.functional as F

from oracle import unitransformer as OU

FLIP_EPS = 5e-6      # |LayerNorm output| below which a GPU / CPU rounding difference (~1e-6) can put a ReLU on the other side of 0


@contextlib.contextmanager
def relu_margins(force=None):
    """While active, every MLP the CPU oracle evaluates records its pre-ReLU values within FLIP_EPS of zero as
    ``{prefix: [(row, unit), ...]}``; ``force = {prefix: [(row, unit)]}`` evaluates those units on the OTHER side of zero (the value
    moves by 2 |y| < 1e-5, the ReLU mask flips, the derivative with respect to everything upstream stays that of the formula).
    This makes the ReLU-flip exception of the gradient checks VERIFIABLE: a deviation is accepted only if the oracle with ONE
    identified near-zero unit flipped reproduces the GPU's gradients within the plain tolerance."""
    near, orig, force = {}, OU.mlp, force or {}

    def mlp(sd, prefix, z):
        y = F.linear(z, sd[prefix + ".net.0.weight"], sd[prefix + ".net.0.bias"])
        y = F.layer_norm(y, (y.shape[-1],), sd[prefix + ".net.1.weight"], sd[prefix + ".net.1.bias"], 1e-5)
        idx = torch.nonzero(y.detach().abs() < FLIP_EPS)
        if idx.numel():
            near.setdefault(prefix, []).extend((int(r), int(u)) for r, u in idx.reshape(-1, y.dim())[:, [0, -1]].tolist())
        if prefix in force:
            delta = torch.zeros_like(y)
            for r, u in force[prefix]:
                delta[r, u] = -2.0 * float(y[r, u].detach())
            y = y + delta
        return F.linear(F.relu(y), sd[prefix + ".net.3.weight"], sd[prefix + ".net.3.bias"])

    OU.mlp = mlp
    try:
        yield near
    finally:
        OU.mlp = orig



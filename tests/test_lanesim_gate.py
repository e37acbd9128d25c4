"""Lane-level model of gate_bwd_mfma_kernel (cbgbench_amd/csrc/train_reduce.hip): what each of the 64 lanes feeds to the two
v_mfma_f32_16x16x4_f32 products of a 16-edge tile and where the results land in the wave's slab, against torch.autograd on the gate MLP
of the reference (unitransformer.py:109-112: GaussianSmearing -> Linear(20,160) -> LayerNorm -> ReLU -> Linear(160,1) -> sigmoid).  It
pins the index math on the CPU; tests/test_gpu_training.py compares the kernel itself with the reference's gradients."""
import numpy as np
import torch

from tests.lanesim import C_, MU, Q_, mfma

G, GH = 20, 160


def row16_sum(v):
    return v.reshape(4, 16).sum(1, keepdims=True).repeat(16, 1).reshape(64)


def xrow_sum(v):
    return np.tile(v.reshape(4, 16).sum(0), 4)


def simulate_tile(dist16, dew16, W1, b1, gam, bet, w2, b2):
    """one tile: dist16 / dew16 [16] by edge slot (dew = 0 on invalid slots).  Returns the slab contributions."""
    f = np.float32
    dist = dist16[C_].astype(f)                # lane (j, *) holds edge j
    dew = dew16[C_].astype(f)
    y = [np.zeros((4, 64), f) for _ in range(10)]
    for s in range(5):
        ra = np.exp(-0.5 * (dist - MU[4 * s + Q_]) ** 2).astype(f)            # A[edge j][g = 4 s + q]
        for nt in range(10):
            y[nt] = mfma(ra, W1[16 * nt + C_, 4 * s + Q_], y[nt])           # B[g = 4 s + q][unit 16 nt + j]
    ssum = np.zeros((4, 64), f)
    for nt in range(10):
        y[nt] = y[nt] + b1[16 * nt + C_][None]
        ssum += y[nt]
    mean = np.stack([row16_sum(ssum[r]) for r in range(4)]) / GH
    var = np.zeros((4, 64), f)
    for nt in range(10):
        y[nt] = y[nt] - mean
        var += y[nt] ** 2
    rstd = 1.0 / np.sqrt(np.stack([row16_sum(var[r]) for r in range(4)]) / GH + 1e-5)
    acc = np.zeros((4, 64), f)
    for nt in range(10):
        y[nt] = y[nt] * rstd
        acc += w2[16 * nt + C_][None] * np.maximum(y[nt] * gam[16 * nt + C_][None] + bet[16 * nt + C_][None], 0)
    dacc = np.zeros((4, 64), f)
    dist_e = np.zeros((4, 64), f)
    for r in range(4):
        a = row16_sum(acc[r]) + b2
        ew = 1.0 / (1.0 + np.exp(-a))
        dacc[r] = dew[4 * Q_ + r] * ew * (1 - ew)          # __shfl(dew, 4 q + r): lanes 0..15 hold the edges
        dist_e[r] = dist[4 * Q_ + r]
    out = dict(W1=np.zeros((GH, G), f), b1=np.zeros(GH, f), g=np.zeros(GH, f), be=np.zeros(GH, f), w2=np.zeros(GH, f))
    s1 = np.zeros((4, 64), f); s2 = np.zeros((4, 64), f)
    aG = np.zeros((10, 64), f); aBe = np.zeros((10, 64), f); aW2 = np.zeros((10, 64), f); aB1 = np.zeros((10, 64), f)
    for nt in range(10):
        ga, be, ww = gam[16 * nt + C_][None], bet[16 * nt + C_][None], w2[16 * nt + C_][None]
        ya = y[nt] * ga + be
        aW2[nt] = (dacc * np.maximum(ya, 0)).sum(0)
        dy = np.where(ya > 0, dacc * ww, 0)
        aG[nt] = (dy * y[nt]).sum(0)
        aBe[nt] = dy.sum(0)
        dn = dy * ga
        s1 += dn
        s2 += dn * y[nt]
    s1 = np.stack([row16_sum(s1[r]) for r in range(4)]) / GH
    s2 = np.stack([row16_sum(s2[r]) for r in range(4)]) / GH
    aB2 = np.where(C_ == 0, dacc.sum(0), 0)
    rb0 = np.exp(-0.5 * (dist_e - MU[C_][None]) ** 2).astype(f)
    rb1 = np.where(C_ < 4, np.exp(-0.5 * (dist_e - MU[16 + (C_ & 3)][None]) ** 2), 0).astype(f)
    for nt in range(10):
        ga, be, ww = gam[16 * nt + C_][None], bet[16 * nt + C_][None], w2[16 * nt + C_][None]
        dn = np.where(y[nt] * ga + be > 0, dacc * ww * ga, 0)
        dp = rstd * (dn - s1 - y[nt] * s2)
        aB1[nt] = dp.sum(0)
        d0 = np.zeros((4, 64), f); d1 = np.zeros((4, 64), f)
        for r in range(4):
            d0 = mfma(dp[r], rb0[r], d0)
            d1 = mfma(dp[r], rb1[r], d1)
        for rr in range(4):           # lane (gc = j, qq = q), register rr <-> unit 16 nt + 4 qq + rr, g = gc | 16 + gc
            u = 16 * nt + 4 * Q_ + rr
            out["W1"][u, C_] += d0[rr]
            m = C_ < 4
            out["W1"][u[m], 16 + C_[m]] += d1[rr][m]
    for nt in range(10):
        sel = Q_ == 0
        out["b1"][16 * nt + C_[sel]] = xrow_sum(aB1[nt])[sel]
        out["g"][16 * nt + C_[sel]] = xrow_sum(aG[nt])[sel]
        out["be"][16 * nt + C_[sel]] = xrow_sum(aBe[nt])[sel]
        out["w2"][16 * nt + C_[sel]] = xrow_sum(aW2[nt])[sel]
    out["b2"] = float(xrow_sum(aB2.astype(f))[0])
    return out


def test_gate_backward_tile_matches_autograd():
    g = torch.Generator().manual_seed(0)
    W1 = (torch.randn(GH, G, generator=g) * 0.4).requires_grad_(True)
    b1 = (torch.randn(GH, generator=g) * 0.2).requires_grad_(True)
    gam = (1 + 0.3 * torch.randn(GH, generator=g)).requires_grad_(True)
    bet = (0.2 * torch.randn(GH, generator=g)).requires_grad_(True)
    w2 = (torch.randn(GH, generator=g) * 0.3).requires_grad_(True)
    b2 = torch.tensor(0.1, requires_grad=True)
    dist = torch.rand(16, generator=g) * 9
    dew = torch.randn(16, generator=g)
    dew[[3, 11, 15]] = 0.0           # invalid slots
    mu = torch.from_numpy(MU)
    r = torch.exp(-0.5 * (dist[:, None] - mu[None]) ** 2)
    y = r @ W1.t() + b1
    n = torch.nn.functional.layer_norm(y, (GH,), gam, bet, 1e-5)
    ew = torch.sigmoid(torch.relu(n) @ w2 + b2)
    (ew * dew).sum().backward()
    out = simulate_tile(dist.numpy(), dew.numpy(), *[t.detach().numpy() for t in (W1, b1, gam, bet, w2)], float(b2.detach()))
    for name, ref in (("W1", W1.grad), ("b1", b1.grad), ("g", gam.grad), ("be", bet.grad), ("w2", w2.grad)):
        ref = ref.numpy()
        assert np.abs(out[name] - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-7, (name, np.abs(out[name] - ref).max(), np.abs(ref).max())
    assert abs(out["b2"] - float(b2.grad)) <= 1e-5 * abs(float(b2.grad)) + 1e-7

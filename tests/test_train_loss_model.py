"""CPU model of train_loss_kernel (cbgbench_amd/csrc/train_loss.hip): the kernel's formulas -- the two losses of TargetDiff's training
step and their hand-derived gradients with respect to the denoiser's logits / positions -- restated in numpy in the kernel's order of
operations, against torch.autograd on the tensor path (cbgbench_amd/targetdiff.py: the restatement of the reference's schedulers,
diffusion_scheduler.py:185-201, 380-441, which tests/test_host.py pins to the oracle).  It pins the DERIVATION on the CPU; the kernel
itself is compared with the tensor path on the GPU in tests/test_gpu_train_loss.py.

The yardstick is the tensor path in fp64: in fp32 the gradient's largest entry (the class that holds almost all of the posterior) is a
difference of two numbers that agree to 4 - 7 digits, so autograd in fp32 is ~1e-4 of the largest entry off; the kernel takes that entry
from the sum rule sum_k d un_k = 0 instead and the model below must show that it is then good to ~1e-6."""
import math

import numpy as np
import pytest
import torch

import cbgbench_amd as C

LOG_TINY = np.float32(-69.07755278982137)


def lae(a, b):
    m = np.maximum(a, b)
    return (m + np.log(np.exp(a - m) + np.exp(b - m))).astype(a.dtype)


def kernel_model(dt, ts, logits, xo, x0, v0, vt, t, bl, gen, B, closure=True):
    """-> (loss_pos, loss_atom, d loss_pos / d xo, d loss_atom / d logits), all in dtype dt"""
    n, Cn = logits.shape
    z = logits.astype(dt)
    tab = lambda p: p.detach().numpy().astype(dt)
    la, l1a, lac, l1ac = tab(ts.log_alphas_v), tab(ts.log_one_minus_alphas_v), tab(ts.log_alphas_cumprod_v), tab(ts.log_one_minus_alphas_cumprod_v)
    logc = dt(np.float32(math.log(Cn)))
    tb = t[bl]
    tm1 = np.maximum(tb - 1, 0)
    mx = z.max(1, keepdims=True)
    lp = z - (mx + np.log(np.exp(z - mx).sum(1, keepdims=True)))
    a0, b0 = lac[tm1][:, None], (l1ac[tm1] - logc)[:, None].astype(dt)
    a1, b1 = la[tb][:, None], (l1a[tb] - logc)[:, None].astype(dt)
    k = np.arange(Cn)[None]
    oh0 = np.where(k == v0[:, None], dt(0), dt(LOG_TINY))
    oht = np.where(k == vt[:, None], dt(0), dt(LOG_TINY))
    Ap = lae(lp + a0, np.broadcast_to(b0, lp.shape).astype(dt))
    Aq = lae(oh0 + a0, np.broadcast_to(b0, lp.shape).astype(dt))
    Bk = lae(oht + a1, np.broadcast_to(b1, lp.shape).astype(dt))
    unp, unq = Ap + Bk, Aq + Bk

    def log_softmax(u):
        m = u.max(1, keepdims=True)
        return u - (m + np.log(np.exp(u - m).sum(1, keepdims=True)))

    logp, logq = log_softmax(unp), log_softmax(unq)
    q = np.exp(logq)
    e = np.where(k == v0[:, None], dt(1), dt(1e-30))
    kl = (q * (logq - logp)).sum(1)
    nll = -(e * logp).sum(1)
    m0 = (tb == 0).astype(dt)
    l_typ = m0 * nll + (1 - m0) * kl
    g = -(m0[:, None] * e + (1 - m0)[:, None] * q)
    du = g - np.exp(logp) * g.sum(1, keepdims=True)
    if closure:        # the dominant class's entry from sum_k du_k = 0
        d = unp.argmax(1)
        rows = np.arange(n)
        du[rows, d] = 0
        du[rows, d] = -du.sum(1)
    r = du * np.exp(lp + a0 - Ap)
    dz = r - np.exp(lp) * r.sum(1, keepdims=True)
    diff = xo.astype(dt) - x0.astype(dt)
    mse = (diff * diff).sum(1)
    cnt = np.zeros(B, dt)
    np.add.at(cnt, bl, gen.astype(dt))
    n_eff = dt(np.where(gen, bl, 0).max() + 1)
    w = np.where(gen, 1 / (np.maximum(cnt[bl], 1) * n_eff), 0).astype(dt)
    return (mse * w).sum(), (l_typ * w).sum(), 2 * diff * w[:, None], dz * w[:, None]


@pytest.mark.parametrize("Cn,B,t_list,scale", [(13, 6, [0, 999, 1, 0], 1.0), (8, 3, [500], 4.0), (32, 17, [0, 0, 2], 0.5), (13, 40, [0, 1, 999], 2.0)])
def test_loss_kernel_formulas_match_autograd(Cn, B, t_list, scale):
    m = C.get_model(C.default_targetdiff_config(Cn))
    ps, ts = m.pos_scheduler, m.type_scheduler
    g = torch.Generator().manual_seed(100 + B)
    nl = torch.randint(3, 30, (B,), generator=g)
    bl = torch.repeat_interleave(torch.arange(B), nl)
    n = int(nl.sum())
    gen = torch.rand(n, generator=g) >= 0.2
    if B > 2:
        gen[bl == 1] = False
    t = torch.randint(0, 1000, (B,), generator=g)
    t[:len(t_list)] = torch.tensor(t_list)
    v0 = torch.randint(0, Cn, (n,), generator=g)
    vt = ts.forward_add_noise(v0, t, bl, gen, uniform=torch.rand(n, Cn, generator=g))[1]
    x0 = torch.randn(n, 3, generator=g) * 3
    xo32 = torch.randn(n, 3, generator=g) * 3
    lg32 = torch.randn(n, Cn, generator=g) * scale
    ref = {}
    for name, dt in (("t64", torch.float64), ("t32", torch.float32)):
        xo, z = xo32.clone().to(dt).requires_grad_(True), lg32.clone().to(dt).requires_grad_(True)
        lp_, _ = ps.get_loss(xo, x0.to(dt), x0.to(dt), t, gen, bl, type="denoise")
        la_, _ = ts.get_loss(z, v0, vt, t, gen, bl, pred_logit=True)
        (lp_ + la_).backward()
        ref[name] = (float(lp_.detach()), float(la_.detach()), xo.grad.double().numpy(), z.grad.double().numpy())
    args = (ts, lg32.numpy(), xo32.numpy(), x0.numpy(), v0.numpy(), vt.numpy(), t.numpy(), bl.numpy(), gen.numpy(), B)
    k64 = kernel_model(np.float64, *args)
    k32 = kernel_model(np.float32, *args)
    k32_direct = kernel_model(np.float32, *args, closure=False)
    r64 = ref["t64"]
    # the derivation: in fp64 the model IS autograd (constants are the fp32 tables on both sides)
    assert abs(k64[0] - r64[0]) <= 1e-12 * abs(r64[0]) and abs(k64[1] - r64[1]) <= 1e-4 * abs(r64[1])
    for a, b in ((k64[2], r64[2]), (k64[3], r64[3])):
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    # fp32: the gradient with the sum-rule entry is two orders closer to fp64 than autograd in fp32 (and than the direct form)
    err = lambda a: np.abs(a.astype(np.float64) - r64[3]).max() / np.abs(r64[3]).max()
    e_model, e_direct, e_autograd = err(k32[3]), err(k32_direct[3]), err(ref["t32"][3])
    assert e_model <= 5e-6, (e_model, e_direct, e_autograd)
    assert e_direct >= 5 * e_model or e_direct <= 5e-6, (e_model, e_direct, e_autograd)
    assert abs(float(k32[1]) - r64[1]) <= 1e-4 * abs(r64[1]) and abs(float(k32[0]) - r64[0]) <= 1e-5 * abs(r64[0])

"""CPU model of the fused DiffBP training losses (cbgbench_amd/csrc/train_loss_diffbp.hip: diffbp_loss_kernel + diffbp_loss_finish_kernel,
combined by cbgbench_amd/diffbp.py::_DiffBPLossFunction.backward): the kernel's formulas -- zero-COM noise / centre-of-mass predictions,
the two score losses, the mask-type loss, the reverse-step mean, the interior loss and the hand-derived gradients of all four with respect
to the two networks' outputs -- restated graph by graph in torch and checked against autograd on the tensor path of ``DiffBP.get_loss``
(diffbp.py:79-101, 18-28, 131-234 of the reference; that path is pinned to the reference's goldens by tests/test_host_models_cpu.py).
The GPU suite compares the kernel itself with the same path (test_diffbp_fused_losses_match_the_tensor_path)."""
import pytest
import torch

import cbgbench_amd as C
from cbgbench_amd.diffsbdd import DiffsbddVariationalScheduler as S
from cbgbench_amd.targetdiff import TargetDiff


def kernel_model(xo, x_in, x_stack, logits, sort_idx, graph_ptr, pos_noise, com_noise, v0, type_flag, gen, t, n_rec, acp, betas,
                 rho=2.0, gamma=5.0):
    N, Cn, B, n_lig = xo.shape[0], logits.shape[1], t.shape[0], pos_noise.shape[0]
    a_pos, a_int, b_com, b_int = (torch.zeros(N, 3) for _ in range(4))
    z_atom = torch.zeros(N, Cn)
    gstats = torch.zeros(B, 6)
    for g in range(B):
        r0, r1 = int(graph_ptr[g]), int(graph_ptr[g + 1])
        rows = torch.arange(r0, r1)
        lrow = rows[sort_idx[rows] >= n_rec]                      # ligand rows: the tail of the graph's range
        prow = rows[sort_idx[rows] < n_rec]
        ai = sort_idx[lrow] - n_rec
        nl = lrow.numel()
        a, b = acp[t[g]], betas[t[g]]
        kap, isb = -b / ((1 - a).sqrt() * (1 - b).sqrt()), 1.0 / (1 - b).sqrt()
        xt = x_in[lrow]
        nz, dl = xo[lrow] - xt, x_stack[lrow] - xt
        eps, com = nz - nz.sum(0) / max(nl, 1), (dl.sum(0) / max(nl, 1)).expand(nl, 3)
        gn, tf = gen[ai], type_flag[ai]
        mp, mc = ((eps - pos_noise[ai]) ** 2).sum(1), ((com - com_noise[ai]) ** 2).sum(1)
        xs = torch.where(gn[:, None], (xt + b * (-(eps + com) / (1 - a).sqrt())) * isb, xt)
        # cross_entropy(p = softmax(z), v): -p_v + log sum exp(p);  d / d z = p (u - sum p u),  u = softmax(p) - onehot(v)
        p = torch.softmax(logits[lrow], 1)
        u = torch.softmax(p, 1) - torch.nn.functional.one_hot(v0[ai], Cn).float()
        ce = -p.gather(1, v0[ai, None])[:, 0] + torch.logsumexp(p, 1)
        dz = p * (u - (p * u).sum(1, keepdim=True))
        cg, ct = max(float(gn.sum()), 1.0), max(float(tf.sum()), 1.0)
        # interior term: every protein atom of the graph adds exp(-d^2 / rho) to every ligand atom
        d = xs[:, None, :] - x_in[prow][None, :, :]
        e = torch.exp(-(d ** 2).sum(-1) / rho)
        acc, sx = e.sum(1), (e[:, :, None] * d).sum(1)
        rr = gamma - (-rho * torch.log(acc + 1e-3))
        w = torch.where(rr >= 0, -2.0 / (n_lig * (acc + 1e-3)), torch.zeros_like(acc))
        gi = torch.where(gn[:, None], kap * w[:, None] * sx, torch.zeros(nl, 3))
        gp = torch.where(gn[:, None], 2.0 * (eps - pos_noise[ai]) / cg, torch.zeros(nl, 3))
        gc = torch.where(gn[:, None], 2.0 * (com - com_noise[ai]) / cg, torch.zeros(nl, 3))
        a_pos[lrow] = gp - gp.mean(0)                             # eps = noise - mean(noise): d / d noise_j = g_j - mean(g)
        a_int[lrow] = gi - gi.mean(0)
        b_com[lrow] = gc.mean(0).expand(nl, 3)                    # com = mean(delta): d / d delta_j = mean(g)
        b_int[lrow] = gi.mean(0).expand(nl, 3)
        z_atom[lrow] = torch.where(tf[:, None], dz / ct, torch.zeros(nl, Cn))
        gstats[g] = torch.stack([mp[gn].sum() / cg, mc[gn].sum() / cg, ce[tf].sum() / ct, rr.clamp(min=0).sum(), gn.sum().float(),
                                 tf.sum().float()])
    top = lambda col: float(max([g for g in range(B) if gstats[g, col] > 0], default=-1) + 1) or 1.0
    dg, dt = top(4), top(5)
    losses = torch.stack([gstats[:, 0].sum() / dg, gstats[:, 2].sum() / dt, gstats[:, 1].sum() / dg, gstats[:, 3].sum() / max(n_lig, 1)])
    return losses, (1.0 / dg, 1.0 / dt), a_pos, a_int, b_com, b_int, z_atom


@pytest.mark.parametrize("seed,empty_graph", [(0, False), (1, True)])
def test_fused_diffbp_loss_formulas_match_autograd_on_the_tensor_path(seed, empty_graph):
    gen = torch.Generator().manual_seed(seed)
    m = C.get_model(C.default_diffbp_config(13)).train()
    Cn, B = m.num_classes, 4
    sizes_r = [37, 52, 41, 45]
    sizes_l = [6, 9, 5, 7]
    br = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes_r))
    bl = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes_l))
    n_rec, n_lig = br.shape[0], bl.shape[0]
    sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TargetDiff.compose_plan(bl, br, B)
    N = n_rec + n_lig
    x_in = torch.randn(N, 3, generator=gen) * 2.0
    x_in[lig_rows] = x_in[lig_rows] * 0.3                                     # ligands inside their pockets: the interior term is active
    xo = (x_in + 0.3 * torch.randn(N, 3, generator=gen)).requires_grad_(True)
    x_stack = (x_in + 0.2 * torch.randn(N, 3, generator=gen)).requires_grad_(True)
    logits = torch.randn(N, Cn, generator=gen).requires_grad_(True)
    pos_noise, com_noise = torch.randn(n_lig, 3, generator=gen), torch.randn(n_lig, 3, generator=gen)
    v0 = torch.randint(0, Cn, (n_lig,), generator=gen)
    gen_l = torch.rand(n_lig, generator=gen) < 0.8
    type_flag = gen_l & (torch.rand(n_lig, generator=gen) < 0.6)
    if empty_graph:                                                           # the last graph has nothing movable / masked: D = 3
        gen_l[bl == B - 1] = False
        type_flag[bl == B - 1] = False
    t = torch.tensor([0, 17, 500, 999])
    ps = m.pos_scheduler
    # ---- the tensor path of DiffBP.get_loss behind its two network calls (cbgbench_amd/diffbp.py)
    x_t, x_rec = x_in[lig_rows], x_in[~lig_flag]
    noise = xo[lig_rows] - x_t
    x_lig_pred = noise - S.scatter_mean(noise, bl, B)[bl]
    x_com_pred = S.scatter_mean((x_stack - x_in)[lig_rows], bl, B)[bl]
    loss_pos = ps.get_score_loss(x_lig_pred, pos_noise, t, gen_l, bl, score_in=False)[0]
    loss_com = ps.get_score_loss(x_com_pred, com_noise, t, gen_l, bl, score_in=False, info_tag="com")[0]
    loss_atom = m.type_scheduler.get_loss(logits[lig_rows], v0, v0, t, type_flag, bl, pred_logit=True)[0]
    xs = ps.xs_mean(x_lig_pred + x_com_pred, x_t, t, bl, gen_flag=gen_l)
    loss_inter = m.interior_loss(xs, x_rec, bl, br, n_graphs=B, max_ligand_atoms=max(sizes_l))
    wts = {"pos": 1.0, "atom": 0.7, "com": 1.3, "inter": 0.9}
    (wts["pos"] * loss_pos + wts["atom"] * loss_atom + wts["com"] * loss_com + wts["inter"] * loss_inter).backward()
    # ---- the kernel's formulas
    with torch.no_grad():
        losses, scal, a_pos, a_int, b_com, b_int, z_atom = kernel_model(
            xo.detach(), x_in, x_stack.detach(), logits.detach(), sort_idx, graph_ptr, pos_noise, com_noise, v0, type_flag, gen_l, t, n_rec,
            ps.alphas_cumprod.float(), ps.betas.float())
    for k, ref in zip(range(4), (loss_pos, loss_atom, loss_com, loss_inter)):
        assert abs(float(losses[k]) - float(ref.detach())) <= 2e-5 * abs(float(ref.detach())) + 1e-7, (k, float(losses[k]), float(ref.detach()))
    assert float(loss_inter.detach()) > 0.0                                    # (the hinge is active: its gradient is tested)
    gx = wts["pos"] * scal[0] * a_pos + wts["inter"] * a_int                    # _DiffBPLossFunction.backward
    gs = wts["com"] * scal[0] * b_com + wts["inter"] * b_int
    gl = wts["atom"] * scal[1] * z_atom
    for got, ref, name in ((gx, xo.grad, "x_out"), (gs, x_stack.grad, "x_stack"), (gl, logits.grad, "logits")):
        assert float(ref[~lig_flag].abs().max()) == 0.0 and float(got[~lig_flag].abs().max()) == 0.0, name
        assert torch.allclose(got, ref, rtol=2e-4, atol=2e-7), (name, float((got - ref).abs().max()), float(ref.abs().max()))

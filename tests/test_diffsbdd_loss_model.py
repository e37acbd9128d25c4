"""CPU model of round 6's fused DiffSBDD training arithmetic (cbgbench_amd/csrc/train_loss_diffsbdd.hip: diffsbdd_noise_kernel,
diffsbdd_loss_kernel, diffsbdd_finish_kernel), graph by graph on the composed row order as the kernels walk it, against the tensor
path of ``DiffSBDD.get_loss`` -- which tests/test_host_models_cpu.py pins to the losses and gradients recorded from the unmodified
reference (diffsbdd.py:91-195, diffusion_scheduler.py:740-960).  The GPU suite compares the kernels themselves with both."""
import pytest
import torch

import cbgbench_amd as C
from cbgbench_amd.targetdiff import TargetDiff
from oracle import weights as W
from tests.test_host_models_cpu import golden_batch, load, with_oracle_denoiser


def kernel_model(model, batch, t_int, eps_x, eps_c, xo, logits):
    """returns (x_t, xr_t, c_t, loss_pos, loss_atom, gpos, gz): what the three launches write, in float32 like they do"""
    f = torch.float32
    T, Cn = model.num_diffusion_timesteps, model.num_classes
    x0, x_rec = batch["ligand_pos"].float(), batch["protein_pos"].float()
    v0, bl, br = batch["ligand_atom_type"], batch["ligand_element_batch"], batch["protein_element_batch"]
    gen = batch.get("ligand_gen_flag", batch["ligand_lig_flag"]).bool()
    B, n_rec = int(t_int.shape[0]), x_rec.shape[0]
    sort_idx, _, _, _, graph_ptr = TargetDiff.compose_plan(bl, br, B)
    ps = model.pos_scheduler
    gam = ps.gamma.gamma
    alpha_tab, sigma_tab = ps.alpha(gam), ps.sigma(gam)
    t_idx = torch.round((t_int / T) * T).long()
    x_t, xr_t, c_t = torch.empty_like(x0), torch.empty_like(x_rec), torch.empty(x0.shape[0], Cn)
    gdata = torch.zeros(B, 4)
    cdf = lambda v: 0.5 * (1.0 + torch.erf(v * 0.70710678118654752440))
    for g in range(B):                                                    # diffsbdd_noise_kernel: one workgroup per graph
        rows = sort_idx[graph_ptr[g]:graph_ptr[g + 1]]
        lig = rows[rows >= n_rec] - n_rec
        rec = rows[rows < n_rec]
        nl = lig.numel()
        a, s, aT, sT = alpha_tab[t_idx[g]], sigma_tab[t_idx[g]], alpha_tab[T], sigma_tab[T]
        m0 = x0[lig].sum(0) / max(nl, 1)
        xc = x0[lig] - m0
        xn = a * xc + s * eps_x[lig]
        m2 = xn.sum(0) / max(nl, 1)
        xr_t[rec] = (x_rec[rec] - m0) - m2
        x_t[lig] = torch.where(gen[lig, None], xn - m2, xc)
        c0 = torch.nn.functional.one_hot(v0[lig], Cn).to(f) * 0.25
        ct = torch.where(gen[lig, None], a * c0 + s * eps_c[lig], c0)
        c_t[lig] = ct
        ctr, sig0 = ct * 4.0 - 1.0, s * 4.0
        lp = torch.log(cdf((ctr + 0.5) / sig0) - cdf((ctr - 0.5) / sig0) + 1e-10)
        l0a = -(lp.gather(1, v0[lig, None])[:, 0] - torch.logsumexp(lp, 1)).sum()
        dp = float((nl - 1) * 3)
        gdata[g, 0] = nl
        gdata[g, 1] = dp * torch.log(1.0 / sT) + 0.5 * (dp * sT * sT + ((aT * xc) ** 2).sum()) - 0.5 * dp
        gdata[g, 2] = torch.log(1.0 / sT) + 0.5 * (sT * sT + ((aT * c0) ** 2).sum()) - 0.5
        gdata[g, 3] = l0a * float(t_idx[g] == 0)
    gpos, gz = torch.zeros_like(x0), torch.zeros(x0.shape[0], Cn)
    gl = torch.zeros(B, 2)
    for g in range(B):                                                    # diffsbdd_loss_kernel
        r = torch.arange(int(graph_ptr[g]), int(graph_ptr[g + 1]))
        r = r[sort_idx[r] >= n_rec]
        lig = sort_idx[r] - n_rec
        n, t0 = float(gdata[g, 0]), float(t_idx[g] == 0)
        dx, dc = eps_x[lig] - xo[r], eps_c[lig] - logits[r]
        gpos[lig] = -dx * (((1.0 - t0) / (n * 3.0) + t0) / B)
        gz[lig] = -dc * (((1.0 - t0) / (n * Cn)) / B)
        ep, ea = (dx ** 2).sum(), (dc ** 2).sum()
        gl[g, 0] = (0.5 * ep * (1.0 - t0) / (n * 3.0) + 0.5 * ep * t0) + gdata[g, 1]
        gl[g, 1] = (0.5 * ea * (1.0 - t0) / (n * Cn) + gdata[g, 3]) + gdata[g, 2]
    return x_t, xr_t, c_t, gl[:, 0].sum() / B, gl[:, 1].sum() / B, gpos, gz      # diffsbdd_finish_kernel: mean over the graphs


@pytest.mark.parametrize("case", ["train_loss_diffsbdd", "train_loss_diffsbdd_t0"])
def test_fused_diffsbdd_arithmetic_equals_the_tensor_path(golden_dir, case):
    g = load(golden_dir, case)
    sd = W.synthetic_state_dict_diffsbdd(8, 9, seed=0, num_timesteps=1000)
    m = with_oracle_denoiser(C.get_model(C.default_diffsbdd_config(8)), sd).train()
    batch = golden_batch(g)
    seen = {}
    inner = m.denoiser

    class Spy(torch.nn.Module):                 # records what the tensor path feeds the denoiser and gets back
        def forward(self, **kw):
            out = inner(**kw)
            xo, logits = out[0].detach().requires_grad_(True), out[2].detach().requires_grad_(True)
            seen.update(x=kw["x"].detach(), h=kw["h"].detach(), xo=xo, logits=logits)
            return xo, out[1], logits
    m.denoiser = Spy()
    ld, res = m(batch, t=g["t"], noise=(g["eps_x"], g["eps_c"]))
    (ld["pos"] + ld["atom"]).backward()
    with torch.no_grad():
        x_t, xr_t, c_t, lp, la, gpos, gz = kernel_model(m, batch, g["t"], g["eps_x"].float(), g["eps_c"].float(), seen["xo"], seen["logits"])
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    sort_idx, _, _, lig_rows, _ = TargetDiff.compose_plan(bl, br, int(g["t"].shape[0]))
    n_rec = br.shape[0]
    # the noised inputs: composed x of the tensor path = cat(xr_t, x_t)[sort_idx]
    assert torch.allclose(torch.cat([xr_t, x_t])[sort_idx], seen["x"], atol=2e-6)
    # losses (the goldens pin the tensor path: 5e-6) and the gradients with respect to the denoiser outputs
    assert abs(float(lp) - float(ld["pos"].detach())) <= 5e-6 * abs(float(ld["pos"].detach())) + 1e-7
    assert abs(float(la) - float(ld["atom"].detach())) <= 5e-6 * abs(float(ld["atom"].detach())) + 1e-7
    assert abs(float(lp) - g["loss_pos"]) <= 2e-5 * abs(g["loss_pos"]) + 1e-7 and abs(float(la) - g["loss_atom"]) <= 2e-5 * abs(g["loss_atom"]) + 1e-7
    gx_ref, gl_ref = seen["xo"].grad, seen["logits"].grad
    assert float(gx_ref[sort_idx < n_rec].abs().max()) == 0.0 and float(gl_ref[sort_idx < n_rec].abs().max()) == 0.0
    assert torch.allclose(gpos, gx_ref[lig_rows], rtol=1e-5, atol=1e-9) and torch.allclose(gz, gl_ref[lig_rows], rtol=1e-5, atol=1e-9)
    assert torch.allclose(res["eps_pred_pos"], seen["xo"].detach()[lig_rows]) and torch.equal(res["eps_0_atom"], g["eps_c"].float())

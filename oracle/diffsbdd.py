"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DiffSBDD sampler around the shared denoiser.

Restates ``DiffSBDD.sample`` (repo/models/diffusion/diffsbdd.py:240-361) and the pieces of
``DiffsbddVariationalScheduler`` it uses (repo/models/diffusion/diffusion_scheduler.py:670-760, 965-1040;
``PredefinedNoiseSchedule`` schedule_utils.py:42-96) in plain torch on the CPU, with every Gaussian draw an
explicit input so the HIP path and the oracle can replay the same noise.  Draw order in the reference:
initial positions [n_lig,3], initial types [n_lig,C], then per step positions, types, and finally the two
draws of ``sample_p_xh_given_z0``.  Pinned against the reference by ``oracle/make_golden.py`` (the reference
ships no tests).  Quirks kept: positions use std 1 / types std 4 normalisation; every ligand atom
is updated (``gen_flag`` is not consulted by the sampler); the pocket is re-centred on the ligand mean at every
draw; the final ``c`` is the un-normalised *input* of the last call (its sampled value is discarded,
diffsbdd.py:345-350)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import targetdiff as T
from . import unitransformer as U


def polynomial_gamma(timesteps, power=2.0, precision=5e-4):
    """PredefinedNoiseSchedule('polynomial_<power>', timesteps, precision).gamma  (schedule_utils.py:42-90)."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(x / steps, power)) ** 2
    a2 = np.concatenate([np.ones(1), alphas2], axis=0)
    step = np.clip(a2[1:] / a2[:-1], a_min=0.001, a_max=1.0)
    alphas2 = np.cumprod(step, axis=0)
    alphas2 = (1 - 2 * precision) * alphas2 + precision
    sigmas2 = 1 - alphas2
    return torch.from_numpy(-(np.log(alphas2) - np.log(sigmas2))).float()


def gamma_at(gamma_tab, t, timesteps):
    """PredefinedNoiseSchedule.forward (schedule_utils.py:92-94): t in [0,1] -> table lookup."""
    return gamma_tab[torch.round(t * timesteps).long()]


def scatter_mean(src, index, n):
    s = torch.zeros((n,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    return s / cnt.clamp(min=1).view(-1, *[1] * (src.dim() - 1))


def remove_mean_batch(x_lig, x_rec, bl, br, B):
    """diffusion_scheduler.py:708-712."""
    mean = scatter_mean(x_lig, bl, B)
    return x_lig - mean[bl], x_rec - mean[br]


def sample_normal_zero_com(mu_lig, xh0_pocket, sigma, bl, br, B, eps, com):
    """diffusion_scheduler.py:965-980 with the randn draw `eps` given."""
    out = mu_lig + sigma[bl] * eps
    if com:
        return remove_mean_batch(out, xh0_pocket, bl, br, B)
    return out


def sample_p_zs_given_zt(gamma_tab, Tn, s, t, zt_lig, xh0_pocket, bl, br, B, eps_t_lig, eps, com):
    """diffusion_scheduler.py:1012-1040 (+ sigma_and_alpha_t_given_s :982-1010)."""
    gs, gt = gamma_at(gamma_tab, s, Tn), gamma_at(gamma_tab, t, Tn)
    sigma2_ts = (-torch.expm1(F.softplus(gs) - F.softplus(gt))).view(-1, 1)
    alpha_ts = torch.exp(0.5 * (F.logsigmoid(-gt) - F.logsigmoid(-gs))).view(-1, 1)
    sigma_ts = torch.sqrt(sigma2_ts)
    sigma_s = torch.sqrt(torch.sigmoid(gs)).view(-1, 1)
    sigma_t = torch.sqrt(torch.sigmoid(gt)).view(-1, 1)
    mu = zt_lig / alpha_ts[bl] - (sigma2_ts / alpha_ts / sigma_t)[bl] * eps_t_lig
    sigma = sigma_ts * sigma_s / sigma_t
    if com:
        return sample_normal_zero_com(mu, xh0_pocket, sigma, bl, br, B, eps, True)
    return sample_normal_zero_com(mu, xh0_pocket, sigma, bl, br, B, eps, False), xh0_pocket


def denoise(sd, batch, x_lig, c_lig, x_rec, v_rec_norm):
    """embed + compose + denoiser exactly as diffsbdd.py:283-295 (time embedding absent in shipped configs)."""
    aa = F.one_hot(batch["protein_aa_type"], 20).to(x_lig.dtype)
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    n_lig, n_rec = x_lig.shape[0], x_rec.shape[0]
    h_lig, h_rec = T.context_embed(sd, c_lig, v_rec_norm, aa)
    sort_idx, batch_idx = T.compose(bl, br)
    x = torch.cat([x_rec, x_lig], 0)[sort_idx]
    h = torch.cat([h_rec, h_lig], 0)[sort_idx]
    lig_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), torch.ones(n_lig, dtype=torch.bool)], 0)[sort_idx]
    gen_l = batch.get("ligand_gen_flag", torch.ones(n_lig, dtype=torch.bool))
    gen_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), gen_l], 0)[sort_idx]
    xo, _, logits = U.unitransformer_forward(sd, x, h, batch_idx, lig_flag, gen_flag)
    return xo[lig_flag], logits[lig_flag]


def sample(sd, batch, num_classes, Tn, draws):
    """Full DiffSBDD.sample; ``draws`` is the list of randn tensors in reference order. Returns the trajectory
    dict like the reference (keys Tn-1 .. -1; key 0 overwritten by the final x / c)."""
    gamma_tab = sd["pos_scheduler.gamma.gamma"]          # the sampler uses pos_scheduler for both (diffsbdd.py:297-304)
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    B = int(bl.max()) + 1
    x_rec = batch["protein_pos"]                          # normalize_pos: std 1, mean 0
    v_rec = batch["protein_atom_feature"] / 4.0           # normalize_type: std 4
    it = iter(draws)
    mu_x = scatter_mean(x_rec, br, B)[bl]
    mu_h = torch.zeros(B, num_classes)[bl]
    sigma1 = torch.ones(B, 1)
    x_lig, x_rec = sample_normal_zero_com(mu_x, x_rec, sigma1, bl, br, B, next(it), True)
    c_lig = sample_normal_zero_com(mu_h, v_rec, sigma1, bl, br, B, next(it), False)
    traj = {Tn - 1: (x_lig, c_lig)}
    for t_idx in reversed(range(Tn)):
        x_lig, c_lig = traj[t_idx]
        s = torch.full((B,), t_idx) / Tn
        t = (torch.full((B,), t_idx) + 1) / Tn
        x_pred, c_out = denoise(sd, batch, x_lig, c_lig, x_rec, v_rec)
        x_next, x_rec = sample_p_zs_given_zt(gamma_tab, Tn, s, t, x_lig, x_rec, bl, br, B, x_pred, next(it), True)
        c_next, _ = sample_p_zs_given_zt(gamma_tab, Tn, s, t, c_lig, v_rec, bl, br, B, c_out, next(it), False)
        traj[t_idx - 1] = (x_next, c_next)
    # sample_p_xh_given_z0 (diffsbdd.py:321-352)
    x_lig, c_lig = traj[-1]
    g0 = gamma_at(gamma_tab, torch.zeros(B), Tn)
    sigma0 = torch.exp(0.5 * g0).unsqueeze(1)
    x_pred, c_out = denoise(sd, batch, x_lig, c_lig, x_rec, v_rec)
    sig_t = torch.sqrt(torch.sigmoid(g0)).view(-1, 1)
    alp_t = torch.sqrt(torch.sigmoid(-g0)).view(-1, 1)
    mu_x = 1.0 / alp_t[bl] * (x_lig - sig_t[bl] * x_pred)
    x_fin, _ = sample_normal_zero_com(mu_x, x_rec, sigma0, bl, br, B, next(it), True)
    next(it)                                              # the type draw is made and discarded
    traj[0] = (x_fin * 1.0, c_lig * 4.0)
    return traj

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


# ---- training objective (diffsbdd.py:91-195; DiffsbddVariationalScheduler diffusion_scheduler.py:670-960) ----------
def _sum_per_graph(x, index, n):
    return torch.zeros(n, dtype=x.dtype).index_add(0, index, x.sum(-1))


def _cdf(x):
    return 0.5 * (1.0 + torch.erf(x / np.sqrt(2.0)))


def kl_prior(gamma_tab, Tn, x, bl, B, dimensions):
    """diffusion_scheduler.py:783-800 with gaussian_KL :693-706 (p = standard normal)"""
    g_T = gamma_at(gamma_tab, torch.ones(B, 1), Tn)
    alpha_T = torch.sqrt(torch.sigmoid(-g_T)).view(B, 1)
    mu = alpha_T[bl] * x
    sigma_T = torch.sqrt(torch.sigmoid(g_T)).view(B)
    mu2 = _sum_per_graph(mu ** 2, bl, B)
    d = dimensions
    return d * torch.log(1.0 / sigma_T) + 0.5 * (d * sigma_T ** 2 + mu2) - 0.5 * d


def score_loss_training(gamma_tab, Tn, pred, tgt, t, bl, B, t_is_zero, x0=None, c0=None, c_t=None):
    """get_score_loss in training mode (diffusion_scheduler.py:886-900, 930-945): per graph
    0.5 * sum(err^2) * [t != 0] / (n * dim)  +  (-log p(. | z_0)) * [t == 0]  +  KL prior; then the mean over graphs."""
    n = torch.bincount(bl, minlength=B)
    err = _sum_per_graph((tgt - pred) ** 2, bl, B)
    loss_t = 0.5 * err * (1.0 - t_is_zero) / (n * pred.shape[-1])
    g_t = gamma_at(gamma_tab, t, Tn).view(B, 1)
    if x0 is not None:
        loss_0 = -(-0.5 * err) * t_is_zero
        kl = kl_prior(gamma_tab, Tn, x0, bl, B, (n - 1) * 3)
    else:
        sigma0 = torch.sqrt(torch.sigmoid(g_t)) * 4.0
        onehot = c0 * 4.0
        centred = c_t * 4.0 - 1.0
        logp = torch.log(_cdf((centred + 0.5) / sigma0[bl]) - _cdf((centred - 0.5) / sigma0[bl]) + 1e-10)
        logp = logp - torch.logsumexp(logp, dim=1, keepdim=True)
        loss_0 = -_sum_per_graph(logp * onehot, bl, B) * t_is_zero
        kl = kl_prior(gamma_tab, Tn, c0, bl, B, 1)
    return (loss_t + loss_0 + kl).mean()


def get_loss(sd, batch, t_int, eps_x, eps_c, num_classes, Tn):
    """DiffSBDD.get_loss in training mode (diffsbdd.py:91-195).  ``t_int`` [B] float in {0..T} (the 'random' time sampler,
    _base.py:30-33); draws: randn_like(x) then randn_like(c)."""
    x0 = batch["ligand_pos"]
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    n_lig = x0.shape[0]
    gen_l = batch.get("ligand_gen_flag", torch.ones(n_lig, dtype=torch.bool))
    B = int(bl.max()) + 1
    c0 = F.one_hot(batch["ligand_atom_type"], num_classes) / 4.0
    v_rec = batch["protein_atom_feature"] / 4.0
    t_is_zero = (t_int == 0).float()
    t = t_int / Tn
    gp, gt_ = sd["pos_scheduler.gamma.gamma"], sd["type_scheduler.gamma.gamma"]
    x0c, xr0 = remove_mean_batch(x0, batch["protein_pos"], bl, br, B)
    g = gamma_at(gp, t, Tn).view(B, 1)
    x_noisy = torch.sqrt(torch.sigmoid(-g))[bl] * x0c + torch.sqrt(torch.sigmoid(g))[bl] * eps_x
    x_noisy, xr_t = remove_mean_batch(x_noisy, xr0.detach().clone(), bl, br, B)
    x_t = torch.where(gen_l.unsqueeze(-1), x_noisy, x0c)
    g2 = gamma_at(gt_, t, Tn).view(B, 1)
    c_t = torch.where(gen_l.unsqueeze(-1), torch.sqrt(torch.sigmoid(-g2))[bl] * c0 + torch.sqrt(torch.sigmoid(g2))[bl] * eps_c, c0)
    x_pred, c_pred = denoise(sd, batch, x_t, c_t, xr_t, v_rec)
    return {"pos": score_loss_training(gp, Tn, x_pred, eps_x, t, bl, B, t_is_zero, x0=x0c),
            "atom": score_loss_training(gt_, Tn, c_pred, eps_c, t, bl, B, t_is_zero, c0=c0, c_t=c_t)}


def loss_and_grads(sd, batch, t_int, eps_x, eps_c, num_classes, Tn, weights=None):
    sd = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in sd if "scheduler" not in k and not k.endswith(".offset")]
    for k in keys:
        sd[k].requires_grad_(True)
    losses = get_loss(sd, batch, t_int, eps_x, eps_c, num_classes, Tn)
    w = weights or {"pos": 1.0, "atom": 1.0}
    sum(w[k] * v for k, v in losses.items()).backward()
    grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}
    return {k: v.detach() for k, v in losses.items()}, grads


# ---- evaluation-mode objective (diffsbdd.py:72-86,138-153; diffusion_scheduler.py:679-691,864-928) ----------------
def neg_log_constants(gamma_tab, Tn, n, dim):
    """-log_constants_p_x_given_z0 (:679-691): (n-1) dim (0.5 gamma_0 + 0.5 log 2 pi) per graph"""
    g0 = gamma_at(gamma_tab, torch.zeros(n.shape[0], 1), Tn).view(-1)
    return ((n - 1) * dim) * (0.5 * g0 + 0.5 * np.log(2 * np.pi))


def score_loss_eval(gamma_tab, Tn, pred, tgt, s, t, bl, B, pred0, tgt0, x0=None, c0=None, c_t0=None):
    """get_score_loss outside training (:902-928): per graph  -T/2 (1 - SNR(gamma_s - gamma_t)) sum(err^2)  +  KL prior
    +  -log p(. | z_0) from a second network call on the t = 0 noising (pred0, tgt0; c_t0 = the t = 0 noised types)
    + the Gaussian normalisation constant (added to the type term too, with dim = C, as the reference does)."""
    n = torch.bincount(bl, minlength=B)
    err = _sum_per_graph((tgt - pred) ** 2, bl, B)
    g_s, g_t = gamma_at(gamma_tab, s, Tn).view(B), gamma_at(gamma_tab, t, Tn).view(B)
    loss_t = -Tn * 0.5 * (1.0 - torch.exp(-(g_s - g_t))) * err
    g_0 = gamma_at(gamma_tab, torch.zeros_like(s), Tn).view(B, 1)
    if x0 is not None:
        kl = kl_prior(gamma_tab, Tn, x0, bl, B, (n - 1) * 3)
        loss_0 = 0.5 * _sum_per_graph((tgt0 - pred0) ** 2, bl, B)
    else:
        kl = kl_prior(gamma_tab, Tn, c0, bl, B, 1)
        sigma0 = torch.sqrt(torch.sigmoid(g_0)) * 4.0
        centred = c_t0 * 4.0 - 1.0
        logp = torch.log(_cdf((centred + 0.5) / sigma0[bl]) - _cdf((centred - 0.5) / sigma0[bl]) + 1e-10)
        logp = logp - torch.logsumexp(logp, dim=1, keepdim=True)
        loss_0 = -_sum_per_graph(logp * (c0 * 4.0), bl, B)
    loss_0 = loss_0 + neg_log_constants(gamma_tab, Tn, n, tgt0.shape[-1])
    return (loss_t + loss_0 + kl).mean()


def _noised(sd, batch, t, eps_x, eps_c, x0c, xr0, c0, gen_l, bl, br, B, Tn):
    gp, gt_ = sd["pos_scheduler.gamma.gamma"], sd["type_scheduler.gamma.gamma"]
    g = gamma_at(gp, t, Tn).view(B, 1)
    x_noisy = torch.sqrt(torch.sigmoid(-g))[bl] * x0c + torch.sqrt(torch.sigmoid(g))[bl] * eps_x
    x_noisy, xr_t = remove_mean_batch(x_noisy, xr0.detach().clone(), bl, br, B)
    x_t = torch.where(gen_l.unsqueeze(-1), x_noisy, x0c)
    g2 = gamma_at(gt_, t, Tn).view(B, 1)
    c_t = torch.where(gen_l.unsqueeze(-1), torch.sqrt(torch.sigmoid(-g2))[bl] * c0 + torch.sqrt(torch.sigmoid(g2))[bl] * eps_c, c0)
    return x_t, c_t, xr_t


def get_loss_eval(sd, batch, t_long, draws, num_classes, Tn):
    """DiffSBDD.get_loss with the model in eval mode (diffsbdd.py:95-191): ``t_long`` [B] integer times;
    ``draws`` = (eps_x, eps_c, eps_x0, eps_c0) in the reference's draw order."""
    x0 = batch["ligand_pos"]
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    gen_l = batch.get("ligand_gen_flag", torch.ones(x0.shape[0], dtype=torch.bool))
    B = int(bl.max()) + 1
    c0 = F.one_hot(batch["ligand_atom_type"], num_classes) / 4.0
    v_rec = batch["protein_atom_feature"] / 4.0
    s = (t_long - 1) / Tn
    t = t_long / Tn
    gp, gt_ = sd["pos_scheduler.gamma.gamma"], sd["type_scheduler.gamma.gamma"]
    x0c, xr0 = remove_mean_batch(x0, batch["protein_pos"], bl, br, B)
    eps_x, eps_c, eps_x0, eps_c0 = draws
    x_t, c_t, xr_t = _noised(sd, batch, t, eps_x, eps_c, x0c, xr0, c0, gen_l, bl, br, B, Tn)
    x_pred, c_pred = denoise(sd, batch, x_t, c_t, xr_t, v_rec)
    x_z, c_z, xr_z = _noised(sd, batch, torch.zeros_like(s), eps_x0, eps_c0, x0c, xr0, c0, gen_l, bl, br, B, Tn)
    x_pred0, c_pred0 = denoise(sd, batch, x_z, c_z, xr_z, v_rec)
    return {"pos": score_loss_eval(gp, Tn, x_pred, eps_x, s, t, bl, B, x_pred0, eps_x0, x0=x0c),
            "atom": score_loss_eval(gt_, Tn, c_pred, eps_c, s, t, bl, B, c_pred0, eps_c0, c0=c0, c_t0=c_z)}


def eval_times(Tn, eval_interval=10):
    """diffsbdd.py:74-78: np.linspace(1, T, eval_interval), truncated to integers"""
    return [int(v) for v in np.linspace(1, Tn, eval_interval)]


def forward_eval(sd, batch, draws_per_time, num_classes, Tn, eval_interval=10):
    """DiffSBDD.forward in eval mode (diffsbdd.py:72-86): the mean of get_loss over the evaluation times."""
    B = int(batch["ligand_element_batch"].max()) + 1
    tot = {"pos": 0.0, "atom": 0.0}
    times = eval_times(Tn, eval_interval)
    for tv, draws in zip(times, draws_per_time):
        ld = get_loss_eval(sd, batch, torch.full((B,), tv, dtype=torch.long), draws, num_classes, Tn)
        for k in tot:
            tot[k] = tot[k] + ld[k]
    return {k: v / len(times) for k, v in tot.items()}

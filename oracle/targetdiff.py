"""TEST INFRASTRUCTURE ONLY -- CPU oracle for one TargetDiff reverse-diffusion step.

Restates, in plain torch on CPU, what ``TargetDiff.sample`` does per step around
the denoiser: context embedding, pocket+ligand composition, position posterior
sampling and categorical type posterior sampling.  Noise is an explicit input
(``eps`` ~ N(0,1) [N_lig,3], ``u`` ~ U(0,1) [N_lig,C]) so a host-generated noise
tape can teacher-force the HIP path and the oracle identically (the reference
draws ``randn_like`` then ``rand_like`` inside the two scheduler calls,
diffusion_scheduler.py:163 and models/utils/categorical.py:27).

Pinned against the reference's own classes by ``oracle/make_golden.py`` /
``tests/test_oracle_golden.py`` (the reference ships no tests: SURVEY.md 8c).
Paths cited are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import unitransformer as U


# ----------------------------------------------------------------------------
# schedule tables (diffusion_scheduler.py:27-100, 320-337), float64 -> float32
# ----------------------------------------------------------------------------
def vp_betas(num_timestep, beta_start, beta_end, type, cosine_s=0.008):
    """VPScheduler.init_betas, diffusion_scheduler.py:56-100 ('sigmoid' and 'cosine')."""
    if type == "sigmoid":
        b = np.linspace(-6, 6, num_timestep)
        return 1 / (np.exp(-b) + 1) * (beta_end - beta_start) + beta_start
    if type == "cosine":
        steps = num_timestep + 1
        x = np.linspace(0, steps, steps)
        ac = np.cos(((x / steps) + cosine_s) / (1 + cosine_s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        alphas = np.clip(ac[1:] / ac[:-1], a_min=0.001, a_max=1.0)
        return 1.0 - np.sqrt(alphas)
    raise NotImplementedError(type)


def vp_tables(betas):
    """VPScheduler.__init__, diffusion_scheduler.py:31-54."""
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    t = {
        "betas": betas, "alphas": alphas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac), "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac), "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_mean_c0_coef": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_ct_coef": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
        "posterior_var": post_var,
    }
    # register_from_numpy casts posterior_var to float32 *before* the log (line 54 reads self.posterior_var)
    pv32 = post_var.astype(np.float32)
    t["posterior_logvar"] = np.log(np.append(pv32[1], pv32[1:]))
    return {k: torch.from_numpy(np.asarray(v)).float() for k, v in t.items()}


def type_tables(betas):
    """TypeVPScheduler.__init__, diffusion_scheduler.py:320-337 (log tables from the fp32 alphas)."""
    t = vp_tables(betas)
    alphas_v = t["alphas"].numpy()
    log_alphas_v = np.log(alphas_v)
    log_ac = np.cumsum(log_alphas_v)

    def log_1_min_a(a):
        return np.log(1 - np.exp(a) + 1e-40)

    t["log_alphas_v"] = torch.from_numpy(log_alphas_v).float()
    t["log_one_minus_alphas_v"] = torch.from_numpy(log_1_min_a(log_alphas_v)).float()
    t["log_alphas_cumprod_v"] = torch.from_numpy(log_ac).float()
    t["log_one_minus_alphas_cumprod_v"] = torch.from_numpy(log_1_min_a(log_ac)).float()
    return t


# ----------------------------------------------------------------------------
# per-step pieces
# ----------------------------------------------------------------------------
def context_embed(sd, c_lig, v_rec, aa_rec_onehot, prefix="context_embedder"):
    """PLContextEmbedder.forward, modules/context_emb.py:179-230 with the shipped
    config (atom: linear, residue: linear, no time, no vec): time embedding = zeros."""
    def lin(name, z):
        return F.linear(z, sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"])

    n_lig, n_rec = c_lig.shape[0], v_rec.shape[0]
    h_lig = lin("ligand_atom_emb", c_lig) + 0.0 + lin("ligand_indicator", torch.ones(n_lig, 1, dtype=c_lig.dtype))
    h_rec = lin("protein_atom_emb", v_rec) + 0.0 + lin("residue_emb", aa_rec_onehot) \
        + lin("ligand_indicator", torch.zeros(n_rec, 1, dtype=c_lig.dtype))
    return h_lig, h_rec


def compose(batch_idx_lig, batch_idx_rec):
    """compose_context, modules/common.py:189-214: cat(rec, lig) then stable sort by graph id."""
    batch_ctx = torch.cat([batch_idx_rec, batch_idx_lig], 0)
    sort_idx = torch.sort(batch_ctx, stable=True).indices
    return sort_idx, batch_ctx[sort_idx]


def pos_backward_denoise(tb, x_pred, x_noisy, t, batch_idx, gen_flag, eps):
    """CTNVPScheduler.backward_remove_noise(type='denoise'), diffusion_scheduler.py:144-165."""
    nonzero = (1 - (t == 0).to(x_noisy.dtype))[batch_idx].unsqueeze(-1)
    mean = tb["posterior_mean_c0_coef"][t][batch_idx][:, None] * x_pred \
        + tb["posterior_mean_ct_coef"][t][batch_idx][:, None] * x_noisy
    logvar = tb["posterior_logvar"][t][batch_idx][:, None]
    xs = mean + nonzero * (0.5 * logvar).exp() * eps
    return torch.where(gen_flag.unsqueeze(-1), xs, x_noisy)


def log_add_exp(a, b):
    """models/utils/categorical.py:35-37."""
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def type_backward(tb, num_classes, c_pred_logits, ct, t, batch_idx, gen_flag, u):
    """TypeVPScheduler.backward_remove_noise(pred_logit=True), diffusion_scheduler.py:367-378,
    q_v_posterior :407-418, q_v_pred :420-429, q_v_pred_one_timestep :431-441,
    log_sample_categorical categorical.py:26-32."""
    log_c_pred = F.log_softmax(c_pred_logits, dim=-1)
    log_ct = torch.log(ct + 1e-8)
    tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)
    lc = math.log(num_classes)  # np.log(self.num_classes) is a float64 python scalar there
    a = tb["log_alphas_cumprod_v"][tm1][batch_idx].unsqueeze(-1)
    b = tb["log_one_minus_alphas_cumprod_v"][tm1][batch_idx].unsqueeze(-1)
    log_qvt1_v0 = log_add_exp(log_c_pred + a, b - lc)
    a1 = tb["log_alphas_v"][t][batch_idx].unsqueeze(-1)
    b1 = tb["log_one_minus_alphas_v"][t][batch_idx].unsqueeze(-1)
    log_qvs1_vt = log_add_exp(log_ct + a1, b1 - lc)
    un = log_qvt1_v0 + log_qvs1_vt
    logp = un - torch.logsumexp(un, dim=-1, keepdim=True)
    gumbel = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    v_next = (gumbel + logp).argmax(dim=-1)
    v_next = torch.where(gen_flag, v_next, ct.argmax(-1))
    return F.one_hot(v_next, num_classes).to(ct.dtype), v_next


def tables_from_state_dict(sd):
    pos = {k[len("pos_scheduler."):]: v for k, v in sd.items() if k.startswith("pos_scheduler.")}
    typ = {k[len("type_scheduler."):]: v for k, v in sd.items() if k.startswith("type_scheduler.")}
    return pos, typ


def denoise_step(sd, batch, x_lig, c_lig, t_idx, eps, u, num_classes, return_net_out=False):
    """One iteration of the loop at models/diffusion/targetdiff.py:150-182.

    ``batch`` holds the reference batch keys: protein_pos,
    protein_atom_feature, protein_aa_type, ligand_element_batch,
    protein_element_batch, ligand_gen_flag (optional).  Returns (x_next, c_next)."""
    x_rec = batch["protein_pos"]
    v_rec = batch["protein_atom_feature"]
    aa = F.one_hot(batch["protein_aa_type"], 20).to(x_lig.dtype)
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    n_lig, n_rec = x_lig.shape[0], x_rec.shape[0]
    lig_flag_l = torch.ones(n_lig, dtype=torch.bool)
    gen_l = batch.get("ligand_gen_flag", lig_flag_l)
    B = int(bl.max()) + 1
    t = torch.full((B,), t_idx, dtype=torch.long)

    h_lig, h_rec = context_embed(sd, c_lig, v_rec, aa)
    sort_idx, batch_idx = compose(bl, br)
    x = torch.cat([x_rec, x_lig], 0)[sort_idx]
    h = torch.cat([h_rec, h_lig], 0)[sort_idx]
    lig_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), lig_flag_l], 0)[sort_idx]
    gen_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), gen_l], 0)[sort_idx]

    xo, ho, logits = U.unitransformer_forward(sd, x, h, batch_idx, lig_flag, gen_flag)
    x_pred, c_pred = xo[lig_flag], logits[lig_flag]

    pos_tb, typ_tb = tables_from_state_dict(sd)
    x_next = pos_backward_denoise(pos_tb, x_pred, x_lig, t, bl, gen_l, eps)
    c_next, _ = type_backward(typ_tb, num_classes, c_pred, c_lig, t, bl, gen_l, u)
    if return_net_out:
        return x_next, c_next, x_pred, c_pred
    return x_next, c_next

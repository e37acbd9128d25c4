"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the TargetDiff training objective.

Follows the reference (file:line under /root/reference):
  * time sampler 'symmetric'                      repo/models/diffusion/_base.py:21-28
  * position forward noising                      repo/models/diffusion/diffusion_scheduler.py:117-134
  * categorical forward noising (Gumbel-argmax)   diffusion_scheduler.py:339-346,380-397; utils/categorical.py:26-32
  * TargetDiff.get_loss                           repo/models/diffusion/targetdiff.py:82-124
  * position loss type='denoise'                  diffusion_scheduler.py:185-201
  * type loss (KL / decoder NLL at t=0)           diffusion_scheduler.py:348-365,398-441; utils/categorical.py:18-23
  * weighted sum of the two losses                repo/utils/train.py:121-133, configs/denovo/train/targetdiff.yml:34-36

Everything is plain differentiable torch, so ``torch.autograd`` on this restatement is the reference for the
hand-written backward kernels of libcbgx.  Pinned to the unmodified reference by
tests/golden/train_loss_*.npz (oracle/make_golden.py: loss values and parameter gradients from the
reference ``TargetDiff.get_loss`` + ``loss.backward()``).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

from . import targetdiff as T
from . import unitransformer as U


def sample_time_symmetric(num_graphs, num_timesteps, draws):
    """_base.py:21-28.  ``draws``: the ``num_graphs // 2 + 1`` integers torch.randint would return."""
    time = torch.as_tensor(draws, dtype=torch.long)
    assert time.numel() == num_graphs // 2 + 1
    return torch.cat([time, num_timesteps - time - 1], 0)[:num_graphs]


def scatter_mean(src, index, n):
    """pure-torch torch_scatter.scatter_mean over dim 0 (count clamped to >= 1)."""
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
    return out / cnt.view((n,) + (1,) * (src.dim() - 1))


def pos_forward_add_noise(tb, x0, t, batch_idx, gen_flag, eps):
    """diffusion_scheduler.py:117-134 (zero_center=False)."""
    a = tb["alphas_cumprod"].index_select(0, t)[batch_idx].unsqueeze(-1)
    x_noisy = a.sqrt() * x0 + (1.0 - a).sqrt() * eps
    return torch.where(gen_flag.unsqueeze(-1), x_noisy, x0)


def index_to_log_onehot(v, num_classes):
    return torch.log(F.one_hot(v, num_classes).float().clamp(min=1e-30))


def q_v_pred(tb, num_classes, log_v0, t, batch):
    return T.log_add_exp(log_v0 + tb["log_alphas_cumprod_v"][t][batch].unsqueeze(-1),
                         tb["log_one_minus_alphas_cumprod_v"][t][batch].unsqueeze(-1) - math.log(num_classes))


def q_v_pred_one_timestep(tb, num_classes, log_vt_1, t, batch):
    return T.log_add_exp(log_vt_1 + tb["log_alphas_v"][t][batch].unsqueeze(-1),
                         tb["log_one_minus_alphas_v"][t][batch].unsqueeze(-1) - math.log(num_classes))


def q_v_posterior(tb, num_classes, log_v0, log_vt, t, batch):
    """diffusion_scheduler.py:407-418."""
    tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)
    un = q_v_pred(tb, num_classes, log_v0, tm1, batch) + q_v_pred_one_timestep(tb, num_classes, log_vt, t, batch)
    return un - torch.logsumexp(un, dim=-1, keepdim=True)


def type_forward_add_noise(tb, num_classes, v0, t, batch_idx, gen_flag, u):
    """diffusion_scheduler.py:339-346: q(v_t | v_0) sampled with the Gumbel-argmax of categorical.py:26-32."""
    log_q = q_v_pred(tb, num_classes, index_to_log_onehot(v0, num_classes), t, batch_idx)
    gumbel = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    v_noisy = torch.where(gen_flag, (gumbel + log_q).argmax(-1), v0)
    return F.one_hot(v_noisy, num_classes).float(), v_noisy


def pos_loss(x_pred, x0, gen_flag, batch_idx, n_graphs):
    """diffusion_scheduler.py:185-201 with type='denoise' (the call at targetdiff.py:109-111)."""
    mse = ((x_pred - x0) ** 2).sum(-1)
    return scatter_mean(mse[gen_flag], batch_idx[gen_flag], n_graphs).mean()


def type_loss(tb, num_classes, logits, v0, vt, t, gen_flag, batch_idx, n_graphs):
    """diffusion_scheduler.py:348-365,398-405."""
    log_c0 = index_to_log_onehot(v0, num_classes)
    log_ct = index_to_log_onehot(vt, num_classes)
    log_pred = F.log_softmax(logits, dim=-1)
    log_p = q_v_posterior(tb, num_classes, log_pred, log_ct, t, batch_idx)
    log_q = q_v_posterior(tb, num_classes, log_c0, log_ct, t, batch_idx)
    kl = (log_q.exp() * (log_q - log_p)).sum(1)
    nll = -(log_c0.exp() * log_p).sum(1)
    mask = (t == 0).float()[batch_idx]
    per_atom = mask * nll + (1.0 - mask) * kl
    return scatter_mean(per_atom[gen_flag], batch_idx[gen_flag], n_graphs).mean()


def get_loss(sd, batch, t, eps, u, num_classes, return_net_out=False):
    """TargetDiff.get_loss (targetdiff.py:82-124).  ``t`` [B] long, ``eps`` [N_lig,3], ``u`` [N_lig,C]
    replace the reference's randn_like / rand_like draws.  Returns {'pos','atom'} scalars."""
    x0, v0 = batch["ligand_pos"], batch["ligand_atom_type"]
    x_rec, v_rec = batch["protein_pos"], batch["protein_atom_feature"]
    aa = F.one_hot(batch["protein_aa_type"], 20).to(x0.dtype)
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    n_lig, n_rec = x0.shape[0], x_rec.shape[0]
    lig_l = torch.ones(n_lig, dtype=torch.bool)
    gen_l = batch.get("ligand_gen_flag", lig_l)
    B = int(bl.max()) + 1
    # the reference scatter_mean has dim_size = max(index)+1 over the *generated* rows
    n_loss = int(bl[gen_l].max()) + 1 if bool(gen_l.any()) else 0
    pos_tb, typ_tb = T.tables_from_state_dict(sd)

    x_t = pos_forward_add_noise(pos_tb, x0, t, bl, gen_l, eps)
    c_t, v_t = type_forward_add_noise(typ_tb, num_classes, v0, t, bl, gen_l, u)

    h_lig, h_rec = T.context_embed(sd, c_t, v_rec, aa)
    sort_idx, batch_idx = T.compose(bl, br)
    x = torch.cat([x_rec, x_t], 0)[sort_idx]
    h = torch.cat([h_rec, h_lig], 0)[sort_idx]
    lig_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), lig_l], 0)[sort_idx]
    gen_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), gen_l], 0)[sort_idx]
    xo, ho, logits = U.unitransformer_forward(sd, x, h, batch_idx, lig_flag, gen_flag)
    x_pred, c_pred = xo[lig_flag], logits[lig_flag]

    losses = {"pos": pos_loss(x_pred, x0, gen_l, bl, n_loss),
              "atom": type_loss(typ_tb, num_classes, c_pred, v0, v_t, t, gen_l, bl, n_loss)}
    if return_net_out:
        return losses, {"x_t": x_t, "v_t": v_t, "x_pred": x_pred, "c_pred": c_pred, "x": x, "h": h,
                        "batch_idx": batch_idx, "lig_flag": lig_flag, "gen_flag": gen_flag}
    return losses


def weighted_loss(losses, weights=None):
    """repo/utils/train.py:121-133; the shipped weights are pos 1.0 / atom 100.0."""
    weights = weights or {"pos": 1.0, "atom": 100.0}
    return sum(weights[k] * v for k, v in losses.items())


def trainable(sd):
    """keys the reference optimiser updates: everything except the frozen schedule tables and RBF offsets."""
    return [k for k in sd if "scheduler" not in k and not k.endswith(".offset")]


def loss_and_grads(sd, batch, t, eps, u, num_classes, weights=None):
    """loss.backward() of train.py:185-190 on the restatement; returns (losses, {key: grad})."""
    sd = {k: v.clone() for k, v in sd.items()}
    keys = trainable(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    losses = get_loss(sd, batch, t, eps, u, num_classes)
    weighted_loss(losses, weights).backward()
    grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}
    return {k: v.detach() for k, v in losses.items()}, grads

"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DiffBP sampler (repo/models/diffusion/diffbp.py:240-299).

Per step: the shared denoiser, then ``CoMPredictor`` (diffbp.py:30-101: zero-COM noise prediction + three
``H2XAttention`` blocks on the kNN graph / gate of the *input* coordinates with the denoiser's *output* features),
then the score-type position update (``CTNVPScheduler.backward_remove_noise(type='score')``,
diffusion_scheduler.py:144-158) and the absorbing-state type update (``MaskTypeSchedule.backward_remove_noise``,
:475-496).  Draw order per step in the reference: ``randn_like(x)`` [n_lig,3] then ``rand_like(v)`` [n_lig].
Noise is an explicit input.  Pinned against the reference by ``oracle/make_golden.py``."""
import torch
import torch.nn.functional as F

from . import targetdiff as T
from . import unitransformer as U


def scatter_mean(src, index, n):
    s = torch.zeros((n,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    return s / cnt.clamp(min=1).view(-1, *[1] * (src.dim() - 1))


def com_head(sd, x_lig_pred, bl, x, h, gen_flag, lig_flag, batch_idx, B, num_layers=3, prefix="com_head"):
    """CoMPredictor.forward, diffbp.py:79-101."""
    noise = x_lig_pred - x[lig_flag]
    noise = noise - scatter_mean(noise, bl, B)[bl]
    edge_index = U.knn_graph(x, batch_idx, 32)
    edge_type = U.build_edge_type(edge_index, lig_flag)
    sdp = {k: v for k, v in sd.items() if k.startswith(prefix)}
    e_w = U.edge_gate(sdp, prefix, x, edge_index)
    x_out = x.clone()
    for l in range(num_layers):
        dx = U.h2x_attention(sdp, f"{prefix}.h2xattentions.{l}", x_out, h, edge_type, edge_index, e_w)
        x_out = x_out + dx * gen_flag.unsqueeze(-1).to(x.dtype)
    shift = scatter_mean((x_out - x)[lig_flag], bl, B)[bl]
    return noise, shift


def pos_backward_score(tb, x_pred, x_noisy, t, bl, gen, eps):
    """CTNVPScheduler.backward_remove_noise(type='score'), diffusion_scheduler.py:144-158."""
    a = tb["alphas_cumprod"].index_select(0, t)[:, None][bl].expand_as(x_noisy)
    b = tb["betas"].index_select(0, t)[:, None][bl].expand_as(x_noisy)
    nonzero = (1 - (t == 0).float())[bl].unsqueeze(-1)
    sigma = (1 - a).sqrt()
    score = -x_pred / sigma
    xs = (x_noisy + b * score) / (1 - b).sqrt()
    xs = xs + nonzero * b.sqrt() * eps
    return torch.where(gen.unsqueeze(-1), xs, x_noisy)


def type_backward_mask(T_steps, num_classes, c_pred, ct, t, bl, gen, u, absorbing_state=0):
    """MaskTypeSchedule.backward_remove_noise(pred_logit=True, fix_pred=True), diffusion_scheduler.py:475-496."""
    p = F.softmax(c_pred, dim=-1)
    vt = ct.argmax(-1)
    tt = t[bl]
    prob = ((T_steps - tt) / T_steps).clamp(max=1.0, min=0.0)
    change = (u < prob) & gen & (vt == absorbing_state)
    v_next = torch.where(change, p.argmax(-1), vt)
    return F.one_hot(v_next, num_classes).float(), v_next


def denoise_step(sd, batch, x_lig, c_lig, t_idx, eps, u, num_classes, T_steps):
    """One iteration of the loop at diffbp.py:262-297."""
    x_rec = batch["protein_pos"]
    aa = F.one_hot(batch["protein_aa_type"], 20).to(x_lig.dtype)
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    n_lig, n_rec = x_lig.shape[0], x_rec.shape[0]
    gen_l = batch.get("ligand_gen_flag", torch.ones(n_lig, dtype=torch.bool))
    B = int(bl.max()) + 1
    t = torch.full((B,), t_idx, dtype=torch.long)
    h_lig, h_rec = T.context_embed(sd, c_lig, batch["protein_atom_feature"], aa)
    sort_idx, batch_idx = T.compose(bl, br)
    x = torch.cat([x_rec, x_lig], 0)[sort_idx]
    h = torch.cat([h_rec, h_lig], 0)[sort_idx]
    lig_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), torch.ones(n_lig, dtype=torch.bool)], 0)[sort_idx]
    gen_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), gen_l], 0)[sort_idx]
    xo, ho, logits = U.unitransformer_forward(sd, x, h, batch_idx, lig_flag, gen_flag)
    noise, shift = com_head(sd, xo[lig_flag], bl, x, ho, gen_flag, lig_flag, batch_idx, B)
    pos_tb = {k[len("pos_scheduler."):]: v for k, v in sd.items() if k.startswith("pos_scheduler.")}
    x_next = pos_backward_score(pos_tb, noise + shift, x_lig, t, bl, gen_l, eps)
    c_next, _ = type_backward_mask(T_steps, num_classes, logits[lig_flag], c_lig, t, bl, gen_l, u)
    return x_next, c_next


# ---- training objective (diffbp.py:154-231) -----------------------------------------------------------------------
def knn_cross(x, y, batch_x, batch_y, k):
    """torch_cluster.knn(x, y, k, batch_x, batch_y): for each point of y the (up to) k nearest points of x in the same
    graph.  Returns [2, E]: row 0 = index into y, row 1 = index into x.  Exact squared distances, ties by index."""
    rows, cols = [], []
    for g in torch.unique(batch_y).tolist():
        iy = torch.nonzero(batch_y == g).flatten()
        ix = torch.nonzero(batch_x == g).flatten()
        if ix.numel() == 0:
            continue
        d2 = ((y[iy][:, None, :] - x[ix][None, :, :]) ** 2).sum(-1)
        kk = min(k, ix.numel())
        nn_idx = torch.topk(d2, kk, dim=1, largest=False).indices
        rows.append(iy[:, None].expand(-1, kk).reshape(-1))
        cols.append(ix[nn_idx].reshape(-1))
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def interior_loss(x_ligand, x_protein, batch_ligand, batch_protein, k=48, rho=2.0, gamma=5.0):
    """diffbp.py:18-28"""
    ei = knn_cross(x_ligand, x_protein, batch_ligand, batch_protein, k)
    protein_idx, ligand_idx = ei[0], ei[1]
    dist2 = ((x_ligand[ligand_idx] - x_protein[protein_idx]) ** 2).sum(-1)
    e = (-dist2 / rho).exp()
    acc = torch.zeros(x_ligand.shape[0], dtype=x_ligand.dtype).index_add(0, ligand_idx, e)
    loss_per_ligand = -rho * (acc + 1e-3).log()
    return torch.clamp(gamma - loss_per_ligand, min=0.0).mean()


def pos_forward_add_noise_zero_center(tb, x, t, bl, gen, noise, B):
    """CTNVPScheduler.forward_add_noise(zero_center=True), diffusion_scheduler.py:117-134: the noisy positions use the
    full noise; the returned targets are its zero-COM part and its per-graph mean."""
    com = scatter_mean(noise, bl, B)[bl]
    a = tb["alphas_cumprod"].index_select(0, t)[bl].unsqueeze(-1)
    x_noisy = a.sqrt() * x + (1.0 - a).sqrt() * noise
    return torch.where(gen.unsqueeze(-1), x_noisy, x), noise - com, com


def score_loss(pred, tgt, gen, bl):
    """CTNVPScheduler.get_score_loss(score_in=False), diffusion_scheduler.py:203-218"""
    mse = ((pred - tgt) ** 2).sum(-1)
    n = int(bl[gen].max()) + 1
    return scatter_mean(mse[gen], bl[gen], n).mean()


def xs_mean_score(tb, x_pred, x_noisy, t, bl, gen):
    """CTNVPScheduler.xs_mean(type='score'), diffusion_scheduler.py:166-183"""
    a = tb["alphas_cumprod"].index_select(0, t)[:, None][bl].expand_as(x_noisy)
    b = tb["betas"].index_select(0, t)[:, None][bl].expand_as(x_noisy)
    score = -x_pred / (1 - a).sqrt()
    xs = (x_noisy + b * score) / (1 - b).sqrt()
    return torch.where(gen.unsqueeze(-1), xs, x_noisy)


def mask_forward_add_noise(T_steps, num_classes, v0, t, bl, gen, u, absorbing_state=0):
    """MaskTypeSchedule.forward_add_noise, diffusion_scheduler.py:452-473 -> (v_t, c_t, diff_mask)"""
    prob = t[bl].float().clamp(min=0.0) / T_steps
    diff_mask = (u < prob) & gen
    v_t = torch.where(diff_mask, torch.full_like(v0, absorbing_state), v0)
    return v_t, F.one_hot(v_t, num_classes).float(), diff_mask


def mask_type_loss(logits, v0, flag, bl):
    """MaskTypeSchedule.get_loss(pred_logit=True), :499-511 -- cross_entropy of the *softmax output* (as the reference
    does), over the atoms masked by the forward process"""
    loss_v = F.cross_entropy(F.softmax(logits, dim=-1), v0, reduction="none")
    if not bool(flag.any()):
        return torch.zeros_like(v0).float().mean()
    n = int(bl[flag].max()) + 1
    return scatter_mean(loss_v[flag], bl[flag], n).mean()


def get_loss(sd, batch, t, eps, u, num_classes, T_steps):
    """DiffBP.get_loss (diffbp.py:160-231).  Draw order in the reference: randn_like(x) then rand_like(v)."""
    x0, v0 = batch["ligand_pos"], batch["ligand_atom_type"]
    x_rec = batch["protein_pos"]
    aa = F.one_hot(batch["protein_aa_type"], 20).to(x0.dtype)
    bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
    n_lig, n_rec = x0.shape[0], x_rec.shape[0]
    gen_l = batch.get("ligand_gen_flag", torch.ones(n_lig, dtype=torch.bool))
    B = int(bl.max()) + 1
    pos_tb = {k[len("pos_scheduler."):]: v for k, v in sd.items() if k.startswith("pos_scheduler.")}
    x_t, pos_noise, com_noise = pos_forward_add_noise_zero_center(pos_tb, x0, t, bl, gen_l, eps, B)
    v_t, c_t, type_flag = mask_forward_add_noise(T_steps, num_classes, v0, t, bl, gen_l, u)
    h_lig, h_rec = T.context_embed(sd, c_t, batch["protein_atom_feature"], aa)
    sort_idx, batch_idx = T.compose(bl, br)
    x = torch.cat([x_rec, x_t], 0)[sort_idx]
    h = torch.cat([h_rec, h_lig], 0)[sort_idx]
    lig_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), torch.ones(n_lig, dtype=torch.bool)], 0)[sort_idx]
    gen_flag = torch.cat([torch.zeros(n_rec, dtype=torch.bool), gen_l], 0)[sort_idx]
    xo, ho, logits = U.unitransformer_forward(sd, x, h, batch_idx, lig_flag, gen_flag)
    x_lig_pred, x_com_pred = com_head(sd, xo[lig_flag], bl, x, ho, gen_flag, lig_flag, batch_idx, B)
    xs = xs_mean_score(pos_tb, x_lig_pred + x_com_pred, x_t, t, bl, gen_l)
    return {"pos": score_loss(x_lig_pred, pos_noise, gen_l, bl),
            "atom": mask_type_loss(logits[lig_flag], v0, type_flag, bl),
            "com": score_loss(x_com_pred, com_noise, gen_l, bl),
            "inter": interior_loss(xs, x_rec, bl, br)}


def loss_and_grads(sd, batch, t, eps, u, num_classes, T_steps, weights=None):
    """(weighted) sum of the four losses (configs/denovo/train/diffbp.yml:37-41: all weights 1, the default) and its gradients"""
    sd = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in sd if "scheduler" not in k and not k.endswith(".offset")]
    for k in keys:
        sd[k].requires_grad_(True)
    losses = get_loss(sd, batch, t, eps, u, num_classes, T_steps)
    w = weights or {k: 1.0 for k in losses}
    sum(w[k] * v for k, v in losses.items()).backward()
    grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}
    return {k: v.detach() for k, v in losses.items()}, grads

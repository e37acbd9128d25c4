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
        dx = U.h2x_attention(_rename(sdp, f"{prefix}.h2xattentions.{l}"), "blk", x_out, h, edge_type, edge_index, e_w)
        x_out = x_out + dx * gen_flag.unsqueeze(-1).to(x.dtype)
    shift = scatter_mean((x_out - x)[lig_flag], bl, B)[bl]
    return noise, shift


def _rename(sd, prefix):
    return {"blk" + k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix + ".")}


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

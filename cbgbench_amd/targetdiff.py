"""``targetdiff`` model class behind the reference's model registry
(repo/models/diffusion/targetdiff.py:14-184): same constructor config, same state-dict keys
(``pos_scheduler.*`` / ``type_scheduler.*`` frozen tables, ``context_embedder.*``, ``denoiser.*``),
same ``sample(batch) -> traj`` contract.

What differs by design:
* the denoiser call runs in libcbgx (gfx950 kernels);
* the static parts of a step (protein embedding, pocket+ligand composition permutation, CSR graph
  offsets) are computed once per batch instead of once per step -- protein atoms never move
  (unitransformer.py:182) and the shipped configs have no time embedding (context_emb.py:190-195);
* the trajectory stays on the GPU and is copied to the host once at the end instead of one
  ``.cpu()`` per step (targetdiff.py:182), preserving the returned structure
  ``traj[t] = (pos, type_onehot, batch_idx)`` for t = T-1 .. -1.
"""
import ctypes
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _native
from .registry import get_e3_gnn, register_model
from .unitransformer import graph_ptr_from_batch

NUM_AA = 20          # len(aa_name_number), repo/utils/protein/constants.py:39
PROTEIN_FEAT = 7     # len(atomic_numbers) + 1 (is_backbone), protein_featurizer.py:21-26


def _frozen(a):
    return nn.Parameter(torch.from_numpy(np.asarray(a)).float(), requires_grad=False)


class VPSchedule(nn.Module):
    """Frozen variance-preserving schedule tables, key-for-key the reference's ``VPScheduler``
    (repo/models/diffusion/diffusion_scheduler.py:27-100), computed in float64 then cast to fp32."""

    def __init__(self, num_timestep, beta_start=1e-7, beta_end=2e-3, type="sigmoid", cosine_s=0.008):
        super().__init__()
        betas = self.init_betas(beta_start, beta_end, num_timestep, type, cosine_s)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.betas = _frozen(betas)
        self.alphas = _frozen(alphas)
        self.alphas_cumprod = _frozen(ac)
        self.alphas_cumprod_prev = _frozen(ac_prev)
        self.sqrt_alphas_cumprod = _frozen(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = _frozen(np.sqrt(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = _frozen(np.sqrt(1.0 / ac))
        self.sqrt_recipm1_alphas_cumprod = _frozen(np.sqrt(1.0 / ac - 1))
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_mean_c0_coef = _frozen(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.posterior_mean_ct_coef = _frozen((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        self.posterior_var = _frozen(post_var)
        pv32 = post_var.astype(np.float32)  # the reference takes the log of the fp32 table (line 54)
        self.posterior_logvar = _frozen(np.log(np.append(pv32[1], pv32[1:])))

    @staticmethod
    def init_betas(beta_start, beta_end, num_timestep, type, cosine_s):
        if type == "sigmoid":
            b = np.linspace(-6, 6, num_timestep)
            betas = 1 / (np.exp(-b) + 1) * (beta_end - beta_start) + beta_start
        elif type == "cosine":
            steps = num_timestep + 1
            t = np.linspace(0, steps, steps)
            ac = np.cos(((t / steps) + cosine_s) / (1 + cosine_s) * np.pi * 0.5) ** 2
            ac = ac / ac[0]
            betas = 1.0 - np.sqrt(np.clip(ac[1:] / ac[:-1], a_min=0.001, a_max=1.0))
        elif type == "linear":
            betas = np.linspace(beta_start, beta_end, num_timestep, dtype=np.float64)
        elif type == "quad":
            betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_timestep, dtype=np.float64) ** 2
        elif type == "const":
            betas = beta_end * np.ones(num_timestep, dtype=np.float64)
        else:
            raise NotImplementedError(type)
        assert betas.shape == (num_timestep,)
        return betas


def scatter_mean(src, index, dim_size=None):
    """torch_scatter.scatter_mean over dim 0 (count clamped to >= 1), as called by the reference's losses."""
    n = int(index.max().item()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).index_add(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device).index_add(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).view((n,) + (1,) * (src.dim() - 1))


def masked_graph_mean(values, index, mask, n_graphs):
    """``scatter_mean(values[mask], index[mask]).mean()`` -- the per-graph mean over the masked rows, averaged over the graphs -- without
    the two host synchronisations of that expression (boolean indexing and the output size of the scatter): rows outside the mask
    enter with weight 0, and the outer mean divides by the largest masked graph id + 1, which is how torch_scatter sizes its output."""
    w = mask.to(values.dtype)
    s = torch.zeros(n_graphs, dtype=values.dtype, device=values.device).index_add(0, index, torch.where(mask, values, torch.zeros_like(values)))
    c = torch.zeros(n_graphs, dtype=values.dtype, device=values.device).index_add(0, index, w)
    n_eff = torch.where(mask, index, torch.zeros_like(index)).max() + 1
    return (s / c.clamp(min=1)).sum() / n_eff.to(values.dtype)


class CTNVPScheduler(VPSchedule):
    """Continuous (position) schedule; the posterior step is ``backward_remove_noise``."""

    def forward_add_noise(self, x, t, batch_idx, gen_flag, noise=None, zero_center=False):
        """q(x_t | x_0) (diffusion_scheduler.py:117-134): returns (x_t, noise); gen_flag=False rows keep x."""
        if noise is None:
            noise = torch.randn_like(x)
        a = self.alphas_cumprod.index_select(0, t)[batch_idx].unsqueeze(-1)
        x_noisy = a.sqrt() * x + (1.0 - a).sqrt() * noise
        out = torch.where(gen_flag.unsqueeze(-1), x_noisy, x)
        if zero_center:
            com = scatter_mean(noise, batch_idx, int(t.shape[0]))[batch_idx]
            return out, noise - com, com
        return out, noise

    def get_score_loss(self, pred, tgt, t, gen_flag, batch_idx, score_in=False, info_tag=None):
        """noise-prediction loss of DiffBP (diffusion_scheduler.py:203-218)"""
        a = self.alphas_cumprod.index_select(0, t)[batch_idx][:, None].expand_as(pred)
        sigma = (1 - a).sqrt()
        noise = tgt / sigma if score_in else tgt
        mse = ((pred - noise) ** 2).sum(-1)
        loss = masked_graph_mean(mse, batch_idx, gen_flag, int(t.shape[0]))
        info = {"eps_0": noise, "eps_pred": pred, "score_0": noise * sigma, "score_pred": pred * sigma, "mask_gen": gen_flag}
        if info_tag is not None:
            info = {k + "_{}".format(info_tag): v for k, v in info.items()}
        return loss, info

    def xs_mean(self, x_pred, x_noisy, t, batch_idx, gen_flag, type="score"):
        """mean of the reverse step (diffusion_scheduler.py:166-183)"""
        a = self.alphas_cumprod.index_select(0, t)[:, None][batch_idx].expand_as(x_noisy)
        b = self.betas.index_select(0, t)[:, None][batch_idx].expand_as(x_noisy)
        if type == "score":
            xs = (x_noisy + b * (-x_pred / (1 - a).sqrt())) / (1 - b).sqrt()
        else:
            xs = (self.posterior_mean_c0_coef[t][batch_idx][:, None] * x_pred
                  + self.posterior_mean_ct_coef[t][batch_idx][:, None] * x_noisy)
        return torch.where(gen_flag.unsqueeze(-1), xs, x_noisy)

    def get_loss(self, x_pred, x0, xt, t, gen_flag, batch_idx, type="score"):
        """per-graph mean squared error, averaged over graphs (diffusion_scheduler.py:185-201)."""
        if type == "score":
            a = self.alphas_cumprod.index_select(0, t)[batch_idx][:, None].expand_as(x_pred)
            tgt = (x0 - xt) / (1 - a).sqrt()
        else:
            tgt = x0
        mse = ((x_pred - tgt) ** 2).sum(-1)
        return masked_graph_mean(mse, batch_idx, gen_flag, int(t.shape[0])), {"x0": x0, "xt": xt, "x_pred": x_pred, "mask_gen": gen_flag}

    def backward_remove_noise(self, x_pred, x_noisy, t, batch_idx, gen_flag, type="score", noise=None):
        """x_{t-1} from x_t (diffusion_scheduler.py:144-165).  type='denoise': x_pred is x0, sample the Gaussian
        posterior q(x_{t-1} | x_t, x0); type='score' (the reference's default; DiffBP): x_pred is the noise
        estimate, ancestral step (x_t - beta * eps / sqrt(1 - abar)) / sqrt(1 - beta) + sqrt(beta) * z."""
        tb = t[batch_idx]
        if noise is None:
            noise = torch.randn_like(x_noisy)
        nonzero = (tb != 0).to(x_noisy.dtype)[:, None]
        if type == "score":
            a = self.alphas_cumprod[tb][:, None]
            b = self.betas[tb][:, None]
            score = -x_pred / (1 - a).sqrt()
            xs = (x_noisy + b * score) / (1 - b).sqrt()
            xs = xs + nonzero * b.sqrt() * noise
        else:
            mean = (self.posterior_mean_c0_coef[tb][:, None] * x_pred
                    + self.posterior_mean_ct_coef[tb][:, None] * x_noisy)
            xs = mean + nonzero * (0.5 * self.posterior_logvar[tb][:, None]).exp() * noise
        return torch.where(gen_flag[:, None], xs, x_noisy)


def _index_to_log_onehot(v, num_classes):
    return torch.log(F.one_hot(v, num_classes).float().clamp(min=1e-30))   # models/utils/categorical.py:5-11


def _log_add_exp(a, b):
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


class TypeVPScheduler(VPSchedule):
    """Categorical (atom type) schedule (diffusion_scheduler.py:320-441)."""

    def __init__(self, num_timestep, num_classes, beta_start=1e-7, beta_end=2e-3, type="sigmoid", cosine_s=0.008):
        super().__init__(num_timestep, beta_start, beta_end, type, cosine_s)
        self.num_classes = num_classes
        log_alphas_v = np.log(self.alphas.numpy())          # fp32 like the reference
        log_ac = np.cumsum(log_alphas_v)

        def log_1_min_a(a):
            return np.log(1 - np.exp(a) + 1e-40)

        self.log_alphas_v = _frozen(log_alphas_v)
        self.log_one_minus_alphas_v = _frozen(log_1_min_a(log_alphas_v))
        self.log_alphas_cumprod_v = _frozen(log_ac)
        self.log_one_minus_alphas_cumprod_v = _frozen(log_1_min_a(log_ac))

    # ---- training side (diffusion_scheduler.py:339-365, 380-441) ----
    def q_v_pred(self, log_v0, t, batch):
        return _log_add_exp(log_v0 + self.log_alphas_cumprod_v[t][batch].unsqueeze(-1),
                            self.log_one_minus_alphas_cumprod_v[t][batch].unsqueeze(-1) - math.log(self.num_classes))

    def q_v_pred_one_timestep(self, log_vt_1, t, batch):
        return _log_add_exp(log_vt_1 + self.log_alphas_v[t][batch].unsqueeze(-1),
                            self.log_one_minus_alphas_v[t][batch].unsqueeze(-1) - math.log(self.num_classes))

    def q_v_posterior(self, log_v0, log_vt, t, batch):
        tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)
        un = self.q_v_pred(log_v0, tm1, batch) + self.q_v_pred_one_timestep(log_vt, t, batch)
        return un - torch.logsumexp(un, dim=-1, keepdim=True)

    def forward_add_noise(self, v0, t, batch_idx, gen_flag, uniform=None):
        """v_t ~ q(v_t | v_0) by Gumbel-argmax; returns (one-hot float, index)."""
        log_q = self.q_v_pred(_index_to_log_onehot(v0, self.num_classes), t, batch_idx)
        if uniform is None:
            uniform = torch.rand_like(log_q)
        gumbel = -torch.log(-torch.log(uniform + 1e-30) + 1e-30)
        v_noisy = torch.where(gen_flag, (gumbel + log_q).argmax(-1), v0)
        return F.one_hot(v_noisy, self.num_classes).float(), v_noisy

    def compute_loss(self, log_c_pred_prob, log_c_true_prob, log_v0, t, batch, mask_generate):
        kl = (log_c_true_prob.exp() * (log_c_true_prob - log_c_pred_prob)).sum(1)
        nll = -(log_v0.exp() * log_c_pred_prob).sum(1)
        mask = (t == 0).float()[batch]
        return masked_graph_mean(mask * nll + (1.0 - mask) * kl, batch, mask_generate, int(t.shape[0]))

    def get_loss(self, c_pred, v0, vt, t, gen_flag, batch_idx, pred_logit=True):
        log_c0 = _index_to_log_onehot(v0, self.num_classes)
        log_ct = _index_to_log_onehot(vt, self.num_classes)
        log_c_pred = F.log_softmax(c_pred, dim=-1) if pred_logit else torch.log(c_pred + 1e-8)
        log_p = self.q_v_posterior(log_c_pred, log_ct, t, batch_idx)
        log_q = self.q_v_posterior(log_c0, log_ct, t, batch_idx)
        loss = self.compute_loss(log_p, log_q, log_c0, t, batch_idx, gen_flag)
        return loss, {"v0": v0, "vt": vt, "c_pred": log_c_pred.exp(), "mask_gen": gen_flag}

    def backward_remove_noise(self, c_pred, ct, t, batch_idx, gen_flag, pred_logit=True, uniform=None):
        """v_{t-1} ~ q(v_{t-1} | v_t, v0_pred) by Gumbel-argmax (diffusion_scheduler.py:367-378, 407-441;
        models/utils/categorical.py:26-32). Returns (one-hot, index)."""
        log_c_pred = F.log_softmax(c_pred, dim=-1) if pred_logit else torch.log(c_pred + 1e-8)
        log_ct = torch.log(ct + 1e-8)
        tb = t[batch_idx]
        tm1 = torch.clamp(tb - 1, min=0)
        lc = math.log(self.num_classes)
        log_q0 = _log_add_exp(log_c_pred + self.log_alphas_cumprod_v[tm1][:, None],
                              self.log_one_minus_alphas_cumprod_v[tm1][:, None] - lc)
        log_q1 = _log_add_exp(log_ct + self.log_alphas_v[tb][:, None],
                              self.log_one_minus_alphas_v[tb][:, None] - lc)
        un = log_q0 + log_q1
        logp = un - torch.logsumexp(un, dim=-1, keepdim=True)
        if uniform is None:
            uniform = torch.rand_like(logp)
        gumbel = -torch.log(-torch.log(uniform + 1e-30) + 1e-30)
        v_next = (gumbel + logp).argmax(dim=-1)
        v_next = torch.where(gen_flag, v_next, ct.argmax(-1))
        return F.one_hot(v_next, self.num_classes).to(ct.dtype), v_next


def _native_noise(pos_sched, type_sched, x0, v0, t, bl, gen_l, eps, u):
    """q(x_t | x_0) and q(v_t | v_0) of ``CTNVPScheduler.forward_add_noise`` / ``TypeVPScheduler.forward_add_noise`` in ONE launch
    (csrc/train_loss.hip, cbgx_targetdiff_train_noise).  The draws are made here, in the order and shapes of the tensor path, so a
    seeded run sees the same noise on both."""
    C = type_sched.num_classes
    if eps is None:
        eps = torch.randn_like(x0)
    if u is None:
        u = torch.rand(x0.shape[0], C, dtype=torch.float32, device=x0.device)
    eps, u = eps.float().contiguous(), u.float().contiguous()
    x_t = torch.empty_like(x0)
    c_t = torch.empty(x0.shape[0], C, dtype=torch.float32, device=x0.device)
    v_t = torch.empty_like(v0)
    _native.check(_native.lib().cbgx_targetdiff_train_noise(
        _native.ptr(x0), _native.ptr(v0), _native.ptr(t), _native.ptr(bl), _native.ptr(gen_l), x0.shape[0], C,
        _native.ptr(pos_sched.alphas_cumprod), _native.ptr(type_sched.log_alphas_cumprod_v),
        _native.ptr(type_sched.log_one_minus_alphas_cumprod_v), _native.ptr(eps), _native.ptr(u), _native.ptr(x_t),
        _native.ptr(c_t), _native.ptr(v_t), _native.current_stream(x0.device)), "cbgx_targetdiff_train_noise")
    return x_t, c_t, v_t


class _TargetDiffLossFunction(torch.autograd.Function):
    """``CTNVPScheduler.get_loss(type='denoise')`` + ``TypeVPScheduler.get_loss`` on the ligand rows of the denoiser outputs
    (targetdiff.py:103-121) as one launch, which also leaves the gradients of both losses with respect to those rows; the backward
    is one more launch that scatters them, scaled by the upstream gradients, into full-size dL/dx_out and dL/dlogits.  The tensor
    path (``get_loss(..., fused=False)``) takes ~220 small launches for the same numbers."""

    @staticmethod
    def forward(ctx, xo, logits, lig_rows, sort_idx, n_rec, x0, v0, vt, t, bl, gen_l, tables):
        dev = xo.device
        n_lig, C, B = x0.shape[0], logits.shape[1], t.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        losses = torch.empty(2, **f32)
        x_pred, c_pred = torch.empty(n_lig, 3, **f32), torch.empty(n_lig, C, **f32)
        gpos, gz = torch.empty(n_lig, 3, **f32), torch.empty(n_lig, C, **f32)
        arr = (ctypes.c_void_p * 4)(*[tb.data_ptr() for tb in tables])
        _native.check(_native.lib().cbgx_targetdiff_loss(
            _native.ptr(xo), _native.ptr(logits), _native.ptr(lig_rows), _native.ptr(x0), _native.ptr(v0), _native.ptr(vt),
            _native.ptr(t), _native.ptr(bl), _native.ptr(gen_l), n_lig, B, C, arr, _native.ptr(losses), _native.ptr(x_pred),
            _native.ptr(c_pred), _native.ptr(gpos), _native.ptr(gz), _native.current_stream(dev)), "cbgx_targetdiff_loss")
        ctx.saved = (gpos, gz, sort_idx)
        ctx.dims = (int(n_rec), xo.shape[0], C)
        ctx.mark_non_differentiable(x_pred, c_pred)
        loss_pos, loss_atom = losses.unbind(0)
        return loss_pos, loss_atom, x_pred, c_pred

    @staticmethod
    def backward(ctx, g_pos, g_atom, _gx, _gc):
        gpos, gz, sort_idx = ctx.saved
        n_rec, N, C = ctx.dims
        dev = gpos.device
        cont = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_pos, g_atom = cont(g_pos), cont(g_atom)
        grad_x = torch.empty(N, 3, dtype=torch.float32, device=dev)
        grad_logits = torch.empty(N, C, dtype=torch.float32, device=dev)
        _native.check(_native.lib().cbgx_targetdiff_loss_backward(
            _native.ptr(gpos), _native.ptr(gz), _native.ptr(sort_idx), n_rec, N, C, _native.ptr(g_pos), _native.ptr(g_atom),
            _native.ptr(grad_x), _native.ptr(grad_logits), _native.current_stream(dev)), "cbgx_targetdiff_loss_backward")
        return (grad_x, grad_logits) + (None,) * 10


class PLContextEmbedder(nn.Module):
    """Linear atom / residue / ligand-indicator embeddings (repo/modules/context_emb.py:137-230) for the
    shipped embedder config (atom: linear, residue: linear; no time, no vec)."""

    def __init__(self, cfg):
        super().__init__()
        self.num_classes = cfg.get("num_atomtype", 14)
        emb_dim = cfg.get("emb_dim", 128)
        self.emb_dim = emb_dim
        for key in ("time", "vec"):
            if cfg.get(key, None) is not None:
                raise ValueError(f"embedder.{key} is not used by any shipped diffusion config and is not supported")
        atom, res = cfg.get("atom", None), cfg.get("residue", None)
        if atom is None or atom.type != "linear" or res is None or res.type != "linear":
            raise ValueError("embedder.atom / embedder.residue must be {type: linear}")
        self.ligand_atom_emb = nn.Linear(self.num_classes, emb_dim)
        self.protein_atom_emb = nn.Linear(PROTEIN_FEAT, emb_dim)
        self.residue_emb = nn.Linear(NUM_AA, emb_dim)
        self.ligand_indicator = nn.Linear(1, emb_dim)

    def embed_protein(self, v_rec, aa_rec_onehot):
        ind0 = self.ligand_indicator.bias  # indicator(0) = bias
        return self.protein_atom_emb(v_rec) + self.residue_emb(aa_rec_onehot) + ind0

    def embed_ligand(self, c_lig):
        ind1 = self.ligand_indicator.weight[:, 0] + self.ligand_indicator.bias
        return self.ligand_atom_emb(c_lig) + ind1


class _ComposeEmbedFunction(torch.autograd.Function):
    """``PLContextEmbedder`` on both atom sets + ``compose_context`` of coordinates, features and the movable flag
    (repo/modules/context_emb.py:137-230, repo/modules/common.py:189-214) as one launch; the backward is the weight gradient of ONE
    Linear over "extended input" rows the forward leaves behind (csrc/train_embed.hip, cbgx_embed_compose{,_backward}) -- two launches
    instead of index_put's sort, three thin GEMMs and five column sums."""

    @staticmethod
    def forward(ctx, x_rec, x_lig, feat, aa, c_lig, sort_idx, gen_r, gen_l, w_pa, b_pa, w_res, b_res, w_la, b_la, w_ind, b_ind):
        dev = x_rec.device
        n_rec, n_lig = x_rec.shape[0], x_lig.shape[0]
        N, E = n_rec + n_lig, w_pa.shape[0]
        dims = (w_pa.shape[1], w_res.shape[1], w_la.shape[1])
        x = torch.empty(N, 3, dtype=torch.float32, device=dev)
        h = torch.empty(N, E, dtype=torch.float32, device=dev)
        ext = torch.empty(N, 128, dtype=torch.float32, device=dev)
        gen = torch.empty(N, dtype=torch.bool, device=dev)
        params = [p.detach().contiguous() for p in (w_pa, b_pa, w_res, b_res, w_la, b_la, w_ind, b_ind)]
        arr = (ctypes.c_void_p * 8)(*[p.data_ptr() for p in params])
        _native.check(_native.lib().cbgx_embed_compose(
            _native.ptr(x_rec), _native.ptr(x_lig), _native.ptr(feat), _native.ptr(aa), _native.ptr(c_lig), _native.ptr(sort_idx),
            _native.ptr(gen_r), _native.ptr(gen_l), n_rec, n_lig, dims[0], dims[1], dims[2], arr, _native.ptr(x), _native.ptr(h),
            _native.ptr(ext), _native.ptr(gen), _native.current_stream(dev)), "cbgx_embed_compose")
        ctx.save_for_backward(ext)
        ctx.dims = dims
        ctx.mark_non_differentiable(x, gen)
        return x, h, gen

    @staticmethod
    def backward(ctx, _gx, gh, _gg):
        (ext,) = ctx.saved_tensors
        Fd, A, C = ctx.dims
        N, dev = ext.shape[0], ext.device
        groups = max(1, min(64, (N + 127) // 128))
        partial = torch.empty(groups * 128 * 128, dtype=torch.float32, device=dev)
        out = torch.empty(128 * (Fd + A + C + 2), dtype=torch.float32, device=dev)
        _native.check(_native.lib().cbgx_embed_compose_backward(
            _native.ptr(gh.to(torch.float32).contiguous()), _native.ptr(ext), N, Fd, A, C, _native.ptr(partial), groups,
            _native.ptr(out), _native.current_stream(dev)), "cbgx_embed_compose_backward")
        dw_pa, dw_res, u, dw_la, v = out.split([128 * Fd, 128 * A, 128, 128 * C, 128])
        return (None,) * 8 + (dw_pa.view(128, Fd), u, dw_res.view(128, A), u, dw_la.view(128, C), v, v.view(128, 1), u + v)


def compose_embed(embedder, x_rec, x_lig, feat_rec, aa_rec, c_lig, sort_idx, gen_r, gen_l, fused=True):
    """(x, h, gen_flag) of a training step in composed row order: ``cat(protein, ligand)[sort_idx]`` of the coordinates, of
    ``embed_protein(feat_rec, one_hot(aa_rec))`` / ``embed_ligand(c_lig)`` and of the movable flags (targetdiff.py:89-101 and the same
    lines of the other two model classes).  One launch when the inputs are in the shape the kernel takes (``CBGX_FUSED_EMBED=0``: the
    tensor path, which is also the CPU path)."""
    e = embedder
    n_in = e.protein_atom_emb.in_features + e.residue_emb.in_features + e.ligand_atom_emb.in_features + 2
    if (fused and x_rec.is_cuda and e.emb_dim == 128 and n_in <= 120 and os.environ.get("CBGX_FUSED_EMBED", "1") != "0"
            and aa_rec.dtype == torch.int64 and sort_idx.dtype == torch.int64 and gen_l.dtype == torch.bool
            and gen_r.dtype == torch.bool and all(t.dtype == torch.float32 for t in (x_rec, x_lig, feat_rec, c_lig))
            and feat_rec.shape[1] == e.protein_atom_emb.in_features and c_lig.shape[1] == e.ligand_atom_emb.in_features
            and e.protein_atom_emb.weight.dtype == torch.float32):
        return _ComposeEmbedFunction.apply(
            x_rec.contiguous(), x_lig.contiguous(), feat_rec.contiguous(), aa_rec.contiguous(), c_lig.contiguous(),
            sort_idx.contiguous(), gen_r.contiguous(), gen_l.contiguous(), e.protein_atom_emb.weight, e.protein_atom_emb.bias,
            e.residue_emb.weight, e.residue_emb.bias, e.ligand_atom_emb.weight, e.ligand_atom_emb.bias, e.ligand_indicator.weight,
            e.ligand_indicator.bias)
    aa = F.one_hot(aa_rec, e.residue_emb.in_features).float()
    h = torch.cat([e.embed_protein(feat_rec, aa), e.embed_ligand(c_lig)], 0)[sort_idx]
    return torch.cat([x_rec, x_lig], 0)[sort_idx], h, torch.cat([gen_r, gen_l], 0)[sort_idx]


class _Lap:
    """``lap(key)``: add the wall seconds since the previous lap to ``timing[key]`` after a device synchronisation; a no-op
    without a timing dict."""

    def __init__(self, timing, dev):
        self.timing, self.dev = timing, dev
        if timing is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            self.t = time.perf_counter()

    def __call__(self, key):
        if self.timing is None:
            return
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        now = time.perf_counter()
        self.timing[key] = self.timing.get(key, 0.0) + (now - self.t)
        self.t = now


class BatchesInFlight:
    """``sample_many`` for the sampler classes (TargetDiff, DiffBP, DiffSBDD): ``[self.sample(b) for b in batches]`` with the batches
    IN FLIGHT TOGETHER, round-robin over ``streams`` HIP streams.  The batches are independent (the pocket loop of sample.py:159),
    and while one batch sits in its matrix-bound edge kernel the HBM-bound node kernels and the small kernels of the others fill what
    it leaves (measured +5.5 % with three 200-graph batches).  Each batch's steps stay ordered on its own stream.  With
    ``noise_tapes`` (one per batch, in the form the class's ``sample`` takes) the trajectories equal ``sample``'s bit for bit; without,
    the torch generator is consumed step by step across the batches instead of batch by batch, i.e. the same distribution under a
    different assignment of the draws.  A class provides ``_many_begin(batch, tape) -> state``, ``_many_step(state, t, tape)`` and
    ``_many_finish(state, device) -> trajectory``."""

    def _side_streams(self, dev, n):
        """the model's own side streams, created once per device: the denoiser keeps one workspace per stream (gigabytes at 100 k
        nodes) and libcbgx one auxiliary stream per caller stream, so the handles must not change from call to call"""
        pool = self.__dict__.setdefault("_stream_pool", {})
        have = pool.setdefault(str(dev), [])
        while len(have) < n:
            have.append(torch.cuda.Stream(dev))
        return have[:n]

    @torch.no_grad()
    def sample_many(self, batches, noise_tapes=None, return_device=None, streams=3, use_graph=False, noise_log=None, timing=None):
        """``use_graph`` (classes with ``make_step_graph``; ignored with ``noise_tapes``): every batch's step is captured once as a
        hipGraph on its stream and replayed T times -- one host call per batch and step instead of ~50 launches plus the Python
        around them.  This is what SMALL batches need (the reference's own 10 graphs per batch, sample.py:177-183): one such batch
        is a chain of ~50 dependent 20 us kernels that leaves most of the chip idle, and the host cannot feed eight of them at once;
        eight graphs in flight can.  Same kernels on the same data as the stream launches: identical trajectories
        for identical noise.  ``noise_log`` (tests): a list that receives ``(batch index, t, eps, u)`` after every replay (which
        synchronises the stream).  ``timing`` (a dict): the wall seconds of the three parts of the call are ADDED to its keys
        ``begin_sampling_s`` (static context of every batch), ``steps_s`` (the T steps) and ``traj_download_s`` (trajectory to
        ``return_device``) -- three device synchronisations per call."""
        dev = batches[0]["ligand_pos"].device
        lap = _Lap(timing, dev)
        out_dev = torch.device("cpu") if return_device is None else torch.device(return_device)
        tape = lambda k: None if noise_tapes is None else noise_tapes[k]
        T = self.num_diffusion_timesteps
        graphs = bool(use_graph) and noise_tapes is None and dev.type == "cuda" and hasattr(self, "make_step_graph")
        if dev.type != "cuda" or ((len(batches) == 1 or streams <= 1) and not graphs):
            out = []
            for k, b in enumerate(batches):
                st = self._many_begin(b, tape(k))
                lap("begin_sampling_s")
                for t_idx in reversed(range(T)):
                    self._many_step(st, t_idx, tape(k))
                lap("steps_s")
                out.append(self._many_finish(st, out_dev))
                lap("traj_download_s")
            return out
        states = [self._many_begin(b, tape(k)) for k, b in enumerate(batches)]
        lap("begin_sampling_s")
        cur = torch.cuda.current_stream(dev)
        side = self._side_streams(dev, max(1, min(streams, len(states))))
        for sx in side:
            sx.wait_stream(cur)                      # the states were built on the caller's stream
        if graphs:
            made = [self.make_step_graph(st, stream=side[k % len(side)]) for k, st in enumerate(states)]
            for n in range(T):
                for k, (st, (replay, done)) in enumerate(zip(states, made)):
                    if n < T - done:
                        with torch.cuda.stream(side[k % len(side)]):
                            replay()
                            if noise_log is not None:
                                side[k % len(side)].synchronize()
                                noise_log.append((k, T - 1 - done - n, st["_noise"][0].clone(), st["_noise"][1].clone()))
        else:
            for t_idx in reversed(range(T)):
                for k, st in enumerate(states):
                    with torch.cuda.stream(side[k % len(side)]):
                        self._many_step(st, t_idx, tape(k))
        for sx in side:
            cur.wait_stream(sx)
        lap("steps_s")
        out = [self._many_finish(st, out_dev) for st in states]
        lap("traj_download_s")
        return out


@register_model("targetdiff")
class TargetDiff(BatchesInFlight, nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        gen = cfg.generator
        self.num_diffusion_timesteps = gen.num_diffusion_timesteps
        self.denoise_structure = gen.get("denoise_structure", True)
        self.denoise_atom = gen.get("denoise_atom", True)
        self.time_sampler = gen.get("time_sampler", "symmetric")
        self.num_classes = cfg.num_atomtype
        ps = gen.pos_schedule
        self.pos_scheduler = CTNVPScheduler(self.num_diffusion_timesteps, beta_start=ps.beta_start,
                                            beta_end=ps.beta_end, type=ps.type)
        at = gen.atom_schedule
        self.type_scheduler = TypeVPScheduler(self.num_diffusion_timesteps, num_classes=self.num_classes,
                                              type=at.type, cosine_s=at.cosine_s)
        cfg.embedder.num_atomtype = cfg.num_atomtype
        if cfg.embedder.get("type", "fa") != "fa":
            raise ValueError("only the full-atom context embedder ('fa') is supported")
        self.context_embedder = PLContextEmbedder(cfg.embedder)
        self.denoiser = get_e3_gnn(cfg.encoder, num_classes=self.num_classes)
        # forward noising and the two losses as single launches of libcbgx (csrc/train_loss.hip) instead of tensor operations;
        # False keeps the tensor path (the restatement of the reference's schedulers above), which the tests compare against
        # (CBGX_FUSED_TRAINING_OPS=0 in the environment: A/B runs of bench.py --workload train)
        self.fused_training_ops = os.environ.get("CBGX_FUSED_TRAINING_OPS", "1") != "0"

    # ---- training (targetdiff.py:40-124) ---------------------------------------------------------
    def sample_time(self, batch_size, device="cuda", draws=None):
        """_base.py:13-33; ``draws`` replays the integers of the 'symmetric' sampler in tests."""
        T = self.num_diffusion_timesteps
        if self.time_sampler == "symmetric":
            if draws is None:
                draws = torch.randint(0, T, size=(batch_size // 2 + 1,), device=device)
            draws = draws.to(device)
            return torch.cat([draws, T - draws - 1], 0)[:batch_size]
        if self.time_sampler == "uniform":
            return torch.round((torch.rand(batch_size) * T).clip(0, T - 1)).long().to(device)
        raise ValueError(f"time_sampler {self.time_sampler!r} is not supported")

    def forward(self, batch, t=None, noise=None):
        """``loss_dict, results = model(batch)`` of train.py:185.  Training mode: one sampled time per graph;
        eval mode: the average over ``eval_interval`` evenly spaced times (targetdiff.py:62-78).  ``t`` /
        ``noise=(eps, u)`` replay the random draws in tests."""
        bl = batch["ligand_element_batch"]
        # the graph count: from the batch if the collate recorded it (no host synchronisation in the training step), else as the
        # reference computes it
        B = int(batch["num_graphs"]) if "num_graphs" in batch else (int(t.shape[0]) if t is not None else int(bl.max().item()) + 1)
        dev = batch["ligand_pos"].device
        if self.training or t is not None:
            if t is None:
                t = self.sample_time(B, device=dev)
            return self.get_loss(batch, t, noise)
        dicts, results = [], []
        for tv in np.linspace(0, self.num_diffusion_timesteps - 1, self.cfg.get("eval_interval", 10)):
            ld, res = self.get_loss(batch, torch.tensor([tv] * B).long().to(dev), None)
            dicts.append(ld)
            results.append(res)
        mean = {k: torch.stack([d[k] for d in dicts]).mean() for k in dicts[0]}
        return mean, results

    def get_loss(self, batch, t, noise=None):
        """targetdiff.py:82-124: forward noising, embedding + composition, denoiser, position / type losses."""
        x0 = batch["ligand_pos"].float()
        v0 = batch["ligand_atom_type"]
        x_rec = batch["protein_pos"].float()
        lig_flag_l = batch["ligand_lig_flag"]
        gen_l = batch.get("ligand_gen_flag", lig_flag_l).bool()
        gen_r = batch.get("protein_gen_flag", torch.zeros_like(batch["protein_lig_flag"])).bool()
        bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
        eps, u = noise if noise is not None else (None, None)
        # one launch for the noising, one for both losses (+ one in the backward) when everything is in the shape the kernels take
        fused = (self.fused_training_ops and x0.is_cuda and self.denoise_structure and self.denoise_atom
                 and self.num_classes <= 32 and 0 < int(t.shape[0]) <= 4096 and 0 < x0.shape[0] <= 65536
                 and v0.dtype == torch.int64 and t.dtype == torch.int64 and bl.dtype == torch.int64)
        if fused:
            x0, v0, t, bl, gen_l = x0.contiguous(), v0.contiguous(), t.contiguous(), bl.contiguous(), gen_l.contiguous()
            x_t, c_t, v_t = _native_noise(self.pos_scheduler, self.type_scheduler, x0, v0, t, bl, gen_l, eps, u)
        else:
            x_t = self.pos_scheduler.forward_add_noise(x0, t, bl, gen_l, noise=eps)[0] if self.denoise_structure else x0
            if self.denoise_atom:
                c_t, v_t = self.type_scheduler.forward_add_noise(v0, t, bl, gen_l, uniform=u)
            else:
                c_t, v_t = F.one_hot(v0, self.num_classes).float(), v0
        sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = self.compose_plan(bl, br, int(t.shape[0]))
        x, h, gen_flag = compose_embed(self.context_embedder, x_rec, x_t, batch["protein_atom_feature"].float(), batch["protein_aa_type"],
                                       c_t, sort_idx, gen_r, gen_l, fused=self.fused_training_ops)
        xo, _, logits = self.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen_flag,
                                      graph_ptr=graph_ptr, ligand_outputs_only=True)
        if fused:
            ts = self.type_scheduler
            loss_pos, loss_atom, x_pred, c_prob = _TargetDiffLossFunction.apply(
                xo, logits, lig_rows.contiguous(), sort_idx.contiguous(), x_rec.shape[0], x0, v0, v_t, t, bl, gen_l,
                (ts.log_alphas_v, ts.log_one_minus_alphas_v, ts.log_alphas_cumprod_v, ts.log_one_minus_alphas_cumprod_v))
            results = {"x0": x0, "xt": x_t, "x_pred": x_pred, "mask_gen": gen_l, "v0": v0, "vt": v_t, "c_pred": c_prob}
            return {"pos": loss_pos, "atom": loss_atom}, results
        x_pred, c_pred = xo[lig_rows], logits[lig_rows]
        results = {}
        if self.denoise_structure:
            loss_pos, info = self.pos_scheduler.get_loss(x_pred, x0, x_t, t, gen_l, bl, type="denoise")
            results.update(info)
        else:
            loss_pos = torch.tensor(0.0, device=x0.device)
        if self.denoise_atom:
            loss_atom, info = self.type_scheduler.get_loss(c_pred, v0, v_t, t, gen_l, bl, pred_logit=True)
            results.update(info)
        else:
            loss_atom = torch.tensor(0.0, device=x0.device)
        return {"pos": loss_pos, "atom": loss_atom}, results

    # ---- static per-batch structure ------------------------------------------------------------
    @staticmethod
    def compose_plan(batch_idx_lig, batch_idx_rec, n_graphs=None):
        """compose_context (repo/modules/common.py:189-214): cat(rec, lig) + stable sort by graph id.
        Returns (sort_idx, batch_idx, lig_rows, graph_ptr); computed once per batch."""
        n_rec, n_lig = batch_idx_rec.shape[0], batch_idx_lig.shape[0]
        if (n_graphs is not None and batch_idx_rec.is_cuda and batch_idx_rec.dtype == torch.int64 and batch_idx_lig.dtype == torch.int64
                and 0 < n_graphs <= (1 << 20) and n_rec + n_lig > 0 and os.environ.get("CBGX_FUSED_COMPOSE", "1") != "0"):
            # the same five results from a counting sort in three launches (csrc/train_embed.hip, cbgx_compose_plan) instead of ~20
            dev = batch_idx_rec.device
            N = n_rec + n_lig
            sort_idx = torch.empty(N, dtype=torch.int64, device=dev)
            batch_idx = torch.empty(N, dtype=torch.int64, device=dev)
            lig_flag = torch.empty(N, dtype=torch.bool, device=dev)
            lig_rows = torch.empty(n_lig, dtype=torch.int64, device=dev)
            graph_ptr = torch.empty(n_graphs + 1, dtype=torch.int32, device=dev)
            scratch = torch.empty(4 * n_graphs + 1, dtype=torch.int32, device=dev)
            _native.check(_native.lib().cbgx_compose_plan(
                _native.ptr(batch_idx_rec.contiguous()), _native.ptr(batch_idx_lig.contiguous()), n_rec, n_lig, int(n_graphs),
                _native.ptr(scratch), _native.ptr(sort_idx), _native.ptr(batch_idx), _native.ptr(lig_flag), _native.ptr(lig_rows),
                _native.ptr(graph_ptr), _native.current_stream(dev)), "cbgx_compose_plan")
            return sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr
        batch_ctx = torch.cat([batch_idx_rec, batch_idx_lig], 0)
        sort_idx = torch.sort(batch_ctx, stable=True).indices
        batch_idx = batch_ctx[sort_idx]
        n_rec = batch_idx_rec.shape[0]
        lig_flag = sort_idx >= n_rec
        # composed rows of ligand atoms, in ligand order = the inverse permutation's tail (no nonzero(): that synchronises)
        inv = torch.empty_like(sort_idx)
        inv[sort_idx] = torch.arange(sort_idx.numel(), device=sort_idx.device)
        lig_rows = inv[n_rec:]
        return sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr_from_batch(batch_idx, n_graphs)

    @torch.no_grad()
    def begin_sampling(self, batch, keep_trajectory=True, static_cache=True):
        """Everything that is constant over the T steps of one batch: the composition permutation, CSR
        offsets, flags, the protein half of x / h, trajectory buffers.  Returns a state dict."""
        x_lig = batch["ligand_pos"].float()
        dev = x_lig.device
        v_lig_in = batch["ligand_atom_type"]
        x_rec = batch["protein_pos"].float()
        v_rec = batch["protein_atom_feature"].float()
        lig_flag_l = batch["ligand_lig_flag"]
        gen_l = batch.get("ligand_gen_flag", lig_flag_l).bool()
        gen_r = batch.get("protein_gen_flag", torch.zeros_like(batch["protein_lig_flag"])).bool()
        bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
        T, C = self.num_diffusion_timesteps, self.num_classes
        n_rec, n_lig = x_rec.shape[0], x_lig.shape[0]
        B = int(bl.max().item()) + 1
        aa = F.one_hot(batch["protein_aa_type"], NUM_AA).float()
        c_lig = F.one_hot(v_lig_in, C).float() if v_lig_in.dim() == 1 else v_lig_in.float()
        sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = self.compose_plan(bl, br, B)
        gen_flag = torch.cat([gen_r, gen_l], 0)[sort_idx]
        N = n_rec + n_lig
        rec_rows = torch.nonzero(~lig_flag).flatten()
        x = torch.empty(N, 3, dtype=torch.float32, device=dev)
        h = torch.empty(N, self.context_embedder.emb_dim, dtype=torch.float32, device=dev)
        x[rec_rows] = x_rec
        h[rec_rows] = self.context_embedder.embed_protein(v_rec, aa)
        st = dict(x=x, h=h, x_lig=x_lig.contiguous(), c_lig=c_lig.contiguous(), bl=bl, gen_l=gen_l, batch_idx=batch_idx,
                  lig_flag=lig_flag, gen_flag=gen_flag, lig_rows=lig_rows, graph_ptr=graph_ptr, B=B, n_lig=n_lig, N=N,
                  traj_x=None, traj_c=None)
        st["static_h"] = None
        if dev.type == "cuda" and static_cache and not bool(gen_r.any()):
            # protein rows of (x, h) are the same in all T denoiser calls: cache the ligand-free first two layers once
            st["static_h"] = self.denoiser.static_context(x_rec, h[rec_rows], br, rec_rows, N)
        if dev.type == "cuda":
            # operands of the native prologue / epilogue kernels (include/cbgx.h)
            st["lig_rows32"] = lig_rows.to(torch.int32).contiguous()
            st["gen_l8"] = gen_l.to(torch.uint8).contiguous()
            ps, ts = self.pos_scheduler, self.type_scheduler
            tabs = [ps.posterior_mean_c0_coef, ps.posterior_mean_ct_coef, ps.posterior_logvar, ts.log_alphas_v,
                    ts.log_one_minus_alphas_v, ts.log_alphas_cumprod_v, ts.log_one_minus_alphas_cumprod_v]
            st["tables"] = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in tabs])
        if keep_trajectory:
            # slot s+1 holds the state entering step s; slot 0 = final state (key -1 of the reference's dict)
            st["traj_x"] = torch.empty(T + 1, n_lig, 3, dtype=torch.float32, device=dev)
            st["traj_c"] = torch.empty(T + 1, n_lig, C, dtype=torch.float32, device=dev)
            st["traj_x"][T] = x_lig
            st["traj_c"][T] = c_lig
        return st

    @torch.no_grad()
    def denoise_step(self, st, t_idx, noise=None):
        """One iteration of the reverse-diffusion loop (targetdiff.py:150-182) on the sampling state.
        On the GPU the whole step is native: prologue kernel -> denoiser -> epilogue kernel (cbgx_targetdiff_*);
        only the noise draw is a torch call (or the replayed tape)."""
        dev = st["x"].device
        if dev.type == "cuda" and self.denoise_structure and self.denoise_atom:
            return self._denoise_step_native(st, t_idx, noise)
        t = torch.full((st["B"],), t_idx, dtype=torch.long, device=dev)
        x, h, lig_rows = st["x"], st["h"], st["lig_rows"]
        x[lig_rows] = st["x_lig"]
        h[lig_rows] = self.context_embedder.embed_ligand(st["c_lig"])
        xo, _, logits = self.denoiser(x=x, h=h, batch_idx=st["batch_idx"], lig_flag=st["lig_flag"],
                                      gen_flag=st["gen_flag"], graph_ptr=st["graph_ptr"])
        x_pred, c_pred = xo[lig_rows], logits[lig_rows]
        eps, u = noise if noise is not None else (None, None)
        if self.denoise_structure:
            st["x_lig"] = self.pos_scheduler.backward_remove_noise(x_pred, st["x_lig"], t, st["bl"], st["gen_l"],
                                                                   type="denoise", noise=eps)
        if self.denoise_atom:
            st["c_lig"], _ = self.type_scheduler.backward_remove_noise(c_pred, st["c_lig"], t, st["bl"], st["gen_l"],
                                                                       pred_logit=True, uniform=u)
        if st["traj_x"] is not None:
            st["traj_x"][t_idx] = st["x_lig"]
            st["traj_c"][t_idx] = st["c_lig"]
        return st

    NOISE_CHUNK = 16
    fuse_step_boundary = True      # epilogue(t) + prologue(t - 1) as one kernel (False: the two kernels, for tests)

    def _step_noise(self, st, n_lig, C, dev):
        """(eps [n_lig,3] ~ N(0, I), u [n_lig,C] ~ U(0, 1)) of one step.  The reference draws randn_like(x_lig) then
        rand_like(log-probs) in every step (targetdiff.py:168-175 -> diffusion_scheduler.py:163, categorical.py:27); the same
        two generators are drawn here for NOISE_CHUNK steps at a time and handed out step by step: two launches per chunk
        instead of two per step (10 us of a 900 us one-graph step).  Fresh numbers every step, the same distributions; the order
        in which the global generator is consumed differs from per-step draws (tests replay noise through ``noise=``).
        SEED COMPATIBILITY: with NOISE_CHUNK > 1 a fixed torch seed gives a different (equally distributed) trajectory than the
        reference's per-step draw order, than ``use_graph=True`` (``_traj_step`` draws per step) and than rounds before 5; up to
        NOISE_CHUNK - 1 unused steps of noise are drawn at the end of a run.  ``model.NOISE_CHUNK = 1`` restores the reference's
        draw order exactly (randn of step t, rand of step t, randn of step t - 1, ...), identical to the hipGraph path."""
        q = st.get("_noise_q")
        if q is None or q[2] >= q[0].shape[0] or q[0].shape[1] != n_lig:
            q = [torch.randn(self.NOISE_CHUNK, n_lig, 3, dtype=torch.float32, device=dev),
                 torch.rand(self.NOISE_CHUNK, n_lig, C, dtype=torch.float32, device=dev), 0]
            st["_noise_q"] = q
        k = q[2]
        q[2] = k + 1
        return q[0][k], q[1][k]

    def _denoise_step_native(self, st, t_idx, noise):
        lib = _native.lib()
        dev = st["x"].device
        stream = _native.current_stream(dev)
        n_lig, C = st["n_lig"], self.num_classes
        x_lig, c_lig = st["x_lig"], st["c_lig"]
        emb = self.context_embedder
        # The composed rows of the ligand atoms are already in place when the previous step's boundary kernel wrote them from
        # exactly this state (same tensor objects, not written since: identity + version counters); otherwise -- the first step
        # of a run, a caller that replaced or edited st["x_lig"] / st["c_lig"] -- the prologue kernel composes them.
        comp = st.get("_composed")
        if not (self.fuse_step_boundary and comp is not None and comp[0] is x_lig and comp[1] is c_lig
                and comp[2] == _native.version(x_lig) and comp[3] == _native.version(c_lig)):
            _native.check(lib.cbgx_targetdiff_prologue(
                _native.ptr(x_lig), _native.ptr(c_lig), _native.ptr(st["lig_rows32"]), n_lig, C,
                _native.ptr(emb.ligand_atom_emb.weight), _native.ptr(emb.ligand_atom_emb.bias),
                _native.ptr(emb.ligand_indicator.weight), _native.ptr(emb.ligand_indicator.bias),
                _native.ptr(st["x"]), _native.ptr(st["h"]), stream), "cbgx_targetdiff_prologue")
        xo, _, logits = self.denoiser(x=st["x"], h=st["h"], batch_idx=st["batch_idx"], lig_flag=st["lig_flag"],
                                      gen_flag=st["gen_flag"], graph_ptr=st["graph_ptr"], need_h=False,
                                      static_h=st["static_h"])
        if noise is not None:
            eps, u = noise[0].float().contiguous(), noise[1].float().contiguous()
        else:
            eps, u = self._step_noise(st, n_lig, C, dev)
        if st["traj_x"] is not None:
            x_next, c_next = st["traj_x"][t_idx], st["traj_c"][t_idx]
        else:
            x_next, c_next = torch.empty_like(x_lig), torch.empty_like(c_lig)
        if self.fuse_step_boundary and t_idx > 0:
            # epilogue of this step + prologue of the next in one launch (cbgx_targetdiff_step_boundary: same arithmetic)
            _native.check(lib.cbgx_targetdiff_step_boundary(
                _native.ptr(xo), _native.ptr(logits), _native.ptr(st["lig_rows32"]), _native.ptr(x_lig), _native.ptr(c_lig),
                _native.ptr(st["gen_l8"]), n_lig, C, int(t_idx), self.num_diffusion_timesteps, st["tables"],
                _native.ptr(eps), _native.ptr(u), _native.ptr(x_next), _native.ptr(c_next),
                _native.ptr(emb.ligand_atom_emb.weight), _native.ptr(emb.ligand_atom_emb.bias),
                _native.ptr(emb.ligand_indicator.weight), _native.ptr(emb.ligand_indicator.bias),
                _native.ptr(st["x"]), _native.ptr(st["h"]), stream), "cbgx_targetdiff_step_boundary")
            st["_composed"] = (x_next, c_next, _native.version(x_next), _native.version(c_next))
        else:
            _native.check(lib.cbgx_targetdiff_epilogue(
                _native.ptr(xo), _native.ptr(logits), _native.ptr(st["lig_rows32"]), _native.ptr(x_lig), _native.ptr(c_lig),
                _native.ptr(st["gen_l8"]), n_lig, C, int(t_idx), self.num_diffusion_timesteps, st["tables"],
                _native.ptr(eps), _native.ptr(u), _native.ptr(x_next), _native.ptr(c_next), None, stream),
                "cbgx_targetdiff_epilogue")
            st["_composed"] = None
        st["x_lig"], st["c_lig"] = x_next, c_next
        return st

    # ---- one captured graph replayed for every step (small batches are launch-bound: ~100 short kernels per step) ----
    def _traj_step(self, st):
        """One step with every argument static: the ligand state lives in the trajectory slots, the step index on the
        device (cbgx_targetdiff_prologue_traj / epilogue_traj).  Capturable in a hipGraph."""
        lib = _native.lib()
        dev = st["x"].device
        stream = _native.current_stream(dev)
        n_lig, C = st["n_lig"], self.num_classes
        emb = self.context_embedder
        _native.check(lib.cbgx_targetdiff_prologue_traj(
            _native.ptr(st["traj_x"]), _native.ptr(st["traj_c"]), _native.ptr(st["t_dev"]), _native.ptr(st["lig_rows32"]),
            n_lig, C, _native.ptr(emb.ligand_atom_emb.weight), _native.ptr(emb.ligand_atom_emb.bias),
            _native.ptr(emb.ligand_indicator.weight), _native.ptr(emb.ligand_indicator.bias),
            _native.ptr(st["x"]), _native.ptr(st["h"]), stream), "cbgx_targetdiff_prologue_traj")
        xo, _, logits = self.denoiser(x=st["x"], h=st["h"], batch_idx=st["batch_idx"], lig_flag=st["lig_flag"],
                                      gen_flag=st["gen_flag"], graph_ptr=st["graph_ptr"], need_h=False,
                                      static_h=st["static_h"], workspace=st.get("_ws"))
        eps = torch.randn(n_lig, 3, dtype=torch.float32, device=dev)      # reference draw order: randn then rand
        u = torch.rand(n_lig, C, dtype=torch.float32, device=dev)
        _native.check(lib.cbgx_targetdiff_epilogue_traj(
            _native.ptr(xo), _native.ptr(logits), _native.ptr(st["lig_rows32"]), _native.ptr(st["traj_x"]),
            _native.ptr(st["traj_c"]), _native.ptr(st["gen_l8"]), n_lig, C, _native.ptr(st["t_dev"]), st["tables"],
            _native.ptr(eps), _native.ptr(u), stream), "cbgx_targetdiff_epilogue_traj")
        st["_noise"] = (eps, u)     # under capture these are the graph's static buffers: readable after every replay

    @torch.no_grad()
    def make_step_graph(self, st, warmup=2, stream=None):
        """Capture one reverse-diffusion step as a hipGraph.  ``st`` must come from ``begin_sampling(keep_trajectory=True)``
        on the GPU.  Runs ``warmup`` eager steps first (they count: the step index advances), then returns
        ``(replay, steps_done)``; every ``replay()`` advances the state by one step on the CURRENT stream.  The step index lives
        in ``st['t_dev']``; after the last step the final state is trajectory slot 0.
        ``stream``: the (non-default) stream the warm-up steps run on and the capture is made on -- the one the replays will be
        issued on when several states are kept in flight (``sample_many(use_graph=True)``).  The state gets its own denoiser
        workspace (``st['_ws']``): the graph bakes every pointer in, and two graphs replayed on two streams must not share one."""
        dev = st["x"].device
        T = self.num_diffusion_timesteps
        if dev.type != "cuda" or st["traj_x"] is None or not (self.denoise_structure and self.denoise_atom):
            raise RuntimeError("make_step_graph needs a GPU sampling state with the trajectory kept on the device")
        run_on = stream if stream is not None else torch.cuda.current_stream(dev)
        done = min(warmup, T)
        with torch.cuda.stream(run_on):
            st["t_dev"] = torch.full((1,), T - 1, dtype=torch.int32, device=dev)
            st["_ws"] = torch.empty(self.denoiser.workspace_bytes(st["N"], st["B"]), dtype=torch.uint8, device=dev)
            self.denoiser.packed_weights(dev)   # built (and waited for, below) BEFORE the capture: a capturing stream must not wait
            for _ in range(done):               # on the pack's event; then the eager steps: the allocator warm
                self._traj_step(st)
        run_on.synchronize()
        if done >= T:
            return (lambda: None), done
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            self._traj_step(st)
        st["_graph"] = graph        # keeps the graph's private memory pool alive as long as the state
        return graph.replay, done

    @torch.no_grad()
    def sample(self, batch, noise_tape=None, return_device=None, use_graph=None, timing=None):
        """Reverse diffusion, T-1 .. 0 (targetdiff.py:127-184).

        ``noise_tape`` (tests): dict t -> (eps [N_lig,3], u [N_lig,C]) replacing the torch RNG draws
        (order per step in the reference: randn_like then rand_like).
        ``return_device``: where the returned trajectory lives (default: CPU, like the reference).
        ``use_graph``: replay one captured hipGraph per step instead of ~100 stream launches.  Off by default: measured on
        MI355X it changes nothing (1.22 ms per step at 445 nodes either way) -- small batches are bound by the
        dependent-kernel chain on the device, not by host launches.
        Fixed-seed reproducibility: see ``_step_noise`` (``NOISE_CHUNK``); works under ``torch.inference_mode()`` as well as
        ``torch.no_grad()`` (tensors without a version counter take the uncached routes)."""
        T = self.num_diffusion_timesteps
        lap = _Lap(timing, batch["ligand_pos"].device)      # (``timing``: see sample_many)
        st = self.begin_sampling(batch, keep_trajectory=True)
        lap("begin_sampling_s")
        use_graph = bool(use_graph) and noise_tape is None
        if use_graph:
            replay, done = self.make_step_graph(st)
            for _ in range(T - done):
                replay()
            st["x_lig"], st["c_lig"] = st["traj_x"][0], st["traj_c"][0]
        for t_idx in (reversed(range(T)) if not use_graph else ()):
            self.denoise_step(st, t_idx, noise_tape[t_idx] if noise_tape is not None else None)
        lap("steps_s")
        out_dev = torch.device("cpu") if return_device is None else torch.device(return_device)
        traj_x, traj_c, bl_out = st["traj_x"].to(out_dev), st["traj_c"].to(out_dev), st["bl"].to(out_dev)
        lap("traj_download_s")
        return {t - 1: (traj_x[t], traj_c[t], bl_out) for t in range(T + 1)}

    # hooks of BatchesInFlight.sample_many
    def _many_begin(self, batch, tape):
        return self.begin_sampling(batch, keep_trajectory=True)

    def _many_step(self, st, t_idx, tape):
        self.denoise_step(st, t_idx, tape[t_idx] if tape is not None else None)

    def _many_finish(self, st, out_dev):
        T = self.num_diffusion_timesteps
        traj_x, traj_c, bl_out = st["traj_x"].to(out_dev), st["traj_c"].to(out_dev), st["bl"].to(out_dev)
        return {t - 1: (traj_x[t], traj_c[t], bl_out) for t in range(T + 1)}

"""``diffsbdd`` model class behind the model registry (repo/models/diffusion/diffsbdd.py:23-361): same
constructor config, same state-dict keys (``pos_scheduler.gamma.gamma``, ``type_scheduler.gamma.gamma``,
``context_embedder.*``, ``denoiser.*``), same ``sample(batch) -> traj`` contract.  The denoiser call -- the
hot path -- is the same libcbgx ``UniTransformer`` as TargetDiff; the variational (gamma-schedule) sampler
around it is element-wise work on [N_lig,3] / [N_lig,C] and stays PyTorch on the device.

Sampler semantics reproduced from the reference (quirks included, see oracle/diffsbdd.py): continuous atom
types normalised by 4, the pocket re-centred on the ligand mean at every Gaussian draw (COM-free subspace), every
ligand atom updated, ``traj[0]`` overwritten by the final ``sample_p_xh_given_z0`` result."""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _native
from .registry import get_e3_gnn, register_model
from .targetdiff import NUM_AA, BatchesInFlight, PLContextEmbedder, TargetDiff, compose_embed


class PredefinedNoiseSchedule(nn.Module):
    """Frozen gamma lookup table, key ``gamma`` (repo/models/diffusion/schedule_utils.py:60-96)."""

    def __init__(self, noise_schedule, timesteps, precision):
        super().__init__()
        self.timesteps = timesteps
        if "polynomial" not in noise_schedule:
            raise ValueError(noise_schedule)
        splits = noise_schedule.split("_")
        assert len(splits) == 2
        power = float(splits[1])
        steps = timesteps + 1
        x = np.linspace(0, steps, steps)
        alphas2 = (1 - np.power(x / steps, power)) ** 2
        a2 = np.concatenate([np.ones(1), alphas2], axis=0)
        alphas2 = np.cumprod(np.clip(a2[1:] / a2[:-1], a_min=0.001, a_max=1.0), axis=0)
        alphas2 = (1 - 2 * precision) * alphas2 + precision
        gamma = -(np.log(alphas2) - np.log(1 - alphas2))
        self.gamma = nn.Parameter(torch.from_numpy(gamma).float(), requires_grad=False)

    def forward(self, t):
        return self.gamma[torch.round(t * self.timesteps).long()]


class DiffsbddVariationalScheduler(nn.Module):
    """The reference scheduler (diffusion_scheduler.py:577-1040): sampling half and the training loss."""

    def __init__(self, num_timestep, type="polynomial_2"):
        super().__init__()
        self.num_timestep = num_timestep
        if type == "learned":
            raise NotImplementedError("learned gamma network is not used by any shipped config")
        self.gamma = PredefinedNoiseSchedule(type, timesteps=num_timestep, precision=5e-4)

    @staticmethod
    def scatter_mean(src, index, n, ordered=False):
        """per-graph mean.  ``ordered`` (index non-decreasing, the collated layout): one sequential sum per graph
        (torch.segment_reduce) instead of index_add_'s float atomics, whose summation order -- and with it the last bit of the
        mean, and everything sampled after it -- changes from run to run on the GPU"""
        cnt = torch.zeros(n, dtype=torch.long, device=src.device).index_add_(0, index, torch.ones_like(index))
        if ordered:
            s = torch.segment_reduce(src, "sum", lengths=cnt, axis=0, unsafe=True)
        else:
            s = torch.zeros((n,) + src.shape[1:], dtype=src.dtype, device=src.device).index_add_(0, index, src)
        return s / cnt.clamp(min=1).to(src.dtype).view(-1, *[1] * (src.dim() - 1))

    def remove_mean_batch(self, x_lig, x_rec, bl, br, B, ordered=False):
        mean = self.scatter_mean(x_lig, bl, B, ordered)
        return x_lig - mean[bl], x_rec - mean[br]

    # ---- training side (diffusion_scheduler.py:740-960) ----
    def alpha(self, gamma):
        return torch.sqrt(torch.sigmoid(-gamma))

    def sigma(self, gamma):
        return torch.sqrt(torch.sigmoid(gamma))

    def forward_pos_center_noise(self, x_lig, x_rec, t, bl, br, B, gen_flag, noise=None):
        """q(z_t | x) for the coordinates with the pocket re-centred on the noisy ligand (:740-763, zero_center=False)"""
        if noise is None:
            noise = torch.randn_like(x_lig)
        g = self.gamma(t).view(B, 1)
        x_noisy = self.alpha(g)[bl] * x_lig + self.sigma(g)[bl] * noise
        x_noisy, x_rec = self.remove_mean_batch(x_noisy, x_rec.detach().clone(), bl, br, B)
        return torch.where(gen_flag.unsqueeze(-1), x_noisy, x_lig), noise, x_rec

    def forward_type_add_noise(self, c, t, bl, B, gen_flag, noise=None):
        if noise is None:
            noise = torch.randn_like(c)
        g = self.gamma(t).view(B, 1)
        c_noisy = self.alpha(g)[bl] * c + self.sigma(g)[bl] * noise
        return torch.where(gen_flag.unsqueeze(-1), c_noisy, c), noise

    @staticmethod
    def graph_sizes(index, n):
        """torch.bincount(index, minlength=n) for graph ids < n without its host synchronisation (bincount reads max(index))"""
        return torch.zeros(n, dtype=torch.long, device=index.device).index_add_(0, index, torch.ones_like(index))

    @staticmethod
    def sum_except_batch(x, index, n):
        return torch.zeros(n, dtype=x.dtype, device=x.device).index_add(0, index, x.sum(-1))

    def kl_prior(self, x, bl, B, dimensions):
        g_T = self.gamma(torch.ones(B, 1, device=x.device)).view(B, 1)
        mu = self.alpha(g_T)[bl] * x
        sigma_T = self.sigma(g_T).view(B)
        mu2 = self.sum_except_batch(mu ** 2, bl, B)
        d = dimensions
        return d * torch.log(1.0 / sigma_T) + 0.5 * (d * sigma_T ** 2 + mu2) - 0.5 * d

    def get_score_loss(self, pred, tgt, t, gen_flag, batch_idx, B, t_is_zero, x_lig_0=None, c_lig_0=None, c_lig_t=None):
        """training-mode loss (:886-900, 930-945): per graph  0.5 sum(err^2) [t != 0] / (n dim)  +  -log p(. | z_0) [t == 0]
        + KL prior;  mean over graphs.  Continuous reconstruction term for coordinates, discretised Gaussian for types."""
        bl = batch_idx
        n = self.graph_sizes(bl, B)
        err = self.sum_except_batch((tgt - pred) ** 2, bl, B)
        loss_t = 0.5 * err * (1.0 - t_is_zero) / (n * pred.shape[-1])
        if x_lig_0 is not None:
            loss_0 = 0.5 * err * t_is_zero
            kl = self.kl_prior(x_lig_0, bl, B, (n - 1) * 3)
        else:
            g_t = self.gamma(t).view(B, 1)
            sigma0 = self.sigma(g_t) * 4.0
            centred = c_lig_t * 4.0 - 1.0
            cdf = lambda v: 0.5 * (1.0 + torch.erf(v / math.sqrt(2.0)))
            logp = torch.log(cdf((centred + 0.5) / sigma0[bl]) - cdf((centred - 0.5) / sigma0[bl]) + 1e-10)
            logp = logp - torch.logsumexp(logp, dim=1, keepdim=True)
            loss_0 = -self.sum_except_batch(logp * (c_lig_0 * 4.0), bl, B) * t_is_zero
            kl = self.kl_prior(c_lig_0, bl, B, 1)
        info = {"eps_0": tgt, "eps_pred": pred, "mask_gen": gen_flag}
        return (loss_t + loss_0 + kl).mean(), info

    def neg_log_constants(self, n, dim):
        """-log_constants_p_x_given_z0 (:679-691): (n - 1) dim (0.5 gamma_0 + 0.5 log 2 pi) per graph"""
        g0 = self.gamma(torch.zeros(n.shape[0], 1, device=n.device)).view(-1)
        return ((n - 1) * dim) * (0.5 * g0 + 0.5 * math.log(2 * math.pi))

    def get_score_loss_eval(self, pred, tgt, s, t, gen_flag, batch_idx, B, pred0, tgt0, x_lig_0=None, c_lig_0=None,
                            c_lig_t0=None):
        """evaluation-mode loss (:902-928): per graph  -T/2 (1 - SNR(gamma_s - gamma_t)) sum(err^2)  +  KL prior  +
        -log p(. | z_0) of a second network call on the t = 0 noising (``pred0``/``tgt0``; ``c_lig_t0`` the t = 0 noised
        types)  +  the Gaussian normalisation constant (which the reference also adds to the type term, dim = C)."""
        bl = batch_idx
        n = self.graph_sizes(bl, B)
        err = self.sum_except_batch((tgt - pred) ** 2, bl, B)
        g_s, g_t = self.gamma(s).view(B), self.gamma(t).view(B)
        loss_t = -self.num_timestep * 0.5 * (1.0 - torch.exp(-(g_s - g_t))) * err
        if x_lig_0 is not None:
            kl = self.kl_prior(x_lig_0, bl, B, (n - 1) * 3)
            loss_0 = 0.5 * self.sum_except_batch((tgt0 - pred0) ** 2, bl, B)
        else:
            kl = self.kl_prior(c_lig_0, bl, B, 1)
            g_0 = self.gamma(torch.zeros_like(s)).view(B, 1)
            sigma0 = self.sigma(g_0) * 4.0
            centred = c_lig_t0 * 4.0 - 1.0
            cdf = lambda v: 0.5 * (1.0 + torch.erf(v / math.sqrt(2.0)))
            logp = torch.log(cdf((centred + 0.5) / sigma0[bl]) - cdf((centred - 0.5) / sigma0[bl]) + 1e-10)
            logp = logp - torch.logsumexp(logp, dim=1, keepdim=True)
            loss_0 = -self.sum_except_batch(logp * (c_lig_0 * 4.0), bl, B)
        loss_0 = loss_0 + self.neg_log_constants(n, tgt0.shape[-1])
        info = {"eps_0": tgt, "eps_pred": pred, "mask_gen": gen_flag}
        return (loss_t + loss_0 + kl).mean(), info

    def sample_normal_zero_com(self, mu_lig, xh0_pocket, sigma, bl, br, B, com=False, eps=None, ordered=False):
        if eps is None:
            eps = torch.randn((bl.shape[0], mu_lig.size(1)), device=mu_lig.device)
        out = mu_lig + sigma[bl] * eps
        if com:
            return self.remove_mean_batch(out, xh0_pocket, bl, br, B, ordered)
        return out

    def sample_p_zs_given_zt(self, s, t, zt_lig, xh0_pocket, bl, br, B, eps_t_lig, com=False, eps=None):
        gs, gt = self.gamma(s), self.gamma(t)
        sigma2_ts = (-torch.expm1(F.softplus(gs) - F.softplus(gt))).view(-1, 1)
        alpha_ts = torch.exp(0.5 * (F.logsigmoid(-gt) - F.logsigmoid(-gs))).view(-1, 1)
        sigma_s = torch.sqrt(torch.sigmoid(gs)).view(-1, 1)
        sigma_t = torch.sqrt(torch.sigmoid(gt)).view(-1, 1)
        mu = zt_lig / alpha_ts[bl] - (sigma2_ts / alpha_ts / sigma_t)[bl] * eps_t_lig
        sigma = torch.sqrt(sigma2_ts) * sigma_s / sigma_t
        if com:
            return self.sample_normal_zero_com(mu, xh0_pocket, sigma, bl, br, B, com=True, eps=eps)
        return self.sample_normal_zero_com(mu, xh0_pocket, sigma, bl, br, B, com=False, eps=eps), xh0_pocket


class _DiffSBDDLossFunction(torch.autograd.Function):
    """Both training losses of ``DiffSBDD.get_loss`` on the ligand rows of the denoiser outputs (diffsbdd.py:91-195;
    ``DiffsbddVariationalScheduler.get_score_loss`` x 2) in two launches, which also leave their gradients with respect to those rows
    (csrc/train_loss_diffsbdd.hip, cbgx_diffsbdd_loss); the backward scatters them, scaled by the upstream gradients, into full-size
    dL/dx_out and dL/dlogits with the kernel TargetDiff's losses use (cbgx_targetdiff_loss_backward)."""

    @staticmethod
    def forward(ctx, xo, logits, eps_x, eps_c, t_idx, sort_idx, graph_ptr, n_rec, gdata):
        dev = xo.device
        n_lig, C, B = eps_x.shape[0], logits.shape[1], t_idx.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        losses, glosses = torch.empty(2, **f32), torch.empty(2 * B, **f32)
        x_pred, c_pred = torch.empty(n_lig, 3, **f32), torch.empty(n_lig, C, **f32)
        gpos, gz = torch.empty(n_lig, 3, **f32), torch.empty(n_lig, C, **f32)
        _native.check(_native.lib().cbgx_diffsbdd_loss(
            _native.ptr(xo), _native.ptr(logits), _native.ptr(eps_x), _native.ptr(eps_c), _native.ptr(t_idx), _native.ptr(sort_idx),
            _native.ptr(graph_ptr), int(n_rec), n_lig, B, C, _native.ptr(gdata), _native.ptr(glosses), _native.ptr(losses),
            _native.ptr(x_pred), _native.ptr(c_pred), _native.ptr(gpos), _native.ptr(gz), _native.current_stream(dev)),
            "cbgx_diffsbdd_loss")
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
        return (grad_x, grad_logits) + (None,) * 7


@register_model("diffsbdd")
class DiffSBDD(BatchesInFlight, nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        gen = cfg.generator
        self.num_diffusion_timesteps = gen.num_diffusion_timesteps
        self.denoise_structure = gen.get("denoise_structure", True)
        self.denoise_atom = gen.get("denoise_atom", True)
        self.num_classes = cfg.num_atomtype
        self.pos_scheduler = DiffsbddVariationalScheduler(self.num_diffusion_timesteps, type=gen.pos_schedule.type)
        self.type_scheduler = DiffsbddVariationalScheduler(self.num_diffusion_timesteps, type=gen.atom_schedule.type)
        cfg.embedder.num_atomtype = cfg.num_atomtype
        self.context_embedder = PLContextEmbedder(cfg.embedder)
        self.denoiser = get_e3_gnn(cfg.encoder, num_classes=self.num_classes)
        self.intersect_reg = cfg.get("intersect_reg", True)
        # CBGX_FUSED_TRAINING_OPS=0: the noising and both losses as tensor operations (the path the evaluation mode always takes)
        self.fused_training_ops = os.environ.get("CBGX_FUSED_TRAINING_OPS", "1") != "0"

    # ---- training (diffsbdd.py:45-195, training mode) ----------------------------------------------------------
    def sample_time(self, batch_size, device="cuda"):
        """time_sampler 'random' (_base.py:30-33): integers 0..T as floats"""
        return torch.randint(0, self.num_diffusion_timesteps + 1, size=(batch_size,), device=device).float()

    def forward(self, batch, t=None, noise=None):
        """``loss_dict, results = model(batch)`` (diffsbdd.py:48-86).  Training mode: {'pos', 'atom'} of one random time
        per graph with the reference's training-time weighting; ``t`` [B] float in {0..T} and
        ``noise=(eps_x [N_lig,3], eps_c [N_lig,C])`` replay the draws in tests.  Evaluation mode: the variational bound
        (SNR-weighted loss_t + KL prior + the t = 0 reconstruction term from a second denoiser call) averaged over
        ``cfg.eval_interval`` (10) evenly spaced times, ``results`` a list with one entry per time; ``noise`` is then a
        list of (eps_x, eps_c, eps_x0, eps_c0) per time."""
        bl = batch["ligand_element_batch"]
        # the graph count: from the batch if the collate recorded it (no host synchronisation in the training step), else as the
        # reference computes it
        B = int(batch["num_graphs"]) if "num_graphs" in batch else (int(t.shape[0]) if t is not None else int(bl.max().item()) + 1)
        dev = batch["ligand_pos"].device
        if self.training or t is not None:
            if t is None:
                t = self.sample_time(B, device=dev)
            return self.get_loss(batch, t, noise)
        times = np.linspace(1, self.num_diffusion_timesteps, self.cfg.get("eval_interval", 10))
        tot, results = {"pos": 0.0, "atom": 0.0}, []
        for k, tv in enumerate(times):
            t_long = torch.full((B,), int(tv), dtype=torch.long, device=dev)
            ld, res = self.get_loss(batch, t_long, noise[k] if noise is not None else None, evaluate=True)
            for key in tot:
                tot[key] = tot[key] + ld[key]
            results.append(res)
        return {k: v / len(times) for k, v in tot.items()}, results

    def get_loss(self, batch, t_int, noise=None, evaluate=False):
        T, C = self.num_diffusion_timesteps, self.num_classes
        x0 = batch["ligand_pos"].float()
        bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
        lig_flag_l = batch["ligand_lig_flag"]
        gen_l = batch.get("ligand_gen_flag", lig_flag_l).bool()
        gen_r = batch.get("protein_gen_flag", torch.zeros_like(batch["protein_lig_flag"])).bool()
        B = int(t_int.shape[0])
        v_rec = batch["protein_atom_feature"].float() / 4.0
        t = t_int / T
        v0 = batch["ligand_atom_type"]
        if (self.fused_training_ops and not evaluate and x0.is_cuda and C <= 32 and v0.dtype == torch.int64 and x0.shape[0] > 0
                and batch["protein_pos"].shape[0] > 0 and self.pos_scheduler.gamma.gamma.shape[0] == T + 1):
            sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TargetDiff.compose_plan(bl, br, B)
            return self._get_loss_fused(batch, t_int, t, noise, x0, v0, v_rec, gen_l, gen_r, sort_idx, batch_idx, lig_flag, graph_ptr, B)
        c0 = F.one_hot(v0, C).float() / 4.0
        x0c, xr0 = self.pos_scheduler.remove_mean_batch(x0, batch["protein_pos"].float(), bl, br, B)
        sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TargetDiff.compose_plan(bl, br, B)

        def noise_and_denoise(tt, eps_x, eps_c):
            x_t, pos_noise, xr_t = self.pos_scheduler.forward_pos_center_noise(x0c, xr0, tt, bl, br, B, gen_l, noise=eps_x)
            c_t, type_noise = self.type_scheduler.forward_type_add_noise(c0, tt, bl, B, gen_l, noise=eps_c)
            x, h, gen_flag = compose_embed(self.context_embedder, xr_t, x_t, v_rec, batch["protein_aa_type"], c_t, sort_idx, gen_r,
                                           gen_l)
            xo, _, logits = self.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen_flag,
                                          graph_ptr=graph_ptr, ligand_outputs_only=True)
            return xo[lig_rows], logits[lig_rows], pos_noise, type_noise, c_t

        noise = noise if noise is not None else (None,) * (4 if evaluate else 2)
        x_pred, c_pred, pos_noise, type_noise, c_t = noise_and_denoise(t, noise[0], noise[1])
        if not evaluate:
            t_is_zero = (t_int == 0).float()
            loss_pos, pos_info = self.pos_scheduler.get_score_loss(x_pred, pos_noise, t, gen_l, bl, B, t_is_zero, x_lig_0=x0c)
            loss_atom, atom_info = self.type_scheduler.get_score_loss(c_pred, type_noise, t, gen_l, bl, B, t_is_zero,
                                                                      c_lig_0=c0, c_lig_t=c_t)
        else:
            s = (t_int - 1) / T
            x_pred0, c_pred0, pos_noise0, type_noise0, c_t0 = noise_and_denoise(torch.zeros_like(s), noise[2], noise[3])
            loss_pos, pos_info = self.pos_scheduler.get_score_loss_eval(x_pred, pos_noise, s, t, gen_l, bl, B, x_pred0,
                                                                        pos_noise0, x_lig_0=x0c)
            loss_atom, atom_info = self.type_scheduler.get_score_loss_eval(c_pred, type_noise, s, t, gen_l, bl, B, c_pred0,
                                                                           type_noise0, c_lig_0=c0, c_lig_t0=c_t0)
        results = {k + "_pos": v for k, v in pos_info.items()}
        results.update({k + "_atom": v for k, v in atom_info.items()})
        return {"pos": loss_pos, "atom": loss_atom}, results

    def _schedule_tables(self, dev):
        """alpha(t) / sigma(t) for t = 0 .. T with the tensor path's own operations, so that the fused noising multiplies by the same
        bits; cached (the gamma table is frozen)"""
        ps, ts = self.pos_scheduler, self.type_scheduler
        key = (torch.device(dev), ps.gamma.gamma.data_ptr(), _native.version(ps.gamma.gamma), ts.gamma.gamma.data_ptr(),
               _native.version(ts.gamma.gamma))
        tab = getattr(self, "_alpha_sigma", None)
        if tab is None or tab[0] != key:      # (a load_state_dict / .to() writes or replaces the table's storage: rebuilt)
            if not torch.equal(ps.gamma.gamma, ts.gamma.gamma):
                raise ValueError("the fused DiffSBDD losses take one schedule for coordinates and types (every shipped config); "
                                 "set CBGX_FUSED_TRAINING_OPS=0 for separate ones")
            g = ps.gamma.gamma.detach().to(dev)
            tab = (key, ps.alpha(g).contiguous(), ps.sigma(g).contiguous())
            self._alpha_sigma = tab
        return tab[1], tab[2]

    def _get_loss_fused(self, batch, t_int, t, noise, x0, v0, v_rec, gen_l, gen_r, sort_idx, batch_idx, lig_flag, graph_ptr, B):
        """training-mode ``get_loss`` with its tensor operations in three launches (csrc/train_loss_diffsbdd.hip): the noising and the
        network-independent loss terms before the denoiser, both losses and their gradients after it.  Draws are made here, in the
        order and shapes of the tensor path, so a seeded run sees the same noise on both."""
        T, C = self.num_diffusion_timesteps, self.num_classes
        dev = x0.device
        x_rec = batch["protein_pos"].float().contiguous()
        x0 = x0.contiguous()
        n_rec, n_lig = x_rec.shape[0], x0.shape[0]
        eps_x, eps_c = noise if noise is not None else (None, None)
        eps_x = torch.randn_like(x0) if eps_x is None else eps_x.float().contiguous()
        eps_c = torch.randn(n_lig, C, dtype=torch.float32, device=dev) if eps_c is None else eps_c.float().contiguous()
        t_idx = torch.round(t * T).long()                      # PredefinedNoiseSchedule.forward's index
        alpha_tab, sigma_tab = self._schedule_tables(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        x_t, xr_t, c_t = torch.empty(n_lig, 3, **f32), torch.empty(n_rec, 3, **f32), torch.empty(n_lig, C, **f32)
        gdata = torch.empty(4 * B, **f32)
        sort_idx, gen_l = sort_idx.contiguous(), gen_l.contiguous()
        _native.check(_native.lib().cbgx_diffsbdd_train_noise(
            _native.ptr(x0), _native.ptr(x_rec), _native.ptr(v0.contiguous()), _native.ptr(eps_x), _native.ptr(eps_c), _native.ptr(gen_l),
            _native.ptr(t_idx), _native.ptr(sort_idx), _native.ptr(graph_ptr), n_rec, n_lig, B, C, _native.ptr(alpha_tab),
            _native.ptr(sigma_tab), T, _native.ptr(x_t), _native.ptr(xr_t), _native.ptr(c_t), _native.ptr(gdata),
            _native.current_stream(dev)), "cbgx_diffsbdd_train_noise")
        x, h, gen_flag = compose_embed(self.context_embedder, xr_t, x_t, v_rec, batch["protein_aa_type"], c_t, sort_idx, gen_r, gen_l)
        xo, _, logits = self.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen_flag, graph_ptr=graph_ptr,
                                      ligand_outputs_only=True)
        loss_pos, loss_atom, x_pred, c_pred = _DiffSBDDLossFunction.apply(xo, logits, eps_x, eps_c, t_idx, sort_idx, graph_ptr, n_rec,
                                                                           gdata)
        results = {"eps_0_pos": eps_x, "eps_pred_pos": x_pred, "mask_gen_pos": gen_l,
                   "eps_0_atom": eps_c, "eps_pred_atom": c_pred, "mask_gen_atom": gen_l}
        return {"pos": loss_pos, "atom": loss_atom}, results

    @torch.no_grad()
    def begin_sampling(self, batch, keep_trajectory=True, noise_draws=None, static_cache=True):
        """Everything of ``sample`` (diffsbdd.py:240-319) before the loop: composition plan, protein features, the initial
        zero-COM draws.  ``noise_draws`` (tests): the randn tensors in the reference's draw order, consumed across the steps.
        ``static_cache`` (native step only): keep the composed coordinates in the pocket's own frame -- the reference translates the
        whole pocket by the ligand's new centre of mass every step (diffsbdd.py:296-304), a rigid motion that changes neither the
        pocket's neighbour lists, nor its gate values, nor its ligand-free layer-0/1 features -- and carry the accumulated
        translation per graph in ``st['frame']`` (true position = frame position - frame[graph]; the network is
        translation-equivariant), so that the denoiser's static-context cache holds for all T steps as it does for TargetDiff."""
        sch = self.pos_scheduler
        x_rec = batch["protein_pos"].float()
        dev = x_rec.device
        v_rec = batch["protein_atom_feature"].float() / 4.0
        bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
        lig_flag_l = batch["ligand_lig_flag"]
        gen_l = batch.get("ligand_gen_flag", lig_flag_l).bool()
        gen_r = batch.get("protein_gen_flag", torch.zeros_like(batch["protein_lig_flag"])).bool()
        T, C = self.num_diffusion_timesteps, self.num_classes
        B = int(bl.max().item()) + 1
        n_lig, n_rec = bl.shape[0], x_rec.shape[0]
        draws = iter(noise_draws) if noise_draws is not None else None
        nxt = (lambda: next(draws).to(dev)) if draws is not None else (lambda: None)
        aa = F.one_hot(batch["protein_aa_type"], NUM_AA).float()
        sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TargetDiff.compose_plan(bl, br, B)
        gen_flag = torch.cat([gen_r, gen_l], 0)[sort_idx]
        rec_rows = torch.nonzero(~lig_flag).flatten()
        x = torch.empty(n_rec + n_lig, 3, dtype=torch.float32, device=dev)
        h = torch.empty(n_rec + n_lig, self.context_embedder.emb_dim, dtype=torch.float32, device=dev)
        h[rec_rows] = self.context_embedder.embed_protein(v_rec, aa)          # step-invariant
        # graphs contiguous in both index vectors (the collated layout): deterministic per-graph sums, and the native step
        bl_ordered, br_ordered = torch.stack([(bl[1:] >= bl[:-1]).all(), (br[1:] >= br[:-1]).all()]).tolist()
        mu_x = sch.scatter_mean(x_rec, br, B, br_ordered)[bl]
        mu_h = torch.zeros(B, C, device=dev)[bl]
        sigma1 = torch.ones(B, 1, device=dev)
        x_lig, x_rec = sch.sample_normal_zero_com(mu_x, x_rec, sigma1, bl, br, B, com=True, eps=nxt(), ordered=bl_ordered)
        c_lig = sch.sample_normal_zero_com(mu_h, v_rec, sigma1, bl, br, B, com=False, eps=nxt())
        st = {"B": B, "N": n_rec + n_lig, "n_lig": n_lig, "x": x, "h": h, "x_lig": x_lig, "c_lig": c_lig, "x_rec": x_rec,
              "v_rec": v_rec, "bl": bl, "br": br, "batch_idx": batch_idx, "lig_flag": lig_flag, "gen_flag": gen_flag,
              "lig_rows": lig_rows, "rec_rows": rec_rows, "graph_ptr": graph_ptr, "nxt": nxt, "drawn": draws is not None,
              "traj_x": None, "traj_c": None, "static_h": None, "frame": None}
        if keep_trajectory:
            st["traj_x"] = torch.empty(T + 1, n_lig, 3, dtype=torch.float32, device=dev)
            st["traj_c"] = torch.empty(T + 1, n_lig, C, dtype=torch.float32, device=dev)
            st["traj_x"][T], st["traj_c"][T] = x_lig, c_lig
        # native step (include/cbgx.h: cbgx_diffsbdd_step): the composed x / h are then kept up to date on the device by the
        # step kernel itself (pocket translated in place, ligand rows rewritten), st["x_rec"] is only refreshed on demand
        st["native"] = dev.type == "cuda" and bl_ordered
        if st["native"]:
            st["lig_rows32"] = lig_rows.to(torch.int32).contiguous()
            st["lig8"] = lig_flag.to(torch.uint8).contiguous()
            st["lig_ptr"] = torch.cat([torch.zeros(1, dtype=torch.long, device=dev),
                                       torch.bincount(bl, minlength=B).cumsum(0)]).to(torch.int32).contiguous()
            st["coef"] = self.step_tables()
            st["x_lig"], st["c_lig"] = x_lig.contiguous(), c_lig.contiguous()
            x[rec_rows], x[lig_rows] = x_rec, x_lig
            h[lig_rows] = self.context_embedder.embed_ligand(c_lig)
            if static_cache and not bool(gen_r.any()):
                st["static_h"] = self.denoiser.static_context(x_rec, h[rec_rows], br, rec_rows, n_rec + n_lig)
                if st["static_h"] is not None:
                    st["frame"] = torch.zeros(B, 3, dtype=torch.float32, device=dev)     # sum of the removed means so far
        return st

    @staticmethod
    def pocket_positions(st):
        """the pocket atoms' TRUE positions (the reference's x_rec_0 after the steps taken so far)"""
        if st.get("x_rec") is not None:
            return st["x_rec"]
        x_rec = st["x"][st["rec_rows"]]
        return x_rec - st["frame"][st["br"]] if st.get("frame") is not None else x_rec

    def step_tables(self):
        """(1 / alpha_ts, sigma2_ts / alpha_ts / sigma_t, sigma_ts sigma_s / sigma_t) of every step s = k / T <- t = (k + 1) / T
        as Python floats, from the scheduler's own expressions (sample_p_zs_given_zt above; diffusion_scheduler.py:982-1040),
        evaluated once on the host in fp32"""
        sch, T = self.pos_scheduler, self.num_diffusion_timesteps
        g = sch.gamma.gamma.detach().float().cpu()
        k = torch.arange(T)
        gs, gt = g[torch.round(k / T * T).long()], g[torch.round((k + 1) / T * T).long()]
        sigma2_ts = -torch.expm1(F.softplus(gs) - F.softplus(gt))
        alpha_ts = torch.exp(0.5 * (F.logsigmoid(-gt) - F.logsigmoid(-gs)))
        sigma_s, sigma_t = torch.sqrt(torch.sigmoid(gs)), torch.sqrt(torch.sigmoid(gt))
        return ((1.0 / alpha_ts).tolist(), (sigma2_ts / alpha_ts / sigma_t).tolist(),
                (torch.sqrt(sigma2_ts) * sigma_s / sigma_t).tolist())

    def _denoise(self, st):
        x, h = st["x"], st["h"]
        if st.get("native"):   # x / h already hold the current state
            xo, _, logits = self.denoiser(x=x, h=h, batch_idx=st["batch_idx"], lig_flag=st["lig_flag"],
                                          gen_flag=st["gen_flag"], graph_ptr=st["graph_ptr"], need_h=False,
                                          static_h=st.get("static_h"))
            x_pred = xo[st["lig_rows"]]
            if st.get("frame") is not None:       # the composed x lives in the pocket's frame: back to true coordinates
                x_pred = x_pred - st["frame"][st["bl"]]
            return x_pred, logits[st["lig_rows"]]
        x[st["rec_rows"]] = st["x_rec"]                                       # the pocket is translated every draw
        x[st["lig_rows"]] = st["x_lig"]
        h[st["lig_rows"]] = self.context_embedder.embed_ligand(st["c_lig"])
        xo, _, logits = self.denoiser(x=x, h=h, batch_idx=st["batch_idx"], lig_flag=st["lig_flag"], gen_flag=st["gen_flag"],
                                      graph_ptr=st["graph_ptr"], need_h=False)
        return xo[st["lig_rows"]], logits[st["lig_rows"]]

    @torch.no_grad()
    def denoise_step(self, st, t_idx, noise=None):
        """One reverse step s = t_idx / T <- t = (t_idx + 1) / T (diffsbdd.py:296-304): denoiser call and the two
        sample_p_zs_given_zt draws.  ``noise`` is ignored when the state was built with ``noise_draws``."""
        sch, T, B = self.pos_scheduler, self.num_diffusion_timesteps, st["B"]
        dev = st["x"].device
        bl, br, nxt = st["bl"], st["br"], st["nxt"]
        if st.get("native"):
            return self._denoise_step_native(st, t_idx)
        s = torch.full((B,), t_idx, device=dev) / T
        t = (torch.full((B,), t_idx, device=dev) + 1) / T
        x_pred, c_out = self._denoise(st)
        if self.denoise_structure:
            st["x_lig"], st["x_rec"] = sch.sample_p_zs_given_zt(s, t, st["x_lig"], st["x_rec"], bl, br, B, x_pred, com=True,
                                                                eps=nxt())
        if self.denoise_atom:
            st["c_lig"], _ = sch.sample_p_zs_given_zt(s, t, st["c_lig"], st["v_rec"], bl, br, B, c_out, com=False, eps=nxt())
        if st["traj_x"] is not None:
            st["traj_x"][t_idx], st["traj_c"][t_idx] = st["x_lig"], st["c_lig"]
        return st

    def _denoise_step_native(self, st, t_idx):
        lib = _native.lib()
        dev = st["x"].device
        n_lig, C, B = st["n_lig"], self.num_classes, st["B"]
        x, h, x_lig, c_lig, nxt = st["x"], st["h"], st["x_lig"], st["c_lig"], st["nxt"]
        xo, _, logits = self.denoiser(x=x, h=h, batch_idx=st["batch_idx"], lig_flag=st["lig_flag"], gen_flag=st["gen_flag"],
                                      graph_ptr=st["graph_ptr"], need_h=False, static_h=st.get("static_h"))
        eps_x = eps_c = None
        if self.denoise_structure:      # the reference's draw order: positions, then types
            eps_x = nxt()
            eps_x = torch.randn(n_lig, 3, dtype=torch.float32, device=dev) if eps_x is None else eps_x.float().contiguous()
        if self.denoise_atom:
            eps_c = nxt()
            eps_c = torch.randn(n_lig, C, dtype=torch.float32, device=dev) if eps_c is None else eps_c.float().contiguous()
        if st["traj_x"] is not None:
            x_next, c_next = st["traj_x"][t_idx], st["traj_c"][t_idx]
        else:
            x_next, c_next = torch.empty_like(x_lig), torch.empty_like(c_lig)
        emb = self.context_embedder
        inv_alpha, coef, sigma = (tab[t_idx] for tab in st["coef"])
        _native.check(lib.cbgx_diffsbdd_step(
            _native.ptr(xo), _native.ptr(logits), _native.ptr(st["graph_ptr"]), _native.ptr(st["lig_rows32"]),
            _native.ptr(st["lig_ptr"]), _native.ptr(st["lig8"]), _native.ptr(x_lig), _native.ptr(c_lig), n_lig, B, C,
            inv_alpha, coef, sigma, int(self.denoise_structure), int(self.denoise_atom), _native.ptr(eps_x), _native.ptr(eps_c),
            _native.ptr(emb.ligand_atom_emb.weight), _native.ptr(emb.ligand_atom_emb.bias),
            _native.ptr(emb.ligand_indicator.weight), _native.ptr(emb.ligand_indicator.bias), _native.ptr(x_next),
            _native.ptr(c_next), _native.ptr(x), _native.ptr(h), None, _native.ptr(st.get("frame")),
            _native.current_stream(dev)), "cbgx_diffsbdd_step")
        st["x_lig"], st["c_lig"] = x_next, c_next
        st["x_rec"] = None      # lives in x[rec_rows] now
        return st

    @torch.no_grad()
    def finish_sampling(self, st):
        """sample_p_xh_given_z0 (diffsbdd.py:321-352): the final denoiser call and the x draw; returns (x_final, c_final)."""
        sch, B, C = self.pos_scheduler, st["B"], self.num_classes
        dev = st["x"].device
        bl, br, nxt = st["bl"], st["br"], st["nxt"]
        g0 = sch.gamma(torch.zeros(B, device=dev))
        sigma0 = torch.exp(0.5 * g0).unsqueeze(1)
        x_pred, c_out = self._denoise(st)
        if st["x_rec"] is None:
            st["x_rec"] = self.pocket_positions(st)
        sig_t = torch.sqrt(torch.sigmoid(g0)).view(-1, 1)
        alp_t = torch.sqrt(torch.sigmoid(-g0)).view(-1, 1)
        mu_x = 1.0 / alp_t[bl] * (st["x_lig"] - sig_t[bl] * x_pred)
        x_fin, _ = sch.sample_normal_zero_com(mu_x, st["x_rec"], sigma0, bl, br, B, com=True, eps=nxt())
        nxt() if st["drawn"] else torch.randn(st["n_lig"], C, device=dev)   # the reference draws and discards it
        return x_fin, st["c_lig"] * 4.0

    # hooks of BatchesInFlight.sample_many (the tape of a batch = its list of randn draws in the reference's order)
    @torch.no_grad()
    def _many_begin(self, batch, tape):
        return self.begin_sampling(batch, keep_trajectory=True, noise_draws=tape)

    @torch.no_grad()
    def _many_step(self, st, t_idx, tape):
        self.denoise_step(st, t_idx)

    @torch.no_grad()
    def _many_finish(self, st, out_dev):
        T = self.num_diffusion_timesteps
        x_fin, c_fin = self.finish_sampling(st)
        traj_x, traj_c, bl_out = st["traj_x"].to(out_dev), st["traj_c"].to(out_dev), st["bl"].to(out_dev)
        traj = {t - 1: (traj_x[t], traj_c[t], bl_out) for t in range(T + 1)}
        traj[0] = (x_fin.to(out_dev), c_fin.to(out_dev), bl_out)
        return traj

    @torch.no_grad()
    def sample(self, batch, noise_draws=None, return_device=None):
        """diffsbdd.py:240-319. ``noise_draws`` (tests): list of the randn tensors in the reference's draw order."""
        T = self.num_diffusion_timesteps
        st = self.begin_sampling(batch, keep_trajectory=True, noise_draws=noise_draws)
        for t_idx in reversed(range(T)):
            self.denoise_step(st, t_idx)
        x_fin, c_fin = self.finish_sampling(st)
        out_dev = torch.device("cpu") if return_device is None else torch.device(return_device)
        traj_x, traj_c, bl_out = st["traj_x"].to(out_dev), st["traj_c"].to(out_dev), st["bl"].to(out_dev)
        traj = {t - 1: (traj_x[t], traj_c[t], bl_out) for t in range(T + 1)}
        traj[0] = (x_fin.to(out_dev), c_fin.to(out_dev), bl_out)
        return traj

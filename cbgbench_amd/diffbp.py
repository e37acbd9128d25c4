"""``diffbp`` model class behind the model registry (repo/models/diffusion/diffbp.py:103-299): same constructor
config and state-dict keys (``pos_scheduler.*`` VP tables, ``context_embedder.*``, ``denoiser.*``,
``com_head.{h2xattentions.{l}.*, dist_emb.*}``; the absorbing-state type schedule has no parameters), same
``sample(batch) -> traj`` contract.

Hot path per step = the shared libcbgx denoiser + ``CoMPredictor`` (diffbp.py:30-101), which is a stack of three
H2X blocks on the kNN graph / gate of the step's input coordinates fed with the denoiser's output features:
one ``cbgx_h2x_stack_forward`` call (include/cbgx.h).  The element-wise sampler maths around it (zero-COM noise,
score-type position step, absorbing-state type step) is small PyTorch work on [N_lig, .]."""
import ctypes

import torch
import torch.nn.functional as F
from torch import nn

from . import _native
from .registry import get_e3_gnn, register_model
from .targetdiff import NUM_AA, CTNVPScheduler, PLContextEmbedder, TargetDiff
from .unitransformer import GaussianSmearing, H2XAttention, MLP, _MLP_KEYS

ABSORBING_STATE = 0   # repo/utils/molecule/constants.py:8


class MaskTypeSchedule(nn.Module):
    """Absorbing-state categorical schedule (diffusion_scheduler.py:444-496); no parameters."""

    def __init__(self, num_timestep, num_classes, absorbing_state, type="uniform"):
        super().__init__()
        self.num_timestep = num_timestep
        self.num_classes = num_classes
        self.absorbing_state = absorbing_state
        self.schedule_type = type

    def backward_remove_noise(self, c_pred, ct, t, batch_idx, gen_flag, pred_logit=True, fix_pred=True, uniform=None):
        """Unmask with probability (T - t) / T: masked, generated atoms take the predicted argmax (:475-496)."""
        if pred_logit:
            c_pred = F.softmax(c_pred, dim=-1)
        vt = ct.argmax(-1)
        tb = t[batch_idx]
        prob = ((self.num_timestep - tb) / self.num_timestep).clamp(max=1.0, min=0.0)
        if uniform is None:
            uniform = torch.rand_like(vt.float())
        change = (uniform < prob) & gen_flag
        if fix_pred:
            change = change & (vt == self.absorbing_state)
        v_next = torch.where(change, c_pred.argmax(-1), vt)
        return F.one_hot(v_next, num_classes=self.num_classes).float(), v_next


class CoMPredictor(nn.Module):
    """Parameter tree of the reference's ``CoMPredictor`` (diffbp.py:30-53); ``forward`` runs in libcbgx."""

    def __init__(self, cfg):
        super().__init__()
        g = cfg.get
        self.hidden_dim = g("node_feat_dim", 128)
        self.n_heads = g("n_heads", 16)
        self.num_r_gaussian = g("num_r_gaussian", 20)
        self.num_layers = g("num_layers_com", 3)
        if (self.hidden_dim, self.n_heads, self.num_r_gaussian, g("edge_feat_dim", 4), int(g("k", 32))) != (128, 16, 20, 4, 32):
            raise ValueError("CoMPredictor (libcbgx): only node_feat_dim=128, n_heads=16, num_r_gaussian=20, k=32")
        if g("cutoff_mode", "knn") != "knn":
            raise ValueError(f"Not supported cutoff mode: {g('cutoff_mode')}")
        if g("ew_type", "global") != "global":
            raise ValueError("CoMPredictor (libcbgx): only ew_type='global'")
        kv_in = 2 * self.hidden_dim + 4 + 4 * self.num_r_gaussian
        self.h2xattentions = nn.ModuleList([H2XAttention(self.hidden_dim, self.n_heads, kv_in)
                                            for _ in range(self.num_layers)])
        self.dist_emb = nn.Sequential(GaussianSmearing(), MLP(self.num_r_gaussian, 1, self.num_r_gaussian * 8))
        self._packed = None
        self._packed_key = None
        self._workspace = None

    def _ordered_params(self):
        sd = dict(self.named_parameters())
        names = [f"dist_emb.1.{k}" for k in _MLP_KEYS]
        for l in range(self.num_layers):
            for fn in ("xk_func", "xv_func", "xq_func"):
                names += [f"h2xattentions.{l}.{fn}.{k}" for k in _MLP_KEYS]
        return [sd[n] for n in names]

    def packed_weights(self, device):
        params = self._ordered_params()
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is None or self._packed_key != key:
            lib = _native.lib()
            packed = torch.empty(lib.cbgx_packed_h2x_stack_floats(self.num_layers), dtype=torch.float32, device=device)
            srcs = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in params]
            arr = (ctypes.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
            _native.check(lib.cbgx_pack_h2x_stack(arr, len(srcs), self.num_layers, _native.ptr(packed),
                                                  _native.current_stream(device)), "cbgx_pack_h2x_stack")
            torch.cuda.current_stream(device).synchronize()
            self._packed, self._packed_key = packed, key
        return self._packed

    def forward(self, x_lig_pred, batch_idx_lig, x_composed, h_composed, gen_flag_composed, lig_flag_composed,
                batch_idx_composed, graph_ptr=None, n_graphs=None):
        """-> (zero-COM noise prediction [N_lig,3], per-graph mean shift of the ligand [N_lig,3]) (diffbp.py:79-101)."""
        if not x_composed.is_cuda:
            raise RuntimeError("CoMPredictor.forward runs on an MI355X through libcbgx (no CPU fallback exists)")
        from .unitransformer import graph_ptr_from_batch
        from .diffsbdd import DiffsbddVariationalScheduler as _S
        dev = x_composed.device
        if graph_ptr is None:
            graph_ptr = graph_ptr_from_batch(batch_idx_composed)
        B = graph_ptr.numel() - 1
        N = x_composed.shape[0]
        noise = x_lig_pred - x_composed[lig_flag_composed]
        noise = noise - _S.scatter_mean(noise, batch_idx_lig, B)[batch_idx_lig]
        lib = _native.lib()
        need = lib.cbgx_workspace_bytes(N, B)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        x_in = x_composed.detach().float().contiguous()
        h_in = h_composed.detach().float().contiguous()
        x_out = torch.empty_like(x_in)
        rc = lib.cbgx_h2x_stack_forward(
            _native.ptr(self.packed_weights(dev)), self.num_layers, _native.ptr(x_in), _native.ptr(h_in),
            _native.ptr(graph_ptr), _native.ptr(lig_flag_composed.to(torch.uint8).contiguous()),
            _native.ptr(gen_flag_composed.to(torch.uint8).contiguous()), N, B, _native.ptr(x_out),
            _native.ptr(self._workspace), self._workspace.numel(), _native.current_stream(dev))
        _native.check(rc, "cbgx_h2x_stack_forward")
        delta = (x_out - x_in)[lig_flag_composed]
        shift = _S.scatter_mean(delta, batch_idx_lig, B)[batch_idx_lig]
        return noise, shift


@register_model("diffbp")
class DiffBP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        gen = cfg.generator
        self.num_diffusion_timesteps = gen.num_diffusion_timesteps
        self.denoise_structure = gen.get("denoise_structure", True)
        self.denoise_atom = gen.get("denoise_atom", True)
        self.num_classes = cfg.num_atomtype
        ps = gen.pos_schedule
        self.pos_scheduler = CTNVPScheduler(self.num_diffusion_timesteps, beta_start=ps.beta_start,
                                            beta_end=ps.beta_end, type=ps.type)
        self.type_scheduler = MaskTypeSchedule(self.num_diffusion_timesteps, num_classes=self.num_classes,
                                               type=gen.atom_schedule.type, absorbing_state=ABSORBING_STATE)
        cfg.embedder.num_atomtype = cfg.num_atomtype
        self.context_embedder = PLContextEmbedder(cfg.embedder)
        self.denoiser = get_e3_gnn(cfg.encoder, num_classes=self.num_classes)
        self.com_head = CoMPredictor(cfg.encoder)
        self.intersect_reg = cfg.get("intersect_reg", True)

    def forward(self, batch):
        raise NotImplementedError("training loss (diffbp.py:131-234) needs the backward kernels (DESIGN.md section 8)")

    @torch.no_grad()
    def sample(self, batch, noise_tape=None, return_device=None):
        """diffbp.py:240-299. ``noise_tape``: dict t -> (eps [N_lig,3], u [N_lig]) replacing randn_like / rand_like."""
        x_lig = batch["ligand_pos"].float()
        dev = x_lig.device
        x_rec = batch["protein_pos"].float()
        v_rec = batch["protein_atom_feature"].float()
        lig_flag_l = batch["ligand_lig_flag"]
        gen_l = batch.get("ligand_gen_flag", lig_flag_l).bool()
        gen_r = batch.get("protein_gen_flag", torch.zeros_like(batch["protein_lig_flag"])).bool()
        bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
        T, C = self.num_diffusion_timesteps, self.num_classes
        B = int(bl.max().item()) + 1
        n_lig, n_rec = x_lig.shape[0], x_rec.shape[0]
        aa = F.one_hot(batch["protein_aa_type"], NUM_AA).float()
        c_lig = F.one_hot(batch["ligand_atom_type"], C).float()
        sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TargetDiff.compose_plan(bl, br, B)
        gen_flag = torch.cat([gen_r, gen_l], 0)[sort_idx]
        rec_rows = torch.nonzero(~lig_flag).flatten()
        x = torch.empty(n_rec + n_lig, 3, dtype=torch.float32, device=dev)
        h = torch.empty(n_rec + n_lig, self.context_embedder.emb_dim, dtype=torch.float32, device=dev)
        x[rec_rows] = x_rec
        h[rec_rows] = self.context_embedder.embed_protein(v_rec, aa)
        traj_x = torch.empty(T + 1, n_lig, 3, dtype=torch.float32, device=dev)
        traj_c = torch.empty(T + 1, n_lig, C, dtype=torch.float32, device=dev)
        traj_x[T], traj_c[T] = x_lig, c_lig
        for t_idx in reversed(range(T)):
            t = torch.full((B,), t_idx, dtype=torch.long, device=dev)
            x[lig_rows] = x_lig
            h[lig_rows] = self.context_embedder.embed_ligand(c_lig)
            xo, ho, logits = self.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen_flag,
                                           graph_ptr=graph_ptr)
            eps_t, eps_com = self.com_head(xo[lig_rows], bl, x, ho, gen_flag, lig_flag, batch_idx, graph_ptr=graph_ptr)
            eps, u = noise_tape[t_idx] if noise_tape is not None else (None, None)
            if self.denoise_structure:
                x_lig = self.pos_scheduler.backward_remove_noise(eps_t + eps_com, x_lig, t, bl, gen_l, type="score",
                                                                 noise=eps)
            if self.denoise_atom:
                c_lig, _ = self.type_scheduler.backward_remove_noise(logits[lig_rows], c_lig, t, bl, gen_l,
                                                                     pred_logit=True, uniform=u)
            traj_x[t_idx], traj_c[t_idx] = x_lig, c_lig
        out_dev = torch.device("cpu") if return_device is None else torch.device(return_device)
        traj_x, traj_c, bl_out = traj_x.to(out_dev), traj_c.to(out_dev), bl.to(out_dev)
        return {t - 1: (traj_x[t], traj_c[t], bl_out) for t in range(T + 1)}

"""Training step around the model classes: the body of ``train(it)`` in the reference's train.py:173-205 with the
helpers of repo/utils/train.py (get_optimizer :8-17, get_scheduler :20-44, sum_weighted_losses :121-133), made
data-parallel the MI355X way: one process per GPU, every rank steps on its own batch, and the gradients -- which
libcbgx's backward writes and autograd accumulates into ONE flat fp32 buffer (2 671 774 elements = 10.7 MB for
the shipped TargetDiff) -- are summed with a single RCCL all-reduce per optimiser step and averaged before clipping,
so ``clip_grad_norm_`` sees the global gradient.  There is no bucketing to tune: the whole model is
one bucket, far below the size where xGMI ring latency stops mattering.
"""
import time

import torch
import torch.distributed as dist
from torch.nn.utils import clip_grad_norm_


# config `type` -> constructor; the keys each one reads are the ones the reference's YAMLs carry
# (configs/*/train/*.yml `train.optimizer` / `train.scheduler`; repo/utils/train.py:8-44)
_OPTIMIZERS = {
    # fused=True on device parameters: one multi-tensor kernel chain instead of a Python loop over 342 tensors (same update rule)
    "adam": lambda c, params: _adam(params, lr=c.lr, weight_decay=c.weight_decay, betas=(c.beta1, c.beta2)),
}
_SCHEDULERS = {
    "plateau": lambda c, opt: torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=c.factor, patience=c.patience,
                                                                         min_lr=c.min_lr),
    "multistep": lambda c, opt: torch.optim.lr_scheduler.MultiStepLR(opt, milestones=c.milestones, gamma=c.gamma),
    "exp": lambda c, opt: torch.optim.lr_scheduler.ExponentialLR(opt, gamma=c.gamma),
}


class FlatAdam(torch.optim.Adam):
    """torch.optim.Adam (default flags: no amsgrad, no maximize) whose parameters, first and second moments are views of three flat
    fp32 buffers, so that ``step()`` is a handful of kernels over 2.7 M elements instead of a Python loop over 342 tensors (1.5 ms of
    host time per step, the device idle meanwhile).  Same update rule, same op order as torch's single-tensor path; ``param_groups``,
    ``state`` and ``state_dict()`` have the stock layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so checkpoints
    are interchangeable with ``torch.optim.Adam`` in both directions -- ``load_state_dict`` copies into the flat buffers.

    Two things differ from the stock class and are part of the contract:
    * ONE step counter is shared by all parameters (stock Adam keeps one per parameter; they only differ when some parameters
      had no gradient in some steps).  A parameter whose ``.grad`` is None in a step is skipped exactly like stock Adam does
      (weight and moments untouched); a loaded checkpoint must carry the same ``step`` for every parameter it has state for.
    * every ``step()`` marks the parameters as modified in place (their autograd version counters are bumped): they are views
      of the flat buffer the update writes, and without the bump neither autograd's saved-tensor checks nor
      ``UniTransformer.packed_weights`` (which re-packs when a version changes) would see the update."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(list(params), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        assert len(self.param_groups) == 1, "one parameter group (what the reference's configs build)"
        self._ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        sizes = [p.numel() for p in self._ps]
        dev = self._ps[0].device
        self._sizes = sizes
        self._w = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self._m = torch.zeros_like(self._w)
        self._v = torch.zeros_like(self._w)
        self._step = torch.zeros((), dtype=torch.float32)       # one counter, referenced by every parameter's state
        with torch.no_grad():
            for p, w in zip(self._ps, self._w.split(sizes)):
                w.copy_(p.detach().reshape(-1))
                p.data = w.view_as(p)                            # the module's tensors now alias the flat buffer
        self._bind_state()

    def _bind_state(self):
        for p, m, v in zip(self._ps, self._m.split(self._sizes), self._v.split(self._sizes)):
            self.state[p] = {"step": self._step, "exp_avg": m.view_as(p), "exp_avg_sq": v.view_as(p)}

    def state_dict(self):
        """stock layout with independent tensors: the shared step counter and the views of the flat moment buffers are cloned, so
        that a consumer (torch.optim.Adam.load_state_dict, torch.save) sees what torch.optim.Adam itself would have produced"""
        sd = super().state_dict()
        sd["state"] = {k: {n: (t.clone() if torch.is_tensor(t) else t) for n, t in st.items()} for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)                      # fills self.state[p] with fresh tensors (stock behaviour)
        steps = set()
        with torch.no_grad():
            for p, m, v in zip(self._ps, self._m.split(self._sizes), self._v.split(self._sizes)):
                st = self.state.get(p, None)
                if st and "exp_avg" in st:
                    m.copy_(st["exp_avg"].reshape(-1))
                    v.copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(float(st["step"]))
                else:                                            # no state for this parameter in the checkpoint: as after construction
                    m.zero_()
                    v.zero_()
        if len(steps) > 1:
            raise ValueError(f"FlatAdam keeps one step counter for all parameters; the checkpoint has {sorted(steps)}")
        self._step = torch.as_tensor(steps.pop() if steps else 0.0, dtype=torch.float32)
        self._bind_state()

    def _mark_modified(self):
        """bump the version counter of every parameter (they alias ``self._w``, which ``step`` has just written)"""
        inc = getattr(torch._C, "_increment_version", None)
        try:
            if inc is None:
                raise TypeError
            inc(self._ps)
        except TypeError:                                        # older / newer torch without the list form: one fused no-op kernel
            torch._foreach_mul_(self._ps, 1.0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None if closure is None else closure()
        g0 = self._ps[0].grad
        # the gradients are views of one flat buffer in parameter order (FlatGradients); recover it, or gather
        base = g0._base if g0 is not None and g0._base is not None else None
        skipped = []     # (w, m, v) slices of parameters without a gradient, restored after the flat update: stock Adam skips them
        if base is not None and base.numel() == self._w.numel() and all(
                p.grad is not None and p.grad._base is base for p in self._ps):
            g = base
        else:
            if all(p.grad is None for p in self._ps):
                return loss
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self._ps])
            for p, w, m, v in zip(self._ps, self._w.split(self._sizes), self._m.split(self._sizes), self._v.split(self._sizes)):
                if p.grad is None:
                    skipped.append(((w, w.clone()), (m, m.clone()), (v, v.clone())))
        grp = self.param_groups[0]
        beta1, beta2 = grp["betas"]
        self._step += 1
        step = float(self._step)
        if grp["weight_decay"] != 0:
            g = g.add(self._w, alpha=grp["weight_decay"])
        self._m.lerp_(g, 1 - beta1)
        self._v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        denom = (self._v.sqrt() / (bc2 ** 0.5)).add_(grp["eps"])
        self._w.addcdiv_(self._m, denom, value=-grp["lr"] / bc1)
        for pairs in skipped:
            for dst, saved in pairs:
                dst.copy_(saved)
        self._mark_modified()
        return loss


def _adam(params, **kw):
    """device parameters: FlatAdam (same update, same checkpoint format, one flat update); host parameters: the stock class"""
    params = list(params)
    if params and all(p.is_cuda for p in params):
        return FlatAdam(params, **kw)
    return torch.optim.Adam(params, **kw)


def _dispatch(table, what, cfg, arg):
    try:
        make = table[cfg.type]
    except KeyError:
        raise NotImplementedError(f"{what} not supported: {cfg.type}") from None
    return make(cfg, arg)


def get_optimizer(cfg, model):
    return _dispatch(_OPTIMIZERS, "Optimizer", cfg, model.parameters())


def get_scheduler(cfg, optimizer):
    """None when the config has no scheduler block / type (the reference returns None there too)"""
    if cfg is None or cfg.get("type", None) is None:
        return None
    return _dispatch(_SCHEDULERS, "Scheduler", cfg, optimizer)


def sum_weighted_losses(losses, weights):
    """weighted sum of a loss dict; ``weights=None`` means all ones (repo/utils/train.py:121-133)"""
    return sum(v if weights is None else weights[k] * v for k, v in losses.items())


class FlatGradients:
    """All trainable gradients as views of one contiguous fp32 buffer (parameter order), so that data-parallel
    training is one all-reduce.  ``p.grad`` is pointed at the views once; autograd then accumulates in place."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.views = [v.view_as(p) for p, v in zip(self.params, self.flat.split(sizes))]
        for p, v in zip(self.params, self.views):
            p.grad = v
        # libcbgx-backed encoders write their gradients straight into these views (one backward per step, zeroed by
        # zero() before it) instead of returning tensors for autograd to accumulate
        self._direct_modules = [mod for mod in model.modules() if hasattr(mod, "_direct_grads")]
        for mod in self._direct_modules:
            mod._direct_grads, mod._direct_written = True, False

    def zero(self):
        self.flat.zero_()
        for mod in self._direct_modules:
            mod._direct_written = False     # the next backward through it may overwrite (the buffer is zero)
        for p, v in zip(self.params, self.views):
            if p.grad is not v:     # optimizer.zero_grad(set_to_none=True) undoes the views (identity check: no per-tensor device call)
                p.grad = v

    def clip_norm_(self, max_norm, eps=1e-6):
        """torch.nn.utils.clip_grad_norm_(params, max_norm) on the flat buffer: the 2-norm of the per-tensor 2-norms is the 2-norm
        of the concatenation, so one reduction and one scaling replace 342 of each.  Returns the total norm (before clipping)."""
        total = torch.linalg.vector_norm(self.flat, 2)
        self.flat.mul_(torch.clamp(max_norm / (total + eps), max=1.0))
        return total

    def all_reduce_mean(self):
        """sum over ranks / world size; returns the wall time of the collective in seconds (0 when not distributed)."""
        if not (dist.is_available() and dist.is_initialized()):     # a one-rank group (CBGX_DIST_FORCE) still reduces: tests
            return 0.0
        sync = self.flat.is_cuda
        if sync:
            torch.cuda.synchronize(self.flat.device)
        t0 = time.perf_counter()
        if dist.get_backend() == "gloo" and self.flat.is_cuda:
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        if sync:
            torch.cuda.synchronize(self.flat.device)
        return time.perf_counter() - t0


def broadcast_parameters(model, src=0):
    """every rank starts from rank ``src``'s weights (what DDP does at construction)."""
    if dist.is_available() and dist.is_initialized():
        for t in list(model.parameters()) + list(model.buffers()):
            if dist.get_backend() == "gloo" and t.is_cuda:
                host = t.detach().cpu()
                dist.broadcast(host, src)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src)


def train_step(model, batch, optimizer, flat_grads, loss_weights=None, max_grad_norm=8.0, **forward_kwargs):
    """One iteration of train.py:173-190.  Returns (loss, loss_dict, grad_norm, allreduce_seconds)."""
    if not model.training:
        model.train()       # (walks the module tree: 0.7 ms when repeated every step)
    flat_grads.zero()
    loss_dict, _ = model(batch, **forward_kwargs)
    loss = sum_weighted_losses(loss_dict, loss_weights)
    loss.backward()
    t_ar = flat_grads.all_reduce_mean()
    # (every p.grad is a view of the flat buffer unless something replaced it: then fall back to the per-tensor form)
    if all(p.grad is v for p, v in zip(flat_grads.params, flat_grads.views)):
        grad_norm = flat_grads.clip_norm_(max_grad_norm)
    else:
        grad_norm = clip_grad_norm_(flat_grads.params, max_grad_norm)
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in loss_dict.items()}, grad_norm, t_ar


@torch.no_grad()
def validate(model, batches, loss_weights=None, evaluator=None):
    """train.py:207-246: mean weighted loss over the batches at the model's evenly spaced evaluation times (each batch weighted
    by its graph count, like the reference's ScalarMetricAccumulator, train.py:222-238), all-reduced so every rank sees the same
    value (ReduceLROnPlateau input).  ``evaluator`` (``evaluate.Evaluator``, the config's ``eval.metrics``): its metrics of every
    batch's ``results`` are averaged and all-reduced the same way; the call then returns ``(avg_loss, {metric: value})``."""
    model.eval()
    names = sorted(evaluator.evaluators) if evaluator is not None and len(evaluator) else []
    acc = [0.0] * (2 + 2 * len(names))      # [sum loss * B, sum B, (sum metric_k * B, sum B over the batches where it is defined) ...]
    with torch.no_grad():
        for batch in batches:
            loss_dict, results = model(batch)
            # weighted by the number of graphs, like the reference's ScalarMetricAccumulator (train.py:222-232)
            bl = batch.get("ligand_element_batch", None) if isinstance(batch, dict) else None
            B = batch.get("num_graphs") if isinstance(batch, dict) and batch.get("num_graphs") else (
                int(bl.max().item()) + 1 if bl is not None and bl.numel() else 1)
            acc[0] += float(sum_weighted_losses(loss_dict, loss_weights)) * B
            acc[1] += B
            if names:
                md = evaluator(results)
                for k, nm in enumerate(names):
                    v = float(md[nm])
                    if v == v:      # an undefined metric (NaN: a batch without a masked atom) is left out of ITS average -- value
                        acc[2 + 2 * k] += v * B         # and weight -- instead of entering as 0 with full weight
                        acc[3 + 2 * k] += B
    val = torch.tensor(acc, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() != "gloo":
            val = val.to(next(model.parameters()).device)
        dist.all_reduce(val, op=dist.ReduceOp.SUM)
    val = val.cpu()
    n = max(float(val[1]), 1.0)
    avg = float(val[0] / n)
    if evaluator is None:
        return avg
    # a metric that was defined on no batch at all is NaN (the reference's accumulator would propagate the NaN too)
    return avg, {nm: (float(val[2 + 2 * k] / val[3 + 2 * k]) if float(val[3 + 2 * k]) > 0 else float("nan"))
                 for k, nm in enumerate(names)}

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
    "adam": lambda c, params: torch.optim.Adam(params, lr=c.lr, weight_decay=c.weight_decay, betas=(c.beta1, c.beta2)),
}
_SCHEDULERS = {
    "plateau": lambda c, opt: torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=c.factor, patience=c.patience,
                                                                         min_lr=c.min_lr),
    "multistep": lambda c, opt: torch.optim.lr_scheduler.MultiStepLR(opt, milestones=c.milestones, gamma=c.gamma),
    "exp": lambda c, opt: torch.optim.lr_scheduler.ExponentialLR(opt, gamma=c.gamma),
}


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
        for p, v in zip(self.params, self.flat.split(sizes)):
            p.grad = v.view_as(p)
        # libcbgx-backed encoders write their gradients straight into these views (one backward per step, zeroed by
        # zero() before it) instead of returning tensors for autograd to accumulate
        self._direct_modules = [mod for mod in model.modules() if hasattr(mod, "_direct_grads")]
        for mod in self._direct_modules:
            mod._direct_grads, mod._direct_written = True, False

    def zero(self):
        self.flat.zero_()
        for mod in self._direct_modules:
            mod._direct_written = False     # the next backward through it may overwrite (the buffer is zero)
        for p, v in zip(self.params, self.flat.split([p.numel() for p in self.params])):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():   # optimizer.zero_grad(set_to_none=True) undoes the views
                p.grad = v.view_as(p)

    def all_reduce_mean(self):
        """sum over ranks / world size; returns the wall time of the collective in seconds (0 when not distributed)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            if dist.get_backend() == "gloo" and t.is_cuda:
                host = t.detach().cpu()
                dist.broadcast(host, src)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src)


def train_step(model, batch, optimizer, flat_grads, loss_weights=None, max_grad_norm=8.0, **forward_kwargs):
    """One iteration of train.py:173-190.  Returns (loss, loss_dict, grad_norm, allreduce_seconds)."""
    model.train()
    flat_grads.zero()
    loss_dict, _ = model(batch, **forward_kwargs)
    loss = sum_weighted_losses(loss_dict, loss_weights)
    loss.backward()
    t_ar = flat_grads.all_reduce_mean()
    grad_norm = clip_grad_norm_(flat_grads.params, max_grad_norm)
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in loss_dict.items()}, grad_norm, t_ar


@torch.no_grad()
def validate(model, batches, loss_weights=None):
    """train.py:207-246 without the RDKit/auroc evaluator: mean weighted loss over the batches at the model's
    evenly spaced evaluation times (each batch weighted by its graph count), all-reduced so every rank sees the same value
    (ReduceLROnPlateau input)."""
    model.eval()
    tot, n = 0.0, 0
    for batch in batches:
        loss_dict, _ = model(batch)
        # weighted by the number of graphs, like the reference's ScalarMetricAccumulator (train.py:222-232)
        bl = batch.get("ligand_element_batch", None) if isinstance(batch, dict) else None
        B = int(bl.max().item()) + 1 if bl is not None and bl.numel() else 1
        tot += float(sum_weighted_losses(loss_dict, loss_weights)) * B
        n += B
    val = torch.tensor([tot, float(n)], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() != "gloo":
            val = val.to(next(model.parameters()).device)
        dist.all_reduce(val, op=dist.ReduceOp.SUM)
    return float(val[0] / max(float(val[1]), 1.0))

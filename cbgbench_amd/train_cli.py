"""Training driver: the role of the reference's ``train.py`` (config -> model / optimizer / scheduler -> the iteration loop
with periodic validation -- loss and the config's ``eval.metrics`` (the type-prediction AUROC, ``cbgbench_amd/evaluate.py``) -- and
best-so-far checkpoints, ``train.py:99-273``), minus the LMDB datasets and tensorboard (out of scope), made
data-parallel the way SURVEY.md 8e/8f-4 asks:

* one process per GPU (``python -m cbgbench_amd.launch --nproc N -m cbgbench_amd.train_cli ...``: FileStore rendezvous, no port to
  pass; ``torchrun`` works too), every rank holds a full replica;
* a rank-aware loader: one permutation of the training complexes per epoch, seeded identically everywhere, of which rank r
  takes the entries r, r+W, ... -- no sampler object, no collective;
* gradients of all ranks summed by ONE RCCL all-reduce of the flat fp32 gradient buffer per step (``train.FlatGradients``),
  averaged before clipping so that ``clip_grad_norm_`` sees the global gradient;
* validation loss and metrics all-reduced so that ``ReduceLROnPlateau`` takes the same decision on every rank; checkpoints written by
  rank 0 only, in the reference's format ``{'config', 'model', 'optimizer', 'scheduler', 'iteration', 'avg_val_loss'}``
  under ``{logdir}/{tag}/checkpoints/{it}.pt`` (``train.py:266-273``); ``--resume`` restores all of it on every rank
  (``train.py:160-175``; ``--finetune`` keeps only the weights).

    python -m cbgbench_amd.train_cli --config configs/denovo/train/targetdiff.yml --logdir logs
                                     [--data complexes.pt | --synthetic 256] [--resume ckpt.pt] [--max_iters N]

Complex input: a ``torch.save``d list of dicts with ``protein_pos [n,3]``, ``protein_atom_feature [n,7]``,
``protein_aa_type [n]``, ``ligand_pos [m,3]``, ``ligand_atom_type [m]`` and optionally ``ligand_gen_flag [m]`` (what the
reference's featurizers produce); they are centred on the protein mean (``center_pos``, translation.py:5-25) and collated into
the ``MergeKeys`` + ``follow_batch`` schema.  Without ``--data``, synthetic complexes stand in.
"""
import argparse
import os
import time

import numpy as np
import torch

from . import get_model, load_config, set_num_atom_type, sharding, synthetic
from .config import checkpoint_config, load_checkpoint_file
from .evaluate import Evaluator
from .train import FlatGradients, broadcast_parameters, get_optimizer, get_scheduler, train_step, validate


# ---- data ------------------------------------------------------------------------------------------------------
class ComplexSet:
    """Protein-ligand complexes packed once (CSR on the host); ``collate(ids)`` builds a batch dict with index arithmetic."""

    def __init__(self, complexes, center=True):
        t = lambda a, dt: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dt)
        self.n = len(complexes)
        rec_n = [int(np.asarray(c["protein_pos"]).shape[0]) for c in complexes]
        lig_n = [int(np.asarray(c["ligand_pos"]).shape[0]) for c in complexes]
        self.rec_ptr = torch.tensor([0] + list(np.cumsum(rec_n)), dtype=torch.long)
        self.lig_ptr = torch.tensor([0] + list(np.cumsum(lig_n)), dtype=torch.long)
        ppos = [t(c["protein_pos"], torch.float32) for c in complexes]
        lpos = [t(c["ligand_pos"], torch.float32) for c in complexes]
        ctr = [p.mean(0) if center else torch.zeros(3) for p in ppos]
        self.protein_pos = torch.cat([p - m for p, m in zip(ppos, ctr)])
        self.ligand_pos = torch.cat([p - m for p, m in zip(lpos, ctr)])
        self.protein_atom_feature = torch.cat([t(c["protein_atom_feature"], torch.float32) for c in complexes])
        self.protein_aa_type = torch.cat([t(c["protein_aa_type"], torch.long) for c in complexes])
        self.ligand_atom_type = torch.cat([t(c["ligand_atom_type"], torch.long) for c in complexes])
        self.has_gen = all("ligand_gen_flag" in c for c in complexes)
        if self.has_gen:
            self.ligand_gen_flag = torch.cat([t(c["ligand_gen_flag"], torch.bool) for c in complexes])

    def __len__(self):
        return self.n

    @staticmethod
    def _rows(ptr, ids):
        cnt = ptr[ids + 1] - ptr[ids]
        seg = torch.repeat_interleave(torch.arange(ids.numel()), cnt)
        start = torch.cumsum(cnt, 0) - cnt
        return ptr[ids][seg] + (torch.arange(int(cnt.sum())) - start[seg]), seg

    def collate(self, ids, device="cpu"):
        ids = torch.as_tensor(ids, dtype=torch.long)
        rr, rseg = self._rows(self.rec_ptr, ids)
        lr, lseg = self._rows(self.lig_ptr, ids)
        b = {
            "protein_pos": self.protein_pos[rr], "protein_atom_feature": self.protein_atom_feature[rr],
            "protein_aa_type": self.protein_aa_type[rr], "protein_lig_flag": torch.zeros(rr.numel(), dtype=torch.bool),
            "protein_element_batch": rseg,
            "ligand_pos": self.ligand_pos[lr], "ligand_atom_type": self.ligand_atom_type[lr],
            "ligand_lig_flag": torch.ones(lr.numel(), dtype=torch.bool), "ligand_element_batch": lseg,
        }
        if self.has_gen:
            b["ligand_gen_flag"] = self.ligand_gen_flag[lr]
        b = {k: v.to(device) for k, v in b.items()}
        b["num_graphs"] = int(ids.numel())       # known on the host: spares the model a device round trip per step
        b["max_ligand_atoms"] = int((self.lig_ptr[ids + 1] - self.lig_ptr[ids]).max())     # DiffBP's interior loss sizes a tile with it
        return b


def synthetic_complexes(n, seed, num_classes, n_rec_range=(350, 650), n_lig_range=(10, 45)):
    """stand-in data: synthetic pockets with a ligand blob near the centre (jittered lattice points)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        pos, feat, aa = synthetic.make_pocket(rng, int(rng.integers(n_rec_range[0], n_rec_range[1] + 1)))
        m = int(rng.integers(n_lig_range[0], n_lig_range[1] + 1))
        lig = (rng.standard_normal((m, 3)) * 2.0).astype(np.float32)
        out.append({"protein_pos": pos, "protein_atom_feature": feat, "protein_aa_type": aa, "ligand_pos": lig,
                    "ligand_atom_type": rng.integers(0, num_classes, size=m).astype(np.int64)})
    return out


class ShardedLoader:
    """Rank-aware batches of item indices.  Every rank draws the same permutation of ``range(n_items)`` per epoch
    (``seed + epoch``) and keeps the entries ``rank, rank + world, ...``; the tail is padded by wrapping around so that all
    ranks make the same number of steps (a collective per step must not be left waiting)."""

    def __init__(self, n_items, batch_size, rank=0, world=1, seed=0, shuffle=True):
        if n_items < 1:
            raise ValueError("ShardedLoader: empty dataset")
        self.n, self.bs, self.rank, self.world, self.seed, self.shuffle = n_items, batch_size, rank, world, seed, shuffle
        self.per_rank = (n_items + world - 1) // world

    def epoch(self, e):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + e)
            perm = torch.randperm(self.n, generator=g)
        else:
            perm = torch.arange(self.n)
        total = self.per_rank * self.world
        if total > self.n:                                   # wrap around (more than once when n_items < world / 2)
            perm = perm.repeat((total + self.n - 1) // self.n)[:total]
        mine = perm[self.rank::self.world]
        return [mine[i:i + self.bs].tolist() for i in range(0, mine.numel(), self.bs)]

    def __iter__(self):           # the reference's inf_iterator (repo/utils/train.py)
        e = 0
        while True:
            for ids in self.epoch(e):
                yield ids
            e += 1


# ---- checkpoints -----------------------------------------------------------------------------------------------
def save_checkpoint(path, config, model, optimizer, scheduler, iteration, avg_val_loss):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = path + ".tmp"
    # same keys as train.py:266-273; `config` in a form the reference's scripts can unpickle without this package
    torch.save({"config": checkpoint_config(config), "model": model.state_dict(), "optimizer": optimizer.state_dict(),
                "scheduler": scheduler.state_dict() if scheduler is not None else {},
                "iteration": iteration, "avg_val_loss": avg_val_loss}, tmp)
    os.replace(tmp, path)         # never leave a truncated checkpoint behind


def load_checkpoint(path, model, optimizer=None, scheduler=None, finetune=False, device="cpu"):
    """``train.py:160-175``: weights with strict=False; optimizer / scheduler / iteration unless fine-tuning.
    Returns (first iteration, missing keys, unexpected keys)."""
    ckpt = load_checkpoint_file(path, map_location=device)
    res = model.load_state_dict(ckpt["model"], strict=False)
    it_first = 1
    if not finetune:
        if optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
        if scheduler is not None and ckpt.get("scheduler"):
            scheduler.load_state_dict(ckpt["scheduler"])
        it_first = int(ckpt["iteration"])          # the reference resumes AT the saved iteration (no + 1)
    return it_first, list(res.missing_keys), list(res.unexpected_keys)


# ---- the loop --------------------------------------------------------------------------------------------------
def run(config, config_name, train_set, val_set, dev, logdir, tag="", resume=None, finetune=False, max_iters=None,
        log=print):
    rank, world, _ = sharding.env_rank_world()
    tc, ec = config.train, config.get("eval", {})
    max_iters = int(max_iters if max_iters is not None else tc.max_iters)
    val_freq = int(ec.get("val_freq", 1000))
    report_freq = int(tc.get("report_freq", 100))
    weights = tc.get("loss_weights", None)
    torch.manual_seed(int(tc.get("seed", 2022)) + rank)          # per-rank noise / time draws
    model = get_model(config.model).to(dev)
    optimizer = get_optimizer(tc.optimizer, model)
    scheduler = get_scheduler(tc.get("scheduler", None), optimizer)
    evaluator = Evaluator(ec.get("metrics", []))                  # train.py:141
    it_first = 1
    if resume:
        it_first, missing, unexpected = load_checkpoint(resume, model, optimizer, scheduler, finetune, device=dev)
        if rank == 0:
            log(f"[resume] {resume}: iteration {it_first}, missing keys {len(missing)}, unexpected {len(unexpected)}")
    else:
        broadcast_parameters(model)                              # every replica starts from rank 0's initialisation
    flat = FlatGradients(model)
    train_it = iter(ShardedLoader(len(train_set), int(tc.batch_size), rank, world, seed=int(tc.get("seed", 2022))))
    val_batches = ShardedLoader(len(val_set), int(tc.batch_size), rank, world, shuffle=False).epoch(0)
    ckpt_dir = os.path.join(logdir, tag or config_name, "checkpoints")
    best_loss, best_iter, history, metric_history = None, None, [], []
    t_last = time.perf_counter()
    for it in range(it_first, max_iters + 1):
        batch = train_set.collate(next(train_it), dev)
        loss, loss_dict, grad_norm, t_ar = train_step(model, batch, optimizer, flat, weights,
                                                      max_grad_norm=float(tc.get("max_grad_norm", 8.0)))
        if it % report_freq == 0 and rank == 0:
            now = time.perf_counter()
            parts = " | ".join(f"loss({k}) {float(v):.4f}" for k, v in loss_dict.items())
            log(f"[train] iter {it:05d} | loss {float(loss):.4f} | {parts} | grad {float(grad_norm):.4f} | "
                f"lr {optimizer.param_groups[0]['lr']:.3e} | {1e3 * (now - t_last) / report_freq:.1f} ms/iter "
                f"(all-reduce {1e3 * t_ar:.2f} ms)")
            t_last = now
        if it % val_freq == 0:
            avg, metrics = validate(model, (val_set.collate(ids, dev) for ids in val_batches), weights, evaluator)
            history.append((it, avg))
            metric_history.append((it, metrics))
            if scheduler is not None and it != it_first:         # train.py:247-251
                scheduler.step(avg) if tc.scheduler.type == "plateau" else scheduler.step()
            improved = best_loss is None or avg < best_loss or it % int(ec.get("force_save_freq", 1000000)) == 0
            if improved:
                best_loss, best_iter = avg, it
                if rank == 0:
                    save_checkpoint(os.path.join(ckpt_dir, "%d.pt" % it), config, model, optimizer, scheduler, it, avg)
            if rank == 0:
                log(f"[validate] iter {it:05d} | loss {avg:.6f} | " + "".join(f"{k} {v:.4f} | " for k, v in metrics.items()) +
                    ("saved" if improved else
                    f"not improved (best {best_loss:.6f} at iter {best_iter})"))
            sharding.barrier()
    return {"model": model, "optimizer": optimizer, "scheduler": scheduler, "history": history, "metrics": metric_history,
            "best_iter": best_iter,
            "ckpt_dir": ckpt_dir}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--logdir", default="./logs")
    ap.add_argument("--tag", default="")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--data", default=None, help="torch file with a list of complex dicts (see module docstring)")
    ap.add_argument("--val_data", default=None, help="validation complexes; default: the last 10 %% of --data")
    ap.add_argument("--synthetic", type=int, default=256, help="number of synthetic complexes when --data is not given")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--finetune", action="store_true")
    ap.add_argument("--max_iters", type=int, default=None)
    args = ap.parse_args(argv)

    rank, world, local = sharding.init_process_group()
    config, config_name = load_config(args.config)
    set_num_atom_type(config)
    if not (args.device.startswith("cuda") and torch.cuda.is_available()):
        raise SystemExit("training needs an MI355X: the message-passing path has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.data:
        raw = torch.load(args.data, map_location="cpu", weights_only=False)
        if args.val_data:
            tr, va = raw, torch.load(args.val_data, map_location="cpu", weights_only=False)
        else:
            k = max(1, len(raw) // 10)
            tr, va = raw[:-k], raw[-k:]
    else:
        raw = synthetic_complexes(max(args.synthetic, 2), int(config.train.get("seed", 2022)), config.model.num_atomtype)
        k = max(1, len(raw) // 10)
        tr, va = raw[:-k], raw[-k:]
    resume = args.resume or config.get("resume", None)
    out = run(config, config_name, ComplexSet(tr), ComplexSet(va), dev, args.logdir, args.tag, resume, args.finetune,
              args.max_iters)
    if rank == 0:
        print(f"done: best validation loss at iteration {out['best_iter']}, checkpoints in {out['ckpt_dir']}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

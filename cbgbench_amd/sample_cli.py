"""Sampling driver: the role of the reference's ``sample.py`` (config -> model -> per-pocket batches ->
``model.sample`` -> one result file per pocket), minus dataset parsing and RDKit reconstruction (out of scope,
SURVEY.md section 2), plus rank sharding of the pocket loop (``sample.py:159`` is embarrassingly parallel).

    python -m cbgbench_amd.sample_cli --config cfg.yml --out_root results [--checkpoint ckpt.pt]
                                      [--pockets pockets.pt | --synthetic 16] [--num_samples 10] [--pockets_per_batch 10]
    python -m cbgbench_amd.launch --nproc 8 -m cbgbench_amd.sample_cli ...      # rank r takes pockets r, r+W, ...; no port to pass

Pocket input: a ``torch.save``d list of dicts with ``protein_pos [n,3]``, ``protein_atom_feature [n,7]``,
``protein_aa_type [n]`` (what ``featurize_protein_fa`` + ``center_pos`` produce, protein_featurizer.py:21-30),
or synthetic pockets.

The task comes from the config's transform list (``priors.SamplingPlan.from_config``): de-novo configs (``assign_molsize``)
sample whole ligands; linker / frag / scaffold / sidechain configs (``assign_gensize`` + ``assign_genatomtype`` + ``center_pos``
on the context atoms + ``assign_genpos``; configs/linker/test/targetdiff.yml:12-28) keep per-pocket CONTEXT atoms fixed and
generate the rest.  Context atoms -- what ``choose_ctx_gen`` + ``remove_ligand_gen`` leave of the native ligand
(select.py:21-88, molecule_featurizer.py:175-205) -- come with the pockets (``ligand_ctx_pos [c,3]``, ``ligand_ctx_atom_type [c]``
in each pocket dict, same frame as ``protein_pos``) or from ``--context`` (a ``torch.save``d list, one ``{'pos', 'atom_type'}``
per pocket); synthetic pockets get synthetic fragments.  Every sample's frame is then centred on its context atoms' mean and
``protein_translation`` records the shift (sample.py:198-201 adds it back: ``sampling.translate``, on in the shipped configs).
Output: ``{out_root}/{tag}/pocket_{i:05d}.pt`` with the final ligand positions, atom types and
(optionally) the trajectory for each sample -- the tensors ``sample.py:198-206`` hands to reconstruction.
A checkpoint is the reference's format: ``{'config': ..., 'model': state_dict}`` (``sample.py:153-156``)."""
import argparse
import os
import time

import numpy as np
import torch

from . import get_model, load_config, priors, set_num_atom_type, sharding, synthetic
from .config import NUM_ATOM_TYPES, get_atomic_number_from_index, is_aromatic_from_index, load_checkpoint_file


def build_pocket_batch(pockets, num_samples, rng, num_classes, prior_types="uniform", device="cpu", num_dist=None,
                       generator=None, plan=None, context=None):
    """num_samples replicas of every pocket with fresh priors (sample.py:177-183; init_lig.py:232-258, 376-432), built
    for the whole batch at once on ``device`` (cbgbench_amd/priors.py).  ``plan`` (priors.SamplingPlan) selects the priors and the
    centring; ``context``: per pocket (pos, atom_type) of the fixed atoms of a linker / frag / scaffold / sidechain job."""
    ps = priors.PocketSet(pockets, device=device, center=False)   # pocket files are already centred (center_pos)
    if plan is None:
        return priors.build_sampling_batch(ps, num_samples, num_classes, num_dist=num_dist, rng=rng, generator=generator,
                                           type_prior=prior_types)
    if plan.task == "context" and context is None:
        raise ValueError("the config asks for a context task (assign_gensize) but no context atoms were given")
    return priors.build_sampling_batch(ps, num_samples, num_classes, num_dist=num_dist, rng=rng, generator=generator,
                                       type_prior=plan.type_prior, pos_prior=plan.pos_prior,
                                       context=context if plan.task == "context" else None,
                                       center_on_context=plan.task == "context" and plan.center == "context")


def load_context(raw_pockets, path):
    """per-pocket context atoms [(pos [c,3] float32, atom_type [c] int64)] from the pocket dicts' ``ligand_ctx_*`` keys or from a
    ``--context`` file (a list with one {'pos', 'atom_type'} dict or (pos, atom_type) pair per pocket); None when neither exists"""
    def one(pos, typ):
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        typ = np.asarray(typ, np.int64).reshape(-1)
        if pos.shape[0] != typ.shape[0]:
            raise ValueError(f"context: {pos.shape[0]} positions but {typ.shape[0]} atom types")
        return pos, typ
    if path:
        raw = torch.load(path, map_location="cpu", weights_only=False)
        out = [one(c["pos"], c["atom_type"]) if isinstance(c, dict) else one(c[0], c[1]) for c in raw]
        if raw_pockets is not None and len(out) != len(raw_pockets):
            raise ValueError(f"--context has {len(out)} entries for {len(raw_pockets)} pockets")
        return out
    if raw_pockets is not None and all("ligand_ctx_pos" in p for p in raw_pockets):
        return [one(p["ligand_ctx_pos"], p["ligand_ctx_atom_type"]) for p in raw_pockets]
    return None


def split_samples(x, c, batch_idx, n_graphs, mode="add_aromatic"):
    """per-graph records like ``split_batch_into_samples`` (sample.py:16-32): ``pos``, ``type`` (index), ``atom`` (atomic
    numbers) and ``aromatic`` (flags, None in 'basic' mode) -- the arguments of ``reconstruct_mol`` (sample.py:211-214) -- plus
    the raw ``type_vector``."""
    out = []
    for g in range(n_graphs):
        m = batch_idx == g
        typ = c[m].argmax(-1) if c.dim() == 2 else c[m]
        out.append({"pos": x[m].clone(), "type": typ.clone(), "atom_type": typ.clone(),
                    "atom": get_atomic_number_from_index(typ.tolist(), mode),
                    "aromatic": is_aromatic_from_index(typ.tolist(), mode), "type_vector": c[m].clone()})
    return out


def decode_mode(plan_mode, config_mode, num_classes):
    """the atom-type vocabulary the samples are decoded with (sample.py:211-214 passes the transform's ``mode`` on to
    ``reconstruct_mol``): the ``mode`` of the config's assign_atomtype / assign_genatomtype transform (``SamplingPlan.mode``), else the
    top-level ``config.mode`` -- and it must be the vocabulary the model's ``num_atomtype`` was built with"""
    for mode in (plan_mode, config_mode):
        if mode in NUM_ATOM_TYPES and NUM_ATOM_TYPES[mode] == num_classes:
            return mode
    raise ValueError(f"no atom-type vocabulary of {num_classes} classes among the config's modes "
                     f"(transform mode {plan_mode!r}, config.mode {config_mode!r}; known: {dict(NUM_ATOM_TYPES)})")


def main(argv=None, stats=None):
    """``stats`` (optional dict): filled with the wall seconds of the phases (setup = config + model + weights, batch = prior
    construction, sample = model.sample incl. the trajectory download, write = per-pocket records and files)."""
    t_phase = time.perf_counter()
    phases = {"setup": 0.0, "batch": 0.0, "sample": 0.0, "write": 0.0}

    def lap(name, dev=None):
        nonlocal t_phase
        if dev is not None and dev.type == "cuda":
            torch.cuda.synchronize(dev)
        now = time.perf_counter()
        phases[name] += now - t_phase
        t_phase = now

    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--out_root", default="./results")
    ap.add_argument("--tag", default="")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--pockets", default=None, help="torch file with a list of pocket dicts")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic pockets when --pockets is not given")
    ap.add_argument("--num_samples", type=int, default=None)
    ap.add_argument("--pockets_per_batch", type=int, default=10)
    ap.add_argument("--streams", type=int, default=3,
                    help="batches kept in flight together on this many HIP streams (model.sample_many; 1 = one batch after the other)")
    ap.add_argument("--random_init", action="store_true",
                    help="sample from randomly initialised weights when no checkpoint is given / found (smoke runs only)")
    ap.add_argument("--save_traj", action="store_true")
    ap.add_argument("--final_state", action="store_true",
                    help="write traj[-1] (the state after the last step) instead of traj[0], which sample.py:198 uses")
    ap.add_argument("--context", default=None,
                    help="linker / frag / scaffold / sidechain configs: torch file with one {'pos' [c,3], 'atom_type' [c]} per pocket "
                         "(the fixed atoms, same frame as protein_pos); default: the pockets' ligand_ctx_pos / ligand_ctx_atom_type")
    ap.add_argument("--no_translate", action="store_true",
                    help="keep results in the sampling frame (default: add protein_translation back when the config's "
                         "sampling.translate is set, sample.py:198-201)")
    ap.add_argument("--atom_num_dist", default=None,
                    help="the reference's size-conditioned ligand-size histogram (repo/datasets/transforms/_atom_num_dist.npy); "
                         "without it ligand sizes are U{10..45}")
    args = ap.parse_args(argv)

    rank, world, local = sharding.init_process_group()
    config, config_name = load_config(args.config)
    set_num_atom_type(config)
    if args.device.startswith("cuda"):
        if not torch.cuda.is_available():
            raise SystemExit("sampling needs an MI355X: the message-passing path has no CPU fallback")
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device(args.device)

    ckpt_path = args.checkpoint or config.model.get("checkpoint", None)
    if ckpt_path and os.path.exists(ckpt_path):
        ckpt = load_checkpoint_file(ckpt_path, map_location="cpu")
        model_cfg = ckpt["config"].model if "config" in ckpt else config.model
        model_cfg.num_atomtype = config.model.num_atomtype
        model = get_model(model_cfg)
        model.load_state_dict(ckpt["model"])            # strict, like sample.py:156
    elif args.random_init:
        # no checkpoints ship with the reference: explicit opt-in, and said loudly
        print(f"[sample_cli] WARNING: --random_init: sampling from a RANDOMLY INITIALISED {config.model.type} model"
              + (f" (checkpoint {ckpt_path!r} not found)" if ckpt_path else ""), flush=True)
        model = get_model(config.model)
    else:
        raise SystemExit(f"sample_cli: checkpoint {ckpt_path!r} does not exist (the reference fails here too, sample.py:150-156); "
                         f"pass --random_init to sample from random weights on purpose" if ckpt_path else
                         "sample_cli: no checkpoint given (--checkpoint or model.checkpoint in the config); pass --random_init "
                         "to sample from random weights on purpose")
    model = model.to(dev).eval()

    plan = priors.SamplingPlan.from_config(config)       # priors, centring and task of the config's transform list
    num_classes = config.model.num_atomtype
    mode = decode_mode(plan.mode, config.get("mode", None), num_classes)
    if args.pockets:
        raw = torch.load(args.pockets, map_location="cpu", weights_only=False)
        pockets = [(np.asarray(p["protein_pos"], np.float32), np.asarray(p["protein_atom_feature"], np.float32),
                    np.asarray(p["protein_aa_type"], np.int64)) for p in raw]
        context = load_context(raw, args.context)
    else:
        rng0 = np.random.default_rng(args.seed)
        pockets = [synthetic.make_pocket(rng0, int(rng0.integers(350, 651))) for _ in range(max(args.synthetic, 1))]
        context = load_context(None, args.context) if args.context else (
            [synthetic.make_context(rng0, int(rng0.integers(10, 36)), num_classes) for _ in pockets] if plan.task == "context" else None)
    if plan.task == "context":
        if context is None:
            raise SystemExit(f"sample_cli: {args.config} is a context task (assign_gensize) -- give the fixed atoms with --context "
                             f"or as ligand_ctx_pos / ligand_ctx_atom_type in the pocket file")
        for k, (_, typ) in enumerate(context):
            if typ.size and (typ.min() < 0 or typ.max() >= num_classes):
                raise SystemExit(f"sample_cli: context atom type out of range [0, {num_classes}) in pocket {k}")
    num_samples = args.num_samples or config.get("sampling", {}).get("num_samples", 10)
    translate = bool(config.get("sampling", {}).get("translate", False)) and not args.no_translate
    num_dist = priors.NumDist.from_npy(args.atom_num_dist) if args.atom_num_dist else None

    mine = sharding.shard_indices(len(pockets), rank, world)
    out_dir = os.path.join(args.out_root, args.tag or config_name)
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(args.seed + rank)                 # independent noise streams per shard
    rng = np.random.default_rng([args.seed, rank])
    def write_results(ids, traj, batch):
        # sample.py:198-201 hands traj[0] to the reconstruction -- for targetdiff / diffbp that is the state entering the
        # last step, not traj[-1]; kept as the default for drop-in outputs, --final_state selects traj[-1]
        x, c, bidx = traj[-1] if (args.final_state and config.model.type != "diffsbdd") else traj[0]
        x, c, bidx = x.cpu(), c.cpu(), bidx.cpu()
        if translate:       # back to the frame the pockets came in (sample.py:198-199; here per graph: a batch holds many pockets)
            x = x + batch["ligand_translation"].cpu()
        samples = split_samples(x, c, bidx, len(ids) * num_samples, mode)
        if "ligand_gen_flag" in batch:      # context tasks: which atoms of a record were generated
            gen = batch["ligand_gen_flag"].cpu()
            for g, smp in enumerate(samples):
                smp["gen_flag"] = gen[bidx == g].clone()
        for k, pid in enumerate(ids):
            rec = {"pocket_index": pid, "samples": samples[k * num_samples:(k + 1) * num_samples]}
            if args.save_traj:
                rec["traj_keys"] = sorted(traj.keys())
            torch.save(rec, os.path.join(out_dir, f"pocket_{pid:05d}.pt"))
        return len(ids) * num_samples * model.num_diffusion_timesteps

    lap("setup", dev)
    timing = {}
    t0, graph_steps = time.perf_counter(), 0
    # `--streams` batches are in flight together (independent pockets: sample.py:159's loop has no order); models without
    # sample_many take them one after the other
    many = getattr(model, "sample_many", None) if (args.streams > 1 and dev.type == "cuda") else None
    group = args.pockets_per_batch * (args.streams if many is not None else 1)
    for g0 in range(0, len(mine), group):
        chunk = [mine[b0:b0 + args.pockets_per_batch] for b0 in range(g0, min(g0 + group, len(mine)), args.pockets_per_batch)]
        batches = [build_pocket_batch([pockets[i] for i in ids], num_samples, rng, num_classes, device=dev, num_dist=num_dist,
                                      plan=plan, context=[context[i] for i in ids] if context is not None else None)
                   for ids in chunk]
        lap("batch", dev)
        # (targetdiff itemises its share of the `sample` phase: static context, the T steps, the trajectory download)
        tk = {"timing": timing} if (stats is not None and config.model.type == "targetdiff") else {}
        trajs = (many(batches, streams=args.streams, **tk) if many is not None and len(batches) > 1
                 else [model.sample(b, **tk) for b in batches])
        lap("sample", dev)
        for ids, traj, b in zip(chunk, trajs, batches):
            graph_steps += write_results(ids, traj, b)
        lap("write")
    if dev.type == "cuda":
        torch.cuda.synchronize()
    sharding.barrier()
    el, gs = sharding.reduce_max_sum(time.perf_counter() - t0, graph_steps, device=dev)
    if stats is not None:
        stats.update(phases)
        stats.update(timing)
    if rank == 0:
        print(f"sampled {len(pockets)} pockets x {num_samples} samples on {world} rank(s): "
              f"{gs / el:.1f} graph-steps/s, results in {out_dir}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

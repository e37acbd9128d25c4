"""Synthetic pocket+ligand batches in the reference batch schema.

No CrossDocked data (LMDB) is available offline, so benchmarks and parity tests
use synthetic pockets with the statistics SURVEY.md section 8(d) lists:
N_rec ~ U{350..650} heavy atoms inside a 12 A ball at >= 1.2 A spacing, centred
on the protein mean (translation.py:18), protein features = one-hot(element in
[1,6,7,8,16,34]) ++ is_backbone (protein_featurizer.py:21-26), aa type in
[0,20); ligand prior = N(0,I) positions (init_lig.py:424-425) and uniform
categorical types (init_lig.py:392-394).

The batch is a plain dict of torch tensors with the keys ``MergeKeys`` +
``follow_batch`` produce in the reference (merge.py:16-25).
"""
import numpy as np
import torch

_LATTICE_CACHE = {}


def _lattice(radius, spacing):
    key = (radius, spacing)
    if key not in _LATTICE_CACHE:
        n = int(np.ceil(radius / spacing))
        g = np.arange(-n, n + 1) * spacing
        p = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        _LATTICE_CACHE[key] = p[(p ** 2).sum(-1) <= radius ** 2]
    return _LATTICE_CACHE[key]


def make_pocket(rng, n_rec, radius=12.0, spacing=1.5, jitter=0.15):
    """Protein atoms: a random subset of a cubic lattice with jitter (min spacing 1.2 A)."""
    lat = _lattice(radius, spacing)
    idx = rng.choice(lat.shape[0], size=n_rec, replace=False)
    pos = lat[idx] + rng.uniform(-jitter, jitter, size=(n_rec, 3))
    pos = pos - pos.mean(0, keepdims=True)
    elem = rng.choice(6, size=n_rec, p=[0.0, 0.62, 0.17, 0.19, 0.02, 0.0])
    feat = np.zeros((n_rec, 7), np.float32)
    feat[np.arange(n_rec), elem] = 1.0
    feat[:, 6] = rng.random(n_rec) < 0.45
    aa = rng.integers(0, 20, size=n_rec)
    return pos.astype(np.float32), feat, aa.astype(np.int64)


def make_context(rng, n_ctx, num_classes=13, offset=2.5, radius=1.5):
    """Fixed atoms of a linker-style job (what choose_ctx_gen + remove_ligand_gen leave of a native ligand, select.py:21-88):
    two fragments at +-``offset`` A from a random point near the pocket centre, in the pocket's frame.
    -> (pos [n_ctx,3] float32, atom_type [n_ctx] int64)"""
    c0 = rng.standard_normal(3) * 1.5
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    half = n_ctx // 2
    pos = np.concatenate([c0 + offset * axis + rng.standard_normal((half, 3)) * radius / 2,
                          c0 - offset * axis + rng.standard_normal((n_ctx - half, 3)) * radius / 2])
    return pos.astype(np.float32), rng.integers(0, num_classes, size=n_ctx).astype(np.int64)


def make_batch(pockets, n_lig_list, rng, num_classes=13, n_ctx_list=None, ctx_radius=3.0):
    """Collate pockets [(pos, feat, aa), ...] with fresh ligand priors.

    ``n_ctx_list`` (linker/frag/scaffold-style tasks): per graph, the first
    n_ctx ligand atoms are fixed context atoms (gen_flag False) placed around the
    origin; the remaining n_lig - n_ctx atoms are generated (init_lig.py:260-297).
    """
    ppos, pfeat, paa, pb, lpos, ltyp, lb, lgen = [], [], [], [], [], [], [], []
    for g, ((pos, feat, aa), n_lig) in enumerate(zip(pockets, n_lig_list)):
        ppos.append(pos); pfeat.append(feat); paa.append(aa)
        pb.append(np.full(pos.shape[0], g, np.int64))
        lp = rng.standard_normal((n_lig, 3)).astype(np.float32)
        gen = np.ones(n_lig, bool)
        if n_ctx_list is not None:
            nc = int(n_ctx_list[g])
            lp[:nc] = (rng.standard_normal((nc, 3)) * ctx_radius / 2).astype(np.float32)
            gen[:nc] = False
        lpos.append(lp)
        ltyp.append(rng.integers(0, num_classes, size=n_lig).astype(np.int64))
        lb.append(np.full(n_lig, g, np.int64))
        lgen.append(gen)
    n_rec = sum(p.shape[0] for p in ppos)
    n_lig = sum(p.shape[0] for p in lpos)
    batch = {
        "protein_pos": torch.from_numpy(np.concatenate(ppos)),
        "protein_atom_feature": torch.from_numpy(np.concatenate(pfeat)),
        "protein_aa_type": torch.from_numpy(np.concatenate(paa)),
        "protein_lig_flag": torch.zeros(n_rec, dtype=torch.bool),
        "protein_element_batch": torch.from_numpy(np.concatenate(pb)),
        "ligand_pos": torch.from_numpy(np.concatenate(lpos)),
        "ligand_atom_type": torch.from_numpy(np.concatenate(ltyp)),
        "ligand_lig_flag": torch.ones(n_lig, dtype=torch.bool),
        "ligand_element_batch": torch.from_numpy(np.concatenate(lb)),
    }
    if n_ctx_list is not None:
        batch["ligand_gen_flag"] = torch.from_numpy(np.concatenate(lgen))
    return batch


def denovo_batch(n_graphs, seed=0, n_rec_range=(350, 650), n_lig_range=(10, 45), num_classes=13,
                 same_pocket=False):
    """Config-2 style batch: ``n_graphs`` samples (``same_pocket`` => one pocket replicated,
    as sample.py:177 does; else distinct pockets), ligand sizes U{10..45}."""
    rng = np.random.default_rng(seed)
    if same_pocket:
        p = make_pocket(rng, int(rng.integers(n_rec_range[0], n_rec_range[1] + 1)))
        pockets = [p] * n_graphs
    else:
        pockets = [make_pocket(rng, int(rng.integers(n_rec_range[0], n_rec_range[1] + 1)))
                   for _ in range(n_graphs)]
    n_lig = rng.integers(n_lig_range[0], n_lig_range[1] + 1, size=n_graphs)
    return make_batch(pockets, n_lig, rng, num_classes)


def linker_batch(n_graphs, seed=0, n_rec_range=(350, 650), n_ctx_range=(10, 35), n_gen_range=(3, 14),
                 num_classes=13):
    """Config-3 style batch: distinct pockets, fixed context atoms + a few generated linker atoms."""
    rng = np.random.default_rng(seed)
    pockets = [make_pocket(rng, int(rng.integers(n_rec_range[0], n_rec_range[1] + 1)))
               for _ in range(n_graphs)]
    n_ctx = rng.integers(n_ctx_range[0], n_ctx_range[1] + 1, size=n_graphs)
    n_gen = rng.integers(n_gen_range[0], n_gen_range[1] + 1, size=n_graphs)
    return make_batch(pockets, n_ctx + n_gen, rng, num_classes, n_ctx_list=n_ctx)


def batch_to(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}

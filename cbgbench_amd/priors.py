"""Vectorised prior sampling and batch construction for sampling runs.

The reference builds a sampling batch pocket by pocket and sample by sample in Python: ``sample.py:177-183``
replicates one ``Data`` object ``num_samples`` times, each replica runs the transform chain
(``assign_molsize`` -> ``assign_atomtype`` -> ``assign_molpos`` ..., ``repo/datasets/transforms/init_lig.py:232-258,
376-432``; ``center_pos`` ``translation.py:5-25``; ``MergeKeys`` ``merge.py:6-25``) and a PyG ``DataLoader`` collates them.
At thousands of graph-steps per second that host loop becomes the bottleneck, so here the same quantities are
produced for a whole batch of P pockets x S samples at once, on the device the tensors will live on:

* pocket size           ``space_size``: median of the 10 largest pairwise distances (``init_lig.py:255-258``)
* ligand atom count     ``NumDist.sample``: the reference's size-conditioned histogram ``_atom_num_dist.npy``
                        (``sample_atom_num``/``_get_bin_idx``, ``init_lig.py:28-52``); the table is data of the
                        reference, read from a path the caller provides (same pickle format), never copied here
* ligand prior          positions N(0, I) (``assign_molpos: gaussian``), types uniform / absorbing / zeros
                        (``assign_atomtype``), context atoms of linker-style tasks kept (``assign_gensize``)
* the batch             concatenated tensors + ``*_element_batch`` ids in the schema ``MergeKeys`` + ``follow_batch``
                        produce, built with index arithmetic instead of a collate loop.

The random streams necessarily differ from the reference's (numpy global RNG per replica); the distributions are the
same and are tested against the reference's tables and size function (tests/test_priors.py).
"""
import numpy as np
import torch

ABSORBING_STATE = 0     # repo/utils/molecule/constants.py:8
PROTEIN_ELEMENTS = torch.tensor([1, 6, 7, 8, 16, 34])     # repo/utils/protein/constants.py:37 (atomic_numbers)


class NumDist:
    """size-conditioned histogram: ``bounds`` (ascending bin edges) and, per bin, (values, probabilities)."""

    def __init__(self, bounds, bins):
        self.bounds = np.asarray(bounds, dtype=np.float64)
        self.values = [np.asarray(v, dtype=np.int64) for v, _ in bins]
        self.cdfs = []
        for _, p in bins:
            p = np.asarray(p, dtype=np.float64)
            self.cdfs.append(np.cumsum(p / p.sum()))
        if len(self.values) != len(self.bounds) + 1:
            raise ValueError("NumDist: need len(bounds) + 1 bins")

    @classmethod
    def from_npy(cls, path):
        """the reference's ``_atom_num_dist.npy`` / ``_linker_num_dist.npy`` (a pickled {'bounds', 'bins'} dict)"""
        d = np.load(path, allow_pickle=True).item()
        return cls(d["bounds"], d["bins"])

    @classmethod
    def uniform(cls, lo, hi):
        """fallback when the reference's table is not at hand: one bin, U{lo..hi}"""
        v = np.arange(lo, hi + 1)
        return cls([], [(v, np.full(v.shape, 1.0 / v.size))])

    def bin_index(self, space_size):
        """first i with bounds[i] > size, else len(bounds)  (``_get_bin_idx``, init_lig.py:47-52)"""
        return np.searchsorted(self.bounds, np.asarray(space_size, dtype=np.float64), side="right")

    def sample(self, space_size, n_samples, rng):
        """[P] sizes -> [P, n_samples] counts, inverse-CDF sampling per bin (``np.random.choice(values, p=probs)``)."""
        idx = np.atleast_1d(self.bin_index(space_size))
        out = np.empty((idx.shape[0], n_samples), dtype=np.int64)
        u = rng.random((idx.shape[0], n_samples))
        for b in np.unique(idx):
            rows = np.nonzero(idx == b)[0]
            k = np.searchsorted(self.cdfs[b], u[rows], side="right").clip(max=self.values[b].size - 1)
            out[rows] = self.values[b][k]
        return out


def space_size(pos):
    """``AssignMolSize.get_space_size`` (init_lig.py:255-258): torch.median of the 10 largest pairwise distances, i.e.
    their lower median = the 6th largest distance.  ``pos`` [n,3] on any device; n >= 5 (>= 10 pairs)."""
    d = torch.pdist(pos) if pos.shape[0] <= 2048 else torch.cdist(pos, pos).triu(1).flatten()
    top = torch.topk(d, k=min(10, d.numel())).values
    return torch.median(top)



class SamplingPlan:
    """What the transform list of a sampling config (``config.data.test.transform``) asks the batch builder for -- the part of
    the reference's per-sample transform chain that defines the ligand prior:

    ======================  ============================================================  =====================================
    transform               reference                                                     field
    ======================  ============================================================  =====================================
    assign_molsize          init_lig.py:232-258   ligand size from the pocket size        task = 'denovo', size_dist
    assign_gensize          init_lig.py:260-302   context atoms kept, generated appended  task = 'context', size_dist
    assign_atomtype         init_lig.py:376-412   uniform | absorbing | gaussian | zeros  type_prior, mode
    assign_genatomtype      init_lig.py:306-341   uniform | absorbing | gaussian          type_prior, mode
    assign_molpos/_genpos   init_lig.py:415-432, 455-471  gaussian | zero_mean_gaussian   pos_prior
    center_pos              translation.py:5-25   center_flag, mask_flag                  center = 'protein' | 'context' | ...
    center_whole_pos        translation.py:27-50  (ligand removed: the protein mean)      center = 'protein'
    ======================  ============================================================  =====================================

    The five task families differ in exactly these lines (configs/{denovo,linker,frag,scaffold,sidechain}/test/*.yml); which
    ligand atoms are context is the dataset's business (``choose_ctx_gen`` + ``remove_ligand_gen``) and reaches the driver as
    per-pocket context atoms.  Unknown distributions raise ``ValueError('Unknown distribution type: ...')`` as the transforms do."""

    SIZE_DISTS = ("prior_distcond",)
    TYPE_PRIORS = {"assign_atomtype": ("uniform", "absorbing", "gaussian", "zeros"),
                   "assign_genatomtype": ("uniform", "absorbing", "gaussian")}
    POS_PRIORS = ("gaussian", "zero_mean_gaussian")

    def __init__(self, task="denovo", size_dist="prior_distcond", type_prior="uniform", pos_prior="gaussian", center="protein",
                 mode="add_aromatic"):
        self.task, self.size_dist, self.type_prior, self.pos_prior, self.center, self.mode = \
            task, size_dist, type_prior, pos_prior, center, mode

    def __repr__(self):
        return (f"SamplingPlan(task={self.task!r}, size_dist={self.size_dist!r}, type_prior={self.type_prior!r}, "
                f"pos_prior={self.pos_prior!r}, center={self.center!r}, mode={self.mode!r})")

    @classmethod
    def for_model(cls, model_type, mode="add_aromatic"):
        """the de-novo plan of configs/denovo/test/{targetdiff,diffbp,diffsbdd}.yml for a config without a transform list"""
        prior = {"diffbp": "absorbing", "diffsbdd": "zeros"}.get(model_type, "uniform")
        return cls("denovo", "prior_distcond", prior, "zero_mean_gaussian" if model_type == "diffsbdd" else "gaussian",
                   "protein", mode)

    @classmethod
    def from_config(cls, config):
        """plan of ``config.data.test.transform`` (falls back to ``for_model`` when the config has no transform list)"""
        data = config.get("data", None) or {}
        split = data.get("test", None) or data.get("train", None) or {}
        tlist = split.get("transform", None)
        model_type = config.get("model", {}).get("type", "targetdiff")
        if not tlist:
            return cls.for_model(model_type, config.get("mode", "add_aromatic"))
        plan = cls.for_model(model_type, config.get("mode", "add_aromatic"))
        seen = set()
        for t in tlist:
            ty = t["type"] if isinstance(t, dict) else t.type
            seen.add(ty)
            dist = t.get("distribution", None)
            if ty in ("assign_molsize", "assign_gensize"):
                plan.task = "denovo" if ty == "assign_molsize" else "context"
                plan.size_dist = dist if dist is not None else "prior_distcond"
                if plan.size_dist not in cls.SIZE_DISTS:
                    raise ValueError(f"Unknown distribution type: {plan.size_dist}")
            elif ty in cls.TYPE_PRIORS:
                plan.type_prior = dist if dist is not None else "uniform"
                if plan.type_prior not in cls.TYPE_PRIORS[ty]:
                    raise ValueError(f"Unknown distribution type: {plan.type_prior}")
                plan.mode = t.get("mode", plan.mode)
            elif ty in ("assign_molpos", "assign_genpos"):
                plan.pos_prior = dist if dist is not None else "gaussian"
                if plan.pos_prior not in cls.POS_PRIORS:
                    raise ValueError(f"Unknown distribution type: {plan.pos_prior}")
            elif ty == "center_pos":
                flag, mask = t.get("center_flag", "protein"), t.get("mask_flag", None)
                if flag == "protein" and mask is None:
                    plan.center = "protein"
                elif flag == "ligand" and mask == "ctx_flag":
                    plan.center = "context"
                else:
                    raise ValueError(f"center_pos: center_flag={flag!r} mask_flag={mask!r} is not a sampling-time centring "
                                     f"this driver knows (protein | ligand + ctx_flag)")
            elif ty == "center_whole_pos":
                plan.center = "protein"        # sampling configs remove the ligand first: the mean of the protein atoms
        if "assign_molsize" in seen and "assign_gensize" in seen:
            raise ValueError("transform list has both assign_molsize and assign_gensize")
        return plan


class PocketSet:
    """P pockets packed once (CSR): what is constant across all samples and all steps of a sampling run."""

    def __init__(self, pockets, device="cpu", center=True):
        """``pockets``: list of (pos [n,3], atom_feature [n,7], aa_type [n]) arrays/tensors in original coordinates.
        ``center``: subtract each pocket's mean (``center_pos`` with center_flag='protein', translation.py:5-25) and
        remember it as the translation that ``translate`` adds back to the results (sample.py:198-201)."""
        as_t = lambda a, dt: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dt)
        pos = [as_t(p[0], torch.float32) for p in pockets]
        sizes = torch.tensor([p.shape[0] for p in pos], dtype=torch.long)
        self.ptr = torch.zeros(len(pos) + 1, dtype=torch.long)
        self.ptr[1:] = torch.cumsum(sizes, 0)
        self.num_pockets = len(pos)
        centers = torch.stack([p.mean(0) if center else torch.zeros(3) for p in pos])
        self.center = centers.to(device)
        self.pos = torch.cat([p - c for p, c in zip(pos, centers)]).to(device)
        self.atom_feature = torch.cat([as_t(p[1], torch.float32) for p in pockets]).to(device)
        self.aa_type = torch.cat([as_t(p[2], torch.long) for p in pockets]).to(device)
        self.sizes = sizes
        self.space = np.array([float(space_size(self.pos[self.ptr[k]:self.ptr[k + 1]])) for k in range(len(pos))])
        self.device = torch.device(device)


def _segment_ids(counts, device):
    counts = torch.as_tensor(counts, dtype=torch.long, device=device)
    return torch.repeat_interleave(torch.arange(counts.numel(), device=device), counts)


def build_sampling_batch(pocket_set, num_samples, num_classes, num_dist=None, generator=None, rng=None,
                         type_prior="uniform", pos_prior="gaussian", n_lig=None, context=None, gen_dist=None,
                         center_on_context=False):
    """One batch of P x S graphs (graph g = p * S + s, all samples of a pocket adjacent, as sample.py builds them).

    ``num_dist``: NumDist for the ligand size (default U{10..45}, see NumDist.uniform); ``n_lig`` [P,S] overrides it.
    ``context`` (linker / frag / scaffold tasks): per pocket (pos [c,3], atom_type [c]) of the fixed atoms, in original
    coordinates; every sample then has the context atoms first (gen_flag False) followed by generated atoms whose
    number is ``size - c`` or, when that is not positive, c + U{1..7} in total (``assign_gensize``, init_lig.py:267-297).
    ``center_on_context``: the frame of every graph is centred on the mean of its pocket's context atoms (``center_pos`` with
    ``center_flag: ligand, mask_flag: ctx_flag``, translation.py:5-25, as configs/{linker,frag,scaffold,sidechain}/test/*.yml
    ask) instead of on the pocket set's own centre; a pocket without context atoms keeps the frame it came in (the reference
    then averages the all-zero positions ``assign_gensize`` has just written).  ``*_translation`` is the total shift either way.
    Returns the batch dict on ``pocket_set.device``."""
    ps, S, dev = pocket_set, int(num_samples), pocket_set.device
    P = ps.num_pockets
    rng = rng if rng is not None else np.random.default_rng()
    if n_lig is None:
        dist = num_dist if num_dist is not None else NumDist.uniform(10, 45)
        n_lig = dist.sample(ps.space, S, rng)
    n_lig = np.asarray(n_lig, dtype=np.int64).reshape(P, S)
    n_ctx = np.zeros(P, dtype=np.int64)
    if context is not None:
        n_ctx = np.array([np.asarray(c[1]).shape[0] for c in context], dtype=np.int64)
        short = n_lig <= n_ctx[:, None]
        n_lig = np.where(short, n_ctx[:, None] + rng.integers(1, 8, size=n_lig.shape), n_lig)
    # ---- protein side: every graph repeats its pocket's rows --------------------------------------------
    g_pocket = torch.arange(P, device=dev).repeat_interleave(S)                    # pocket of graph g
    rec_counts = ps.sizes.to(dev)[g_pocket]
    rec_graph = _segment_ids(rec_counts, dev)
    rec_start = torch.cumsum(rec_counts, 0) - rec_counts
    rows = ps.ptr.to(dev)[g_pocket][rec_graph] + (torch.arange(rec_graph.numel(), device=dev) - rec_start[rec_graph])
    n_rec = rows.numel()
    # ---- ligand side -----------------------------------------------------------------------------------------
    lig_counts = torch.as_tensor(n_lig.reshape(-1), device=dev)
    lig_graph = _segment_ids(lig_counts, dev)
    n_tot = int(lig_counts.sum())
    lig_start = torch.cumsum(lig_counts, 0) - lig_counts
    local = torch.arange(n_tot, device=dev) - lig_start[lig_graph]                # index of the atom inside its ligand
    if pos_prior not in ("gaussian", "zero_mean_gaussian"):
        raise ValueError(f"Unknown distribution type: {pos_prior}")
    pos = torch.randn(n_tot, 3, device=dev, generator=generator)
    if type_prior == "uniform":
        typ = torch.randint(0, num_classes, (n_tot,), device=dev, generator=generator)
    elif type_prior == "absorbing":
        typ = torch.full((n_tot,), ABSORBING_STATE, dtype=torch.long, device=dev)
    elif type_prior == "zeros":
        typ = torch.zeros(n_tot, num_classes, device=dev)
    elif type_prior == "gaussian":
        typ = torch.randn(n_tot, num_classes, device=dev, generator=generator)
    else:
        raise ValueError(f"Unknown distribution type: {type_prior}")
    batch = {}
    shift = torch.zeros(P, 3, device=dev)            # additional per-pocket shift of the frame (context centring)
    if context is not None:
        ctx_cnt = torch.as_tensor(n_ctx, device=dev)
        ctx_ptr = torch.zeros(P + 1, dtype=torch.long, device=dev)
        ctx_ptr[1:] = torch.cumsum(ctx_cnt, 0)
        ctx_cpu = [torch.as_tensor(np.asarray(c[0]), dtype=torch.float32).reshape(-1, 3) for c in context]
        ctx_seg = _segment_ids(ctx_cnt, dev)
        ctx_pos = torch.cat(ctx_cpu).to(dev) - ps.center[ctx_seg]
        if center_on_context:
            # per pocket the fp32 mean the reference takes (data.ligand.pos[ctx_flag].mean(dim=0), translation.py:11-12), on the
            # host in the same order; with an un-centred PocketSet the frame is then bit-identical to the reference's
            cen = ps.center.cpu()
            shift = torch.stack([(c - cen[k]).mean(dim=0) if c.shape[0] else torch.zeros(3) for k, c in enumerate(ctx_cpu)]).to(dev)
            ctx_pos = ctx_pos - shift[ctx_seg]
        ctx_typ = torch.cat([torch.as_tensor(np.asarray(c[1]), dtype=torch.long) for c in context]).to(dev)
        is_ctx = local < ctx_cnt[g_pocket][lig_graph]
        src = (ctx_ptr[g_pocket][lig_graph] + local)[is_ctx]
        pos[is_ctx] = ctx_pos[src]
        if typ.dim() == 1:
            typ[is_ctx] = ctx_typ[src]
        else:
            typ[is_ctx] = torch.nn.functional.one_hot(ctx_typ[src], num_classes).to(typ.dtype)
        batch["ligand_gen_flag"] = ~is_ctx
        batch["ligand_ctx_flag"] = is_ctx
        # atomic numbers: the context atoms' own when the caller has them (a third array per pocket), zeros for generated atoms as
        # assign_gensize leaves them (init_lig.py:283-289)
        elem = torch.zeros(n_tot, dtype=torch.long, device=dev)
        if all(len(c) > 2 and c[2] is not None for c in context):
            elem[is_ctx] = torch.cat([torch.as_tensor(np.asarray(c[2]), dtype=torch.long).reshape(-1) for c in context]).to(dev)[src]
        batch["ligand_element"] = elem
    if pos_prior == "zero_mean_gaussian":
        mean = torch.zeros(P * S, 3, device=dev).index_add_(0, lig_graph, pos) / lig_counts.clamp(min=1)[:, None]
        pos = pos - mean[lig_graph]
    batch.setdefault("ligand_element", torch.zeros(n_tot, dtype=torch.long, device=dev))     # assign_molsize (init_lig.py:252)
    batch.update({
        "protein_pos": ps.pos[rows] - shift[g_pocket][rec_graph] if center_on_context else ps.pos[rows],
        "protein_atom_feature": ps.atom_feature[rows],
        "protein_aa_type": ps.aa_type[rows],
        "protein_lig_flag": torch.zeros(n_rec, dtype=torch.bool, device=dev),
        # atomic numbers from the one-hot element columns of the atom feature (protein_featurizer.py:21-26)
        "protein_element": PROTEIN_ELEMENTS.to(dev)[ps.atom_feature[rows][:, :6].argmax(-1)],
        "protein_element_batch": rec_graph,
        "protein_translation": (ps.center + shift)[g_pocket][rec_graph],
        "ligand_pos": pos,
        "ligand_atom_type": typ,
        "ligand_lig_flag": torch.ones(n_tot, dtype=torch.bool, device=dev),
        "ligand_element_batch": lig_graph,
        "ligand_translation": (ps.center + shift)[g_pocket][lig_graph],
    })
    return batch


def translate(pos, batch_idx, pocket_set, num_samples):
    """results back to the original frame: pos + translation of the graph's pocket (sample.py:198-201)."""
    g_pocket = torch.arange(pocket_set.num_pockets, device=pos.device).repeat_interleave(num_samples)
    return pos + pocket_set.center.to(pos.device)[g_pocket][batch_idx]

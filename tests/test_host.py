"""Host logic + C-ABI surface, no GPU needed."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import cbgbench_amd as C
from cbgbench_amd import _native, sharding
from oracle import targetdiff as OT
from oracle import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rdzv_file():
    """the FileStore path the ranks of one test job meet on (cbgbench_amd.launch's rendezvous: no port to collide on)"""
    from cbgbench_amd import launch
    return os.path.join(launch.rendezvous_dir(), "store")


def test_config_include_and_num_atomtype():
    cfg, name = C.load_config(os.path.join(ROOT, "tests", "fixtures", "targetdiff_test.yml"))
    assert name == "targetdiff_test"
    assert cfg.model.generator.pos_schedule.beta_end == pytest.approx(2e-3)
    assert cfg.model.encoder.get("k", 32) == 32
    C.set_num_atom_type(cfg)
    assert cfg.model.num_atomtype == 13 and cfg.mode == "add_aromatic"
    C.set_num_atom_type(cfg, 8)
    assert cfg.model.num_atomtype == 8


def test_registry_and_factory_errors():
    assert "targetdiff" in C.registered_models()
    with pytest.raises(KeyError):
        C.get_model(C.Config(type="nope"))
    with pytest.raises(ValueError, match="Unknown model type"):
        C.get_e3_gnn(C.Config(type="gvptransformer"))
    with pytest.raises(ValueError, match="Not supported cutoff mode"):
        C.get_e3_gnn(C.Config(type="unitransformer", cutoff_mode="radius"), num_classes=13)
    with pytest.raises(ValueError, match="n_heads"):
        C.get_e3_gnn(C.Config(type="unitransformer", n_heads=8), num_classes=13)


@pytest.mark.parametrize("tag,Cn", [("add_aromatic", 13), ("basic", 8)])
def test_state_dict_is_reference_compatible(golden_dir, tag, Cn):
    model = C.get_model(C.default_targetdiff_config(Cn))
    with open(os.path.join(golden_dir, f"state_dict_keys_{tag}.json")) as f:
        ref = json.load(f)
    sd = model.state_dict()
    assert set(sd) == set(ref)
    for k, shp in ref.items():
        assert list(sd[k].shape) == shp, k
    model.load_state_dict(W.synthetic_state_dict(Cn, 9), strict=True)
    # frozen tables are not trainable; 2 671 774 trainable parameters for C=13
    if Cn == 13:
        assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 2671774
        assert sum(p.numel() for p in model.parameters()) == 2699774


def test_schedule_tables_bitexact(golden_dir):
    model = C.get_model(C.default_targetdiff_config(13))
    z = np.load(os.path.join(golden_dir, "schedule_tables.npz"))
    sd = model.state_dict()
    for k in z.files:
        assert np.array_equal(z[k], sd[k].numpy()), k


@pytest.mark.parametrize("case", ["step_t500", "step_t0", "step_t999_linker"])
def test_posterior_updates_match_reference(golden_dir, case):
    """The host-side step update (pos + type posterior) on the reference's network outputs."""
    z = np.load(os.path.join(golden_dir, case + ".npz"))
    g = {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}
    model = C.get_model(C.default_targetdiff_config(13))
    bl = g["batch_ligand_element_batch"]
    gen = g["batch_ligand_gen_flag"] if "batch_ligand_gen_flag" in g else g["batch_ligand_lig_flag"]
    B = int(bl.max()) + 1
    t = torch.full((B,), int(g["t_idx"]), dtype=torch.long)
    c_lig = torch.nn.functional.one_hot(g["batch_ligand_atom_type"], 13).float()
    x_next = model.pos_scheduler.backward_remove_noise(g["x_pred"], g["batch_ligand_pos"], t, bl, gen, type="denoise",
                                                       noise=g["eps"])
    c_next, v_next = model.type_scheduler.backward_remove_noise(g["c_pred"], c_lig, t, bl, gen, uniform=g["u"])
    assert torch.equal(x_next, g["x_next"])
    assert torch.equal(v_next, g["v_next"]) and torch.equal(c_next, g["c_next"])


def test_compose_plan_matches_oracle():
    bl = torch.tensor([0, 0, 1, 1, 1, 2])
    br = torch.tensor([0, 0, 0, 1, 2, 2])
    sort_idx, batch_idx, lig_flag, lig_rows, gp = C.TargetDiff.compose_plan(bl, br)
    o_sort, o_batch = OT.compose(bl, br)
    assert torch.equal(sort_idx, o_sort) and torch.equal(batch_idx, o_batch)
    assert gp.tolist() == [0, 5, 9, 12] and gp.dtype == torch.int32
    assert lig_rows.tolist() == [3, 4, 6, 7, 8, 11]
    # protein rows first inside every graph
    for g in range(3):
        seg = lig_flag[gp[g]:gp[g + 1]]
        assert torch.equal(seg, seg.sort().values)


def test_embedder_matches_oracle():
    torch.manual_seed(0)
    model = C.get_model(C.default_targetdiff_config(13))
    sd = W.synthetic_state_dict(13, 9)
    model.load_state_dict(sd)
    c = torch.nn.functional.one_hot(torch.randint(0, 13, (7,)), 13).float()
    v = torch.rand(9, 7)
    aa = torch.nn.functional.one_hot(torch.randint(0, 20, (9,)), 20).float()
    hl, hr = OT.context_embed(sd, c, v, aa)
    with torch.no_grad():
        assert torch.allclose(model.context_embedder.embed_ligand(c), hl, atol=1e-6)
        assert torch.allclose(model.context_embedder.embed_protein(v, aa), hr, atol=1e-6)


def test_compose_embed_host_path_is_embedder_then_compose_context():
    """targetdiff.compose_embed on the host = PLContextEmbedder on both atom sets, then compose_context of coordinates, features and the
    movable flag (targetdiff.py:89-101 of the reference; the oracle's restatement): what the fused kernel (cbgx_embed_compose, GPU tests)
    is compared with.  Also the stacked-weight identity the kernel's backward rests on: h = ext . Wext for the extended input rows."""
    from cbgbench_amd.targetdiff import TargetDiff, compose_embed
    torch.manual_seed(1)
    model = C.get_model(C.default_targetdiff_config(13))
    sd = W.synthetic_state_dict(13, 9)
    model.load_state_dict(sd)
    emb = model.context_embedder
    n_rec, n_lig, B = 23, 7, 3
    br, bl = torch.sort(torch.randint(0, B, (n_rec,))).values, torch.sort(torch.randint(0, B, (n_lig,))).values
    x_rec, x_lig = torch.randn(n_rec, 3), torch.randn(n_lig, 3)
    feat, aa = torch.rand(n_rec, 7), torch.randint(0, 20, (n_rec,))
    c = torch.nn.functional.one_hot(torch.randint(0, 13, (n_lig,)), 13).float()
    gen_r, gen_l = torch.zeros(n_rec, dtype=torch.bool), torch.rand(n_lig) < 0.7
    sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TargetDiff.compose_plan(bl, br, B)
    with torch.no_grad():
        x, h, g = compose_embed(emb, x_rec, x_lig, feat, aa, c, sort_idx, gen_r, gen_l)
    hl, hr = OT.context_embed(sd, c, feat, torch.nn.functional.one_hot(aa, 20).float())
    assert torch.equal(x, torch.cat([x_rec, x_lig])[sort_idx]) and torch.equal(g, torch.cat([gen_r, gen_l])[sort_idx])
    assert torch.allclose(h, torch.cat([hr, hl])[sort_idx], atol=1e-6)
    assert torch.equal(batch_idx, torch.cat([br, bl])[sort_idx]) and torch.equal(lig_flag, sort_idx >= n_rec)
    assert torch.equal(sort_idx[lig_rows], n_rec + torch.arange(n_lig)) and int(graph_ptr[-1]) == n_rec + n_lig
    # extended rows and stacked weights (csrc/train_embed.hip): [feat | onehot aa | 1 | c | 1] . [W_pa^T ; W_res^T ; b_p ; W_la^T ; b_l]
    with torch.no_grad():
        ext = torch.zeros(n_rec + n_lig, 7 + 20 + 1 + 13 + 1)
        ext[:n_rec, :7], ext[:n_rec, 27] = feat, 1.0
        ext[torch.arange(n_rec), 7 + aa] = 1.0
        ext[n_rec:, 28:41], ext[n_rec:, 41] = c, 1.0
        b_p = emb.protein_atom_emb.bias + emb.residue_emb.bias + emb.ligand_indicator.bias
        b_l = emb.ligand_atom_emb.bias + emb.ligand_indicator.weight[:, 0] + emb.ligand_indicator.bias
        wext = torch.cat([emb.protein_atom_emb.weight.T, emb.residue_emb.weight.T, b_p[None], emb.ligand_atom_emb.weight.T, b_l[None]])
        assert torch.allclose((ext @ wext)[sort_idx], h, atol=1e-5)


def test_no_cpu_fallback():
    model = C.get_model(C.default_targetdiff_config(13))
    x = torch.zeros(4, 3); h = torch.zeros(4, 128)
    b = torch.zeros(4, dtype=torch.long); f = torch.zeros(4, dtype=torch.bool)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        model.denoiser(x=x, h=h, batch_idx=b, lig_flag=f, gen_flag=f)
    # the training entry point (targetdiff.py:40-60) reaches the same denoiser: no CPU path either
    import numpy as np
    from cbgbench_amd import synthetic
    rng = np.random.default_rng(0)
    batch = synthetic.make_batch([synthetic.make_pocket(rng, 40, radius=6.0)], [6], rng, 13)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.train()(batch)


# ---- C ABI ---------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cbgx.h")).read()
    declared = set(re.findall(r"\b(cbgx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    lib = _native.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.cbgx_abi_version() == _native.ABI_VERSION == 6
    # the product library carries no debug switch and none of the first-generation kernels; the test-only build has both
    import subprocess
    assert not hasattr(lib, "cbgx_debug_set_edge_kernel")
    from cbgbench_amd.build import LIBPATH, XCHECK_LIBPATH
    sym = subprocess.run(["nm", "-D", "--defined-only", LIBPATH], capture_output=True, text=True).stdout
    assert "cbgx_debug" not in sym and "_v1" not in sym and "edge_attention_kernel" not in sym
    xhdr = open(os.path.join(ROOT, "include", "cbgx_xcheck.h")).read()
    assert set(re.findall(r"\b(cbgx_[a-z0-9_]+)\s*\(", xhdr)) == {"cbgx_debug_set_edge_kernel"}
    xsym = subprocess.run(["nm", "-D", "--defined-only", XCHECK_LIBPATH], capture_output=True, text=True).stdout
    for name in declared | {"cbgx_debug_set_edge_kernel"}:
        assert f" {name}\n" in xsym, name


def test_abi_argument_errors_without_gpu():
    lib = _native.lib()
    assert lib.cbgx_packed_weights_floats(9, 13) >= 2666014   # at least every denoiser parameter
    assert lib.cbgx_workspace_bytes(425, 1) > 425 * (640 + 16 * 128) * 4
    one = ctypes.c_void_p(16)  # never dereferenced: argument checks come first
    assert lib.cbgx_knn_graph(one, one, 1, 10, 16, one, one, None) == -1
    assert b"k=32" in lib.cbgx_last_error()
    assert lib.cbgx_knn_graph(None, None, 0, 0, 32, None, None, None) == 0       # empty input is a no-op
    assert lib.cbgx_x2h_attention(one, 0, one, one, one, one, one, one, 10, one, one, 8, None) == -2
    assert b"workspace" in lib.cbgx_last_error()
    assert lib.cbgx_unitransformer_forward(one, 9, 13, None, one, one, one, one, 10, 1, one, one, one, one,
                                           1 << 30, None) == -1
    arr = (ctypes.c_void_p * 3)(16, 16, 16)
    assert lib.cbgx_pack_weights(arr, 3, 9, 13, one, None) == -1
    with pytest.raises(ValueError):
        _native.check(-1, "x")
    with pytest.raises(_native.NativeError):
        _native.check(-3, "x")


# ---- multi-GPU path: sharding + timing reduction over gloo, world_size 2 ------------------------------
def _worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), CBGX_RDZV_FILE=rdzv)
    os.environ.pop("MASTER_PORT", None)
    r, w, _ = sharding.init_process_group("gloo")
    mine = sharding.shard_indices(11, r, w)
    sharding.barrier()
    elapsed, units = sharding.reduce_max_sum(1.0 + r, len(mine))
    q.put((r, mine, elapsed, units))
    torch.distributed.destroy_process_group()


def test_pocket_sharding_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = _rdzv_file()
    procs = [ctx.Process(target=_worker, args=(r, 2, rdzv, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs: p.join(timeout=60)
    assert res[0][1] == [0, 2, 4, 6, 8, 10] and res[1][1] == [1, 3, 5, 7, 9]
    assert sorted(res[0][1] + res[1][1]) == list(range(11))      # a partition: no pocket lost or duplicated
    for _, _, elapsed, units in res:
        assert elapsed == 2.0 and units == 11.0                  # max over ranks, sum over ranks


def _forced_single_rank_worker(_unused, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", CBGX_DIST_FORCE="1")     # no rendezvous given: a private FileStore
    for k in ("MASTER_ADDR", "MASTER_PORT", "CBGX_RDZV_FILE"):
        os.environ.pop(k, None)
    r, w, _ = sharding.init_process_group("gloo")
    ok = torch.distributed.is_initialized() and torch.distributed.get_world_size() == 1
    sharding.barrier()
    q.put((r, w, ok, sharding.reduce_max_sum(1.5, 7)))
    torch.distributed.destroy_process_group()


def test_forced_single_rank_group_runs_the_collectives():
    """CBGX_DIST_FORCE=1: a one-rank group is created so that barrier / reductions run through the backend (on the GPU box this
    is how the RCCL path is exercised with one GPU, tests/test_gpu_bench.py); without it a single rank stays local"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single_rank_worker, args=(None, q))
    p.start()
    r, w, ok, red = q.get(timeout=120)
    p.join(timeout=60)
    assert (r, w, ok) == (0, 1, True) and red == (1.5, 7.0)


def test_diffsbdd_model_class(golden_dir):
    """registry entry, reference-compatible state dict, bit-identical gamma table, scheduler maths vs the oracle."""
    from oracle import diffsbdd as OD
    model = C.get_model(C.default_diffsbdd_config(8))
    with open(os.path.join(golden_dir, "state_dict_keys_diffsbdd.json")) as f:
        ref = json.load(f)
    sd = model.state_dict()
    sd5 = C.get_model(C.default_diffsbdd_config(8, num_diffusion_timesteps=5)).state_dict()   # the listing is of a T=5 model
    assert set(sd) == set(ref) and all(list(sd5[k].shape) == v for k, v in ref.items())
    z = np.load(os.path.join(golden_dir, "diffsbdd_gamma_T1000.npz"))
    assert np.array_equal(z["gamma"], sd["pos_scheduler.gamma.gamma"].numpy())
    model.load_state_dict(W.synthetic_state_dict_diffsbdd(8, 9), strict=True)
    g = torch.Generator().manual_seed(0)
    bl = torch.tensor([0, 0, 0, 1, 1]); br = torch.tensor([0, 0, 1, 1, 1, 1])
    zt = torch.randn(5, 3, generator=g); pocket = torch.randn(6, 3, generator=g)
    pred = torch.randn(5, 3, generator=g); eps = torch.randn(5, 3, generator=g)
    s, t = torch.full((2,), 400) / 1000, torch.full((2,), 401) / 1000
    sch = model.pos_scheduler
    with torch.no_grad():
        a, b = sch.sample_p_zs_given_zt(s, t, zt, pocket, bl, br, 2, pred, com=True, eps=eps)
    oa, ob = OD.sample_p_zs_given_zt(sd["pos_scheduler.gamma.gamma"], 1000, s, t, zt, pocket, bl, br, 2, pred, eps, True)
    assert torch.equal(a, oa) and torch.equal(b, ob)
    assert torch.allclose(OD.scatter_mean(a, bl, 2), torch.zeros(2, 3), atol=1e-6)   # COM-free after the draw
    # evaluation-mode loss terms of the scheduler (diffusion_scheduler.py:902-928) against the oracle on random inputs
    gam = sd["type_scheduler.gamma.gamma"]
    c0 = torch.nn.functional.one_hot(torch.tensor([1, 3, 0, 7, 2]), 8).float() / 4.0
    p1, e1, p0, e0 = (torch.randn(5, 8, generator=g) for _ in range(4))
    c_t0 = c0 + 0.01 * torch.randn(5, 8, generator=g)
    s2, t2 = torch.tensor([499, 0]) / 1000, torch.tensor([500, 1]) / 1000
    with torch.no_grad():
        got, _ = model.type_scheduler.get_score_loss_eval(p1, e1, s2, t2, None, bl, 2, p0, e0, c_lig_0=c0, c_lig_t0=c_t0)
        ref = OD.score_loss_eval(gam, 1000, p1, e1, s2, t2, bl, 2, p0, e0, c0=c0, c_t0=c_t0)
        assert torch.allclose(got, ref, rtol=1e-6, atol=0), (got, ref)
        x0c = zt - OD.scatter_mean(zt, bl, 2)[bl]
        got, _ = sch.get_score_loss_eval(pred, eps, s2, t2, None, bl, 2, p0[:, :3], e0[:, :3], x_lig_0=x0c)
        ref = OD.score_loss_eval(sd["pos_scheduler.gamma.gamma"], 1000, pred, eps, s2, t2, bl, 2, p0[:, :3], e0[:, :3], x0=x0c)
        assert torch.allclose(got, ref, rtol=1e-6, atol=0), (got, ref)


def test_diffbp_model_class(golden_dir):
    from oracle import diffbp as OD
    assert "diffbp" in C.registered_models()
    model = C.get_model(C.default_diffbp_config(13, num_diffusion_timesteps=5))
    with open(os.path.join(golden_dir, "state_dict_keys_diffbp.json")) as f:
        ref = json.load(f)
    sd = model.state_dict()
    assert set(sd) == set(ref) and all(list(sd[k].shape) == v for k, v in ref.items())
    wsd = W.synthetic_state_dict_diffbp(13, 9, num_timesteps=5)
    model.load_state_dict(wsd, strict=True)
    # element-wise sampler maths vs the oracle
    g = torch.Generator().manual_seed(1)
    bl = torch.tensor([0, 0, 0, 1, 1]); gen = torch.tensor([True, True, False, True, True])
    xt, pred, eps = (torch.randn(5, 3, generator=g) for _ in range(3))
    t = torch.tensor([3, 0])
    pos_tb = {k[len("pos_scheduler."):]: v for k, v in wsd.items() if k.startswith("pos_scheduler.")}
    with torch.no_grad():
        a = model.pos_scheduler.backward_remove_noise(pred, xt, t, bl, gen, type="score", noise=eps)
    assert torch.allclose(a, OD.pos_backward_score(pos_tb, pred, xt, t, bl, gen, eps), atol=1e-6)
    assert torch.equal(a[2], xt[2])
    logits = torch.randn(5, 13, generator=g); ct = torch.nn.functional.one_hot(torch.tensor([0, 4, 0, 0, 7]), 13).float()
    u = torch.tensor([0.1, 0.1, 0.1, 0.99, 0.0])
    c1, v1 = model.type_scheduler.backward_remove_noise(logits, ct, t, bl, gen, uniform=u)
    c2, v2 = OD.type_backward_mask(5, 13, logits, ct, t, bl, gen, u)
    assert torch.equal(v1, v2) and torch.equal(c1, c2)
    assert v1[1] == 4 and v1[2] == 0 and v1[4] == 7      # unmasked / non-generated atoms keep their type
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        model.com_head(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long), torch.zeros(4, 3), torch.zeros(4, 128),
                       torch.zeros(4, dtype=torch.bool), torch.tensor([0, 0, 1, 1], dtype=torch.bool),
                       torch.zeros(4, dtype=torch.long))


def test_training_schedulers_match_oracle():
    """host mirrors of forward_add_noise / get_loss (diffusion_scheduler.py:117-134,185-201,339-365) vs the oracle."""
    import torch
    import cbgbench_amd as C
    from oracle import training as TR, targetdiff as OT, weights as W
    m = C.get_model(C.default_targetdiff_config(13))
    sd = W.synthetic_state_dict(13, 9, seed=0)
    m.load_state_dict(sd, strict=True)
    pos_tb, typ_tb = OT.tables_from_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    n, B, Cn = 40, 4, 13
    bl = torch.sort(torch.randint(0, B, (n,), generator=g)).values
    bl[0], bl[-1] = 0, B - 1
    gen = torch.rand(n, generator=g) > 0.2
    x0, eps = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
    v0, u = torch.randint(0, Cn, (n,), generator=g), torch.rand(n, Cn, generator=g)
    t = torch.tensor([0, 17, 500, 999])
    xt, _ = m.pos_scheduler.forward_add_noise(x0, t, bl, gen, noise=eps)
    assert torch.equal(xt, TR.pos_forward_add_noise(pos_tb, x0, t, bl, gen, eps))
    ct, vt = m.type_scheduler.forward_add_noise(v0, t, bl, gen, uniform=u)
    ct_o, vt_o = TR.type_forward_add_noise(typ_tb, Cn, v0, t, bl, gen, u)
    assert torch.equal(vt, vt_o) and torch.equal(ct, ct_o)
    x_pred, logits = torch.randn(n, 3, generator=g), torch.randn(n, Cn, generator=g)
    nl = int(bl[gen].max()) + 1
    lp, _ = m.pos_scheduler.get_loss(x_pred, x0, xt, t, gen, bl, type="denoise")
    assert torch.equal(lp, TR.pos_loss(x_pred, x0, gen, bl, nl))
    la, _ = m.type_scheduler.get_loss(logits, v0, vt, t, gen, bl)
    assert torch.equal(la, TR.type_loss(typ_tb, Cn, logits, v0, vt, t, gen, bl, nl))
    assert torch.equal(m.sample_time(5, "cpu", draws=torch.tensor([3, 10, 999])),
                       TR.sample_time_symmetric(5, 1000, [3, 10, 999]))


# ---- data-parallel training step over gloo, world_size 2 (the GPU run uses the same code over RCCL) ---------
class _ToyModel(torch.nn.Module):
    """stands in for a model class: forward(batch) -> (loss_dict, results), like targetdiff.py:40-60"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.lin = torch.nn.Linear(5, 3)
        self.frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)

    def forward(self, batch):
        out = self.lin(batch["x"])
        return {"pos": ((out - batch["y"]) ** 2).mean(), "atom": out.abs().mean()}, {}


def _toy_batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return {"x": torch.randn(8, 5, generator=g), "y": torch.randn(8, 3, generator=g)}


def _train_worker(rank, world, rdzv, q):
    from cbgbench_amd import train as TRN
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), CBGX_RDZV_FILE=rdzv)
    os.environ.pop("MASTER_PORT", None)
    sharding.init_process_group("gloo")
    model = _ToyModel()
    with torch.no_grad():
        model.lin.weight.add_(float(rank))          # ranks start different; broadcast must fix it
    TRN.broadcast_parameters(model)
    fg = TRN.FlatGradients(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    w = {"pos": 1.0, "atom": 100.0}
    for _ in range(3):
        loss, ld, gn, t_ar = TRN.train_step(model, _toy_batch(rank), opt, fg, w, max_grad_norm=8.0)
    val = TRN.validate(model, [_toy_batch(rank)], w)
    q.put((rank, model.lin.weight.detach().numpy().tolist(), fg.flat.numpy().tolist(), float(gn), val))
    torch.distributed.destroy_process_group()


def test_data_parallel_train_step_gloo_world2():
    import torch.multiprocessing as mp
    from cbgbench_amd import train as TRN
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = _rdzv_file()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, rdzv, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs: p.join(timeout=60)
    # both ranks hold identical weights and identical (averaged) gradients after every step
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
    # and they equal a single process stepping on the mean of the two per-rank losses
    model = _ToyModel()
    fg = TRN.FlatGradients(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    w = {"pos": 1.0, "atom": 100.0}
    for _ in range(3):
        fg.zero()
        loss = sum(TRN.sum_weighted_losses(model(_toy_batch(r))[0], w) for r in range(2)) / 2
        loss.backward()
        torch.nn.utils.clip_grad_norm_(fg.params, 8.0)
        opt.step()
    assert torch.allclose(model.lin.weight, torch.tensor(res[0][1]), atol=1e-6)
    assert model.lin.weight.grad.data_ptr() == fg.flat.data_ptr()     # gradients live in the one flat buffer


def test_product_side_synthetic_weights_equal_the_oracle_generator():
    """bench.py / smoke fill the GPU model with cbgbench_amd.synthetic_weights (no oracle import on the product side);
    the oracle's own generator must give the same tensors, otherwise the cpu_baseline would time a different model"""
    from cbgbench_amd import synthetic_weights
    model = C.get_model(C.default_targetdiff_config(13))
    synthetic_weights.fill_(model, seed=0)
    sd = W.synthetic_state_dict(13, 9, seed=0)
    msd = model.state_dict()
    assert set(msd) == set(sd)
    for k, v in sd.items():
        assert torch.equal(msd[k], v), k


def test_flat_buffer_clip_equals_clip_grad_norm():
    """FlatGradients.clip_norm_ (one reduction over the flat buffer) is torch.nn.utils.clip_grad_norm_ over the per-tensor views"""
    from cbgbench_amd import train as TRN
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    fg = TRN.FlatGradients(net)
    for max_norm in (0.5, 1e3):          # clipping / not clipping
        fg.flat.copy_(torch.randn_like(fg.flat) * 3.0)
        ps = [torch.nn.Parameter(torch.zeros_like(p)) for p in fg.params]
        for q, p in zip(ps, fg.params):
            q.grad = p.grad.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_(ps, max_norm)
        got_norm = fg.clip_norm_(max_norm)
        assert torch.allclose(got_norm, ref_norm, rtol=1e-6)
        for q, p in zip(ps, fg.params):
            assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-8)
    # zero() restores the views after optimizer-style set_to_none and leaves the buffer zero
    net.zero_grad(set_to_none=True)
    fg.zero()
    assert all(p.grad is v for p, v in zip(fg.params, fg.views)) and float(fg.flat.abs().sum()) == 0.0


def test_flat_adam_is_adam_with_interchangeable_checkpoints():
    """train.FlatAdam (parameters and moments as views of flat buffers, one update) = torch.optim.Adam step for step, and their
    state dicts load into each other mid-run"""
    from cbgbench_amd import train as TRN
    torch.manual_seed(1)

    def make():
        torch.manual_seed(7)
        return torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 4))

    def grads(net, k):
        g = torch.Generator().manual_seed(100 + k)
        for p in net.parameters():
            p.grad = torch.randn(p.shape, generator=g) if p.grad is None else p.grad.copy_(torch.randn(p.shape, generator=g))

    a, b = make(), make()
    kw = dict(lr=3e-3, betas=(0.95, 0.999), eps=1e-8, weight_decay=0.0)
    oa, ob = torch.optim.Adam(a.parameters(), **kw), TRN.FlatAdam(b.parameters(), **kw)
    fg = TRN.FlatGradients(b)                      # gradients of b as views of one buffer: FlatAdam's fast path
    for k in range(4):
        grads(a, k); grads(b, k)
        oa.step(); ob.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=0, atol=1e-7)
    # checkpoints both ways, then two more steps
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa.keys() == sb.keys() and sa["state"].keys() == sb["state"].keys()
    assert set(sb["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sb["state"][0]["step"]) == 4.0
    a2, b2 = make(), make()
    a2.load_state_dict(b.state_dict()); b2.load_state_dict(a.state_dict())
    oa2, ob2 = torch.optim.Adam(a2.parameters(), **kw), TRN.FlatAdam(b2.parameters(), **kw)
    oa2.load_state_dict(sb); ob2.load_state_dict(sa)
    for k in range(4, 6):
        for net, o in ((a, oa), (b, ob), (a2, oa2), (b2, ob2)):
            grads(net, k); o.step()
    for p, q, r, t in zip(a.parameters(), b.parameters(), a2.parameters(), b2.parameters()):
        assert torch.allclose(p, q, atol=1e-7) and torch.allclose(p, r, atol=1e-7) and torch.allclose(p, t, atol=1e-7)
    assert fg.flat.numel() == sum(p.numel() for p in b.parameters())
    # without FlatGradients (gradients are ordinary per-parameter tensors): same update through the gather path
    c, d = make(), make()
    oc, od = torch.optim.Adam(c.parameters(), **kw), TRN.FlatAdam(d.parameters(), **kw)
    for k in range(3):
        grads(c, k); grads(d, k); oc.step(); od.step()
    for p, q in zip(c.parameters(), d.parameters()):
        assert torch.allclose(p, q, rtol=0, atol=1e-7)


def test_flat_adam_marks_parameters_modified_and_skips_missing_gradients():
    """(a) every FlatAdam.step() bumps the parameters' version counters -- UniTransformer.packed_weights keys its cache on them, so
    without the bump the native denoiser would keep training on the weights of step 0; (b) a parameter whose
    .grad is None is skipped like torch.optim.Adam skips it; (c) a checkpoint without state for some parameter zeroes its moments."""
    from cbgbench_amd import train as TRN

    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 2))

    def grads(net, k, skip=None):
        g = torch.Generator().manual_seed(50 + k)
        for i, p in enumerate(net.parameters()):
            v = torch.randn(p.shape, generator=g)
            p.grad = None if i == skip else v

    a, b = make(), make()
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0)
    oa, ob = torch.optim.Adam(a.parameters(), **kw), TRN.FlatAdam(b.parameters(), **kw)
    before = [p._version for p in b.parameters()]
    ptrs = [p.data_ptr() for p in b.parameters()]
    grads(a, 0); grads(b, 0); oa.step(); ob.step()
    assert all(p._version > v for p, v in zip(b.parameters(), before)), "step() must bump the version counters"
    assert [p.data_ptr() for p in b.parameters()] == ptrs
    # parameter 1 (first bias) without a gradient for two steps
    for k in (1, 2):
        grads(a, k, skip=1); grads(b, k, skip=1); oa.step(); ob.step()
    grads(a, 3); grads(b, 3); oa.step(); ob.step()
    pa, pb = list(a.parameters()), list(b.parameters())
    for i, (p, q) in enumerate(zip(pa, pb)):
        if i != 1:      # stock Adam's private step counter of the skipped parameter lags by two: its bias correction differs
            assert torch.allclose(p, q, rtol=0, atol=1e-7), i
    sa = oa.state_dict()
    assert torch.allclose(sa["state"][1]["exp_avg"], ob.state_dict()["state"][1]["exp_avg"], atol=1e-7)   # moments were skipped too
    # (c) a checkpoint that lacks the state of parameter 2 -> its moments start from zero
    sb = ob.state_dict()
    del sb["state"][2]
    c = make()
    oc = TRN.FlatAdam(c.parameters(), **kw)
    oc._m.fill_(9.0)
    oc.load_state_dict(sb)
    m2 = oc._m.split(oc._sizes)[2]
    assert float(m2.abs().sum()) == 0.0 and float(oc._step) == 4.0
    # differing step counters are refused, not silently merged
    sa_bad = oa.state_dict()
    with pytest.raises(ValueError):
        TRN.FlatAdam(make().parameters(), **kw).load_state_dict(sa_bad)


def test_sync_free_host_forms_equal_the_reference_forms():
    """The training step's host code avoids device->host synchronisations; each replacement is the reference
    expression's value: masked_graph_mean == scatter_mean(v[mask], idx[mask]).mean(), ligand rows from the inverse permutation ==
    nonzero(), CSR offsets without bincount."""
    from cbgbench_amd import targetdiff as TD
    from cbgbench_amd.unitransformer import graph_ptr_from_batch
    g = torch.Generator().manual_seed(5)
    for trailing_empty in (False, True):
        B = 7
        idx = torch.sort(torch.randint(0, B - (2 if trailing_empty else 0), (60,), generator=g)).values
        v = torch.randn(60, generator=g, dtype=torch.float64)
        mask = torch.rand(60, generator=g) < 0.6
        mask[0] = True
        ref = TD.scatter_mean(v[mask], idx[mask]).mean()
        got = TD.masked_graph_mean(v, idx, mask, B)
        assert torch.allclose(got, ref, rtol=1e-12, atol=0)
        # a NaN outside the mask must not leak in
        v2 = v.clone(); v2[~mask] = float("nan")
        assert torch.allclose(TD.masked_graph_mean(v2, idx, mask, B), ref, rtol=1e-12, atol=0)
    # composition plan: ligand rows in ligand order
    br = torch.sort(torch.randint(0, 4, (30,), generator=g)).values
    bl = torch.sort(torch.randint(0, 4, (9,), generator=g)).values
    sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = TD.TargetDiff.compose_plan(bl, br, 4)
    assert torch.equal(lig_rows, torch.nonzero(lig_flag).flatten()[torch.argsort(sort_idx[lig_flag])])
    assert torch.equal(sort_idx[lig_rows], torch.arange(30, 39))
    counts = torch.bincount(batch_idx, minlength=4)
    assert torch.equal(graph_ptr, torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32))
    assert torch.equal(graph_ptr_from_batch(batch_idx, 4), graph_ptr) and torch.equal(graph_ptr_from_batch(batch_idx), graph_ptr)


def test_ordered_parameter_list_is_cached_and_revalidated():
    """UniTransformer._ordered_params(): the list cbgx_pack_weights / cbgx_unitransformer_backward take is cached (walking the module
    tree costs ~1 ms, three times per training step) and re-validated against the owning modules' parameter slots."""
    import cbgbench_amd as C
    m = C.get_model(C.default_targetdiff_config(13))
    d = m.denoiser
    a = d._ordered_params()
    assert len(a) == 6 + 36 * d.num_layers + 4 and d._ordered_params() is a           # cached
    sd = dict(d.named_parameters())
    assert a[0] is sd["dist_emb.1.net.0.weight"] and a[-1] is sd["classifier.2.bias"] and a[6] is sd["blocks.0.x2h_layers.0.hk_func.net.0.weight"]
    # a replaced Parameter is seen (identity check against the modules' slots), as is a conversion through _apply
    lin = d.blocks[3].h2x_layers[0].xv_func.net[3]
    lin.weight = torch.nn.Parameter(torch.zeros_like(lin.weight))
    b = d._ordered_params()
    assert b is not a and any(p is lin.weight for p in b) and len(b) == len(a)
    m.double()
    assert d._ordered_params()[0].dtype == torch.float64
    # load_state_dict copies in place: same objects, same list
    c = d._ordered_params()
    m.load_state_dict(m.state_dict())
    assert d._ordered_params() is c


def test_version_of_an_inference_tensor_never_matches():
    """caches keyed on tensor._version must not crash under torch.inference_mode() (inference tensors keep no counter):
    _native.version hands out a value that never compares equal, so such tensors always take the uncached route"""
    t = torch.zeros(3)
    assert _native.version(t) == _native.version(t) == 0
    t.add_(1)
    assert _native.version(t) == 1
    with torch.inference_mode():
        f = torch.zeros(3, dtype=torch.bool)
        assert f.is_inference()
        assert not (_native.version(f) == _native.version(f))
        assert _native.version(f) != _native.version(f)
        key = (1, _native.version(f))
        assert key != (1, _native.version(f))

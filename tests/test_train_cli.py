"""Training driver (cbgbench_amd/train_cli.py, the role of the reference's train.py:99-273): rank-aware loader, collate,
checkpoint / resume, and the whole loop on a stub model over gloo with two ranks; the same loop on the real TargetDiff on a GPU."""
import os

import numpy as np
import pytest
import torch

import cbgbench_amd as C
from cbgbench_amd import registry, sharding, train_cli
from cbgbench_amd.config import Config


def _rdzv_file():
    """the FileStore path the ranks of one test job meet on (cbgbench_amd.launch's rendezvous: no port to collide on)"""
    from cbgbench_amd import launch
    return os.path.join(launch.rendezvous_dir(), "store")


def test_sharded_loader_partitions_every_epoch():
    n, world, bs = 11, 3, 2
    loaders = [train_cli.ShardedLoader(n, bs, r, world, seed=5) for r in range(world)]
    for e in (0, 1):
        per_rank = [ld.epoch(e) for ld in loaders]
        assert len({len(b) for b in per_rank}) == 1                      # same number of steps on every rank
        flat = [i for b in per_rank for ids in b for i in ids]
        assert set(flat) == set(range(n)) and len(flat) == 12            # a cover; the tail wraps around (12 = 3 * 4)
        assert per_rank == [ld.epoch(e) for ld in loaders]               # deterministic
    assert loaders[0].epoch(0) != loaders[0].epoch(1)                    # reshuffled per epoch
    plain = train_cli.ShardedLoader(7, 4, 1, 2, shuffle=False).epoch(0)
    assert plain == [[1, 3, 5, 0]]                                       # arange split r::W, padded by wrapping
    it = iter(train_cli.ShardedLoader(3, 2, 0, 1, seed=0))
    seen = [next(it) for _ in range(4)]                                  # infinite: two epochs of two batches
    assert sorted(seen[0] + seen[1]) == [0, 1, 2] and sorted(seen[2] + seen[3]) == [0, 1, 2]
    tiny = [train_cli.ShardedLoader(1, 2, r, 3, shuffle=False).epoch(0) for r in range(3)]
    assert tiny == [[[0]], [[0]], [[0]]]                                  # fewer items than ranks: everyone still steps
    with pytest.raises(ValueError):
        train_cli.ShardedLoader(0, 2)


def test_complex_set_collate_matches_manual_concat():
    cx = train_cli.synthetic_complexes(5, seed=3, num_classes=13, n_rec_range=(20, 40), n_lig_range=(3, 9))
    cs = train_cli.ComplexSet(cx)
    ids = [3, 0, 4]
    b = cs.collate(ids)
    off_r = off_l = 0
    for g, i in enumerate(ids):
        c = cx[i]
        ctr = c["protein_pos"].mean(0)
        nr, nl = c["protein_pos"].shape[0], c["ligand_pos"].shape[0]
        assert torch.allclose(b["protein_pos"][off_r:off_r + nr], torch.from_numpy(c["protein_pos"] - ctr), atol=1e-6)
        assert torch.allclose(b["ligand_pos"][off_l:off_l + nl], torch.from_numpy(c["ligand_pos"] - ctr), atol=1e-6)
        assert torch.equal(b["ligand_atom_type"][off_l:off_l + nl], torch.from_numpy(c["ligand_atom_type"]))
        assert torch.equal(b["protein_aa_type"][off_r:off_r + nr], torch.from_numpy(c["protein_aa_type"]))
        assert bool((b["protein_element_batch"][off_r:off_r + nr] == g).all())
        assert bool((b["ligand_element_batch"][off_l:off_l + nl] == g).all())
        off_r, off_l = off_r + nr, off_l + nl
    assert b["protein_pos"].shape[0] == off_r and b["ligand_pos"].shape[0] == off_l
    assert not b["protein_lig_flag"].any() and b["ligand_lig_flag"].all() and "ligand_gen_flag" not in b
    for c in cx:
        c["ligand_gen_flag"] = np.arange(c["ligand_pos"].shape[0]) >= 2
    b = train_cli.ComplexSet(cx).collate([1])
    assert b["ligand_gen_flag"].tolist() == (np.arange(cx[1]["ligand_pos"].shape[0]) >= 2).tolist()


class _Stub(torch.nn.Module):
    """a model class in the registry's sense: __init__(cfg), forward(batch) -> (loss_dict, results)"""

    def __init__(self, cfg):
        super().__init__()
        torch.manual_seed(int(os.environ.get("RANK", 0)))        # ranks initialise differently; the driver must broadcast
        self.lin = torch.nn.Linear(3, 3)
        self.register_buffer("table", torch.arange(4.0))

    def forward(self, batch):
        out = self.lin(batch["ligand_pos"])
        return {"pos": (out ** 2).mean(), "atom": (out - 1.0).abs().mean()}, {}


def _stub_config(max_iters, scheduler="plateau"):
    sch = dict(type="plateau", factor=0.5, patience=0, min_lr=1e-6) if scheduler == "plateau" else dict(type="exp", gamma=0.9)
    return Config(model=dict(type="stub_cpu"),
                  train=dict(loss_weights=dict(pos=1.0, atom=2.0), max_iters=max_iters, report_freq=2, batch_size=2, seed=7,
                             max_grad_norm=8.0, optimizer=dict(type="adam", lr=1e-2, weight_decay=0.0, beta1=0.9, beta2=0.999),
                             scheduler=sch),
                  eval=dict(val_freq=3))


def _sets():
    cx = train_cli.synthetic_complexes(10, seed=1, num_classes=13, n_rec_range=(8, 12), n_lig_range=(3, 6))
    return train_cli.ComplexSet(cx[:8]), train_cli.ComplexSet(cx[8:])


def test_loop_checkpoints_and_resume_single_process(tmp_path):
    registry.register_model("stub_cpu")(_Stub)
    tr, va = _sets()
    lines = []
    out = train_cli.run(_stub_config(6), "stub", tr, va, torch.device("cpu"), str(tmp_path), log=lines.append)
    assert [it for it, _ in out["history"]] == [3, 6]
    ck3 = os.path.join(out["ckpt_dir"], "3.pt")
    assert os.path.exists(ck3) and not os.path.exists(ck3 + ".tmp")
    ck = torch.load(ck3, weights_only=False)
    assert set(ck) == {"config", "model", "optimizer", "scheduler", "iteration", "avg_val_loss"} and ck["iteration"] == 3
    assert any(l.startswith("[train] iter 00002") for l in lines) and any(l.startswith("[validate] iter 00003") for l in lines)
    # resume: same weights / optimizer state, continues AT the saved iteration (train.py:175 does not add one)
    out2 = train_cli.run(_stub_config(4), "stub2", tr, va, torch.device("cpu"), str(tmp_path), resume=ck3, log=lines.append)
    assert any("[resume]" in l and "iteration 3" in l for l in lines)
    m = _Stub(None)
    it_first, missing, unexpected = train_cli.load_checkpoint(ck3, m, finetune=True)
    assert it_first == 1 and not missing and not unexpected
    assert torch.equal(m.lin.weight, ck["model"]["lin.weight"])
    st = out2["optimizer"].state_dict()["state"]
    assert int(st[0]["step"]) == 3 + 2            # 3 restored steps + iterations 3 and 4


def _worker(rank, world, rdzv, logdir, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), CBGX_RDZV_FILE=rdzv)
    os.environ.pop("MASTER_PORT", None)
    sharding.init_process_group("gloo")
    registry.register_model("stub_cpu")(_Stub)
    tr, va = _sets()
    out = train_cli.run(_stub_config(6, scheduler="exp"), "ddp", tr, va, torch.device("cpu"), logdir, log=lambda s: None)
    q.put((rank, out["model"].lin.weight.detach().numpy().tolist(), out["history"], out["optimizer"].param_groups[0]["lr"]))
    torch.distributed.destroy_process_group()


def test_loop_data_parallel_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = _rdzv_file()
    procs = [ctx.Process(target=_worker, args=(r, 2, rdzv, str(tmp_path), q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs: p.join(timeout=60)
    assert res[0][1] == res[1][1]                        # replicas stay identical (broadcast + averaged gradients)
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]   # same validation losses -> same scheduler decisions
    files = sorted(os.listdir(os.path.join(str(tmp_path), "ddp", "checkpoints")))
    assert files and all(f.endswith(".pt") for f in files)     # written once (rank 0), complete


@pytest.mark.gpu
def test_train_cli_end_to_end_on_gpu(tmp_path):
    """yaml config -> TargetDiff -> 4 iterations with validation + checkpoint -> resume, through libcbgx"""
    cfg = tmp_path / "targetdiff_train.yml"
    cfg.write_text("""
model:
  type: targetdiff
  encoder: {type: unitransformer, node_feat_dim: 128, n_heads: 16, num_layers: 3}
  generator:
    pos_schedule: {type: sigmoid, beta_start: 1.0e-7, beta_end: 2.0e-3}
    atom_schedule: {type: cosine, cosine_s: 0.01}
    num_diffusion_timesteps: 1000
    time_sampler: symmetric
  embedder: {emb_dim: 128, atom: {type: linear}, residue: {type: linear}}
  eval_interval: 2
data:
  train:
    transform:
      - {type: featurize_ligand_fa, mode: add_aromatic}
train:
  loss_weights: {pos: 1.0, atom: 100.0}
  max_iters: 4
  report_freq: 1
  batch_size: 4
  seed: 2022
  max_grad_norm: 8.0
  optimizer: {type: adam, lr: 5.e-4, weight_decay: 0.0, beta1: 0.95, beta2: 0.999}
  scheduler: {type: plateau, factor: 0.6, patience: 10, min_lr: 1.e-6}
eval:
  val_freq: 2
  metrics:
    - {name: auroc, true_key: v0, pred_key: c_pred, mask_key: mask_gen}
""")
    logdir = str(tmp_path / "logs")
    lines = []
    orig_run = train_cli.run
    train_cli.run = lambda *a, **kw: orig_run(*a, **{**kw, "log": lines.append})
    try:
        assert train_cli.main(["--config", str(cfg), "--logdir", logdir, "--synthetic", "12"]) == 0
    finally:
        train_cli.run = orig_run
    # the config's metric (type-prediction AUROC, repo/utils/evaluate.py:35-73) is part of every validation report
    val = [l for l in lines if l.startswith("[validate]")]
    assert len(val) == 2 and all("auroc_atom" in l for l in val)
    a = float(val[0].split("auroc_atom")[1].split("|")[0])
    assert 0.0 <= a <= 1.0
    ck = os.path.join(logdir, "targetdiff_train", "checkpoints", "2.pt")
    assert os.path.exists(ck)
    saved = torch.load(ck, weights_only=False)
    assert saved["iteration"] == 2 and np.isfinite(saved["avg_val_loss"])
    assert train_cli.main(["--config", str(cfg), "--logdir", logdir, "--tag", "resumed", "--synthetic", "12",
                           "--resume", ck, "--max_iters", "3"]) == 0


def test_checkpoint_config_is_interchangeable_with_the_reference(tmp_path):
    """(1) a checkpoint written here carries its config as plain nested dicts (or an EasyDict when easydict is installed): the
    pickle names no cbgbench_amd class, so the reference's scripts can load it; (2) a checkpoint written the reference's
    way -- config pickled as easydict.EasyDict (train.py:266-273) -- loads here without easydict installed."""
    import pickle
    import pickletools
    import sys
    import types
    from cbgbench_amd.config import load_checkpoint_file
    cfg = Config({"model": {"type": "targetdiff", "encoder": {"num_layers": 9}}, "train": {"seed": 2022, "list": [1, {"a": 2}]}})
    model = torch.nn.Linear(2, 2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    path = str(tmp_path / "ck" / "10.pt")
    train_cli.save_checkpoint(path, cfg, model, opt, None, 10, 0.5)
    import zipfile
    with zipfile.ZipFile(path) as z:
        pkl = [n for n in z.namelist() if n.endswith("data.pkl")][0]
        ops = [(op.name, arg) for op, arg, _ in pickletools.genops(z.read(pkl))]
    named = " ".join(str(a) for n, a in ops if n in ("GLOBAL", "STACK_GLOBAL", "SHORT_BINUNICODE", "BINUNICODE") and a)
    assert "cbgbench_amd" not in named
    ck = load_checkpoint_file(path)
    assert isinstance(ck["config"], Config) and ck["config"].model.encoder.num_layers == 9 and ck["config"].train.list[1].a == 2
    assert ck["scheduler"] == {} and ck["iteration"] == 10

    # the reference's way: a dict subclass `easydict.EasyDict` with attribute access, pickled by reference
    had = sys.modules.pop("easydict", None)
    fake = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None):
            super().__init__()
            for k, v in (d or {}).items():
                setattr(self, k, EasyDict(v) if isinstance(v, dict) else v)

        def __setattr__(self, k, v):
            super().__setattr__(k, v)
            super().__setitem__(k, v)
    EasyDict.__module__, EasyDict.__qualname__ = "easydict", "EasyDict"
    fake.EasyDict = EasyDict
    sys.modules["easydict"] = fake
    try:
        ref_path = str(tmp_path / "ref.pt")
        torch.save({"config": EasyDict({"model": {"type": "targetdiff", "generator": {"num_diffusion_timesteps": 1000}}}),
                    "model": model.state_dict(), "optimizer": opt.state_dict(), "scheduler": {}, "iteration": 3,
                    "avg_val_loss": 1.0}, ref_path)
    finally:
        del sys.modules["easydict"]
        if had is not None:
            sys.modules["easydict"] = had
    try:
        import easydict  # noqa: F401
        pytest.skip("easydict is installed here: the stand-in path is not exercised")
    except ImportError:
        pass
    ck = load_checkpoint_file(ref_path)
    assert isinstance(ck["config"], Config) and ck["config"].model.generator.num_diffusion_timesteps == 1000
    assert "easydict" not in sys.modules

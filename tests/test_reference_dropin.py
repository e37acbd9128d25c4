"""The literal drop-in of INTEGRATION.md section 2, executed: the reference's OWN model classes (repo/models/diffusion/
targetdiff.py:14-184, diffbp.py, diffsbdd.py -- unmodified, imported through oracle/ref_shim.py) built with the encoder factory
(repo/modules/e3nn/__init__.py:5-18) pointed at the libcbgx-backed UniTransformer.  The reference tree is absent on the GPU box, so
this is a build-container test and goes as far as a CPU container can: construction through the reference's factory call, strict
``load_state_dict`` in both directions (sample.py:153-156), the keyword call of targetdiff.py:158-162 reaching our ``forward`` from
inside the reference's own ``sample()`` loop with the tensors the reference composed, the loud refusal to compute on the CPU, and the
reference's ValueError texts for the encoder options libcbgx does not implement."""
import inspect
import os

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "repo")), reason="needs the reference tree at /root/reference")


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_shim
    return ref_shim.load_reference()


@pytest.fixture()
def patched_factory(ref, monkeypatch):
    """`if cfg.type == 'unitransformer': return CbgxUniTransformer(cfg)` in get_e3_gnn.  The model modules
    bound the factory by `from repo.modules.e3nn import get_e3_gnn`, so the patched function is installed under each of those names."""
    import repo.modules.e3nn as E
    import repo.models.diffusion.targetdiff as RT
    import repo.models.diffusion.diffbp as RB
    import repo.models.diffusion.diffsbdd as RS
    from cbgbench_amd.unitransformer import UniTransformer as CbgxUniTransformer
    original = E.get_e3_gnn
    made = []

    def get_e3_gnn(cfg, num_classes=None, num_edge_classes=None):
        if num_classes is not None:
            cfg.num_classes = num_classes
        if num_edge_classes is not None:
            cfg.num_edge_classes = num_edge_classes
        if cfg.type == "unitransformer":
            made.append(CbgxUniTransformer(cfg))            # was: UniTransformer(cfg)
            return made[-1]
        return original(cfg, num_classes, num_edge_classes)

    for mod in (E, RT, RB, RS):
        if hasattr(mod, "get_e3_gnn"):
            monkeypatch.setattr(mod, "get_e3_gnn", get_e3_gnn)
    made.append(original)          # made[0]: the reference's own factory, for the tests that build the unpatched model next to it
    return made


def _cfg(name, C=13):
    from oracle import ref_shim
    import cbgbench_amd as CB
    if name == "targetdiff":
        return ref_shim.targetdiff_config(C, 9)
    ours = {"diffbp": CB.default_diffbp_config, "diffsbdd": CB.default_diffsbdd_config}[name](C)
    return ref_shim.AttrDict(_plain(ours))


def _plain(c):
    if isinstance(c, dict):
        return {k: _plain(v) for k, v in c.items()}
    if isinstance(c, (list, tuple)):
        return [_plain(v) for v in c]
    return c


@pytest.mark.parametrize("name", ["targetdiff", "diffbp", "diffsbdd"])
def test_reference_model_class_builds_with_the_libcbgx_encoder_and_loads_strictly_both_ways(ref, patched_factory, name):
    from cbgbench_amd.unitransformer import UniTransformer as CbgxUniTransformer
    C = 8 if name == "diffsbdd" else 13
    torch.manual_seed(0)
    swapped = ref.get_model(_cfg(name, C))                   # the reference's class, our encoder inside
    assert type(swapped).__module__.startswith("repo.models.diffusion")
    assert isinstance(swapped.denoiser, CbgxUniTransformer) and patched_factory[1] is swapped.denoiser
    # the unpatched reference model of the same config
    import importlib
    pure_mod = importlib.import_module(type(swapped).__module__)
    torch.manual_seed(1)
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr(pure_mod, "get_e3_gnn", patched_factory[0])
        pure = ref.get_model(_cfg(name, C))
    assert not isinstance(pure.denoiser, CbgxUniTransformer)
    sd_pure, sd_swapped = pure.state_dict(), swapped.state_dict()
    assert list(sd_pure.keys()) == list(sd_swapped.keys())                    # same names, same ORDER (checkpoints are ordered dicts)
    for k in sd_pure:
        assert sd_pure[k].shape == sd_swapped[k].shape and sd_pure[k].dtype == sd_swapped[k].dtype, k
    # sample.py:153-156: model.load_state_dict(ckpt['model']) is strict -- a reference checkpoint into the swapped model ...
    res = swapped.load_state_dict(sd_pure, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in swapped.state_dict().items():
        assert torch.equal(v, sd_pure[k]), k
    # ... and a checkpoint written with the swapped model back into the pure reference
    torch.manual_seed(2)
    for p in swapped.parameters():
        if p.requires_grad:
            p.data.add_(torch.randn_like(p) * 1e-3)
    res = pure.load_state_dict(swapped.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sum(p.numel() for p in pure.parameters()) == sum(p.numel() for p in swapped.parameters())
    # buffers the reference registers (GaussianSmearing offsets) carry the same values
    for (kb, b), (kb2, b2) in zip(pure.named_buffers(), swapped.named_buffers()):
        assert kb == kb2 and torch.equal(b, b2), kb


def test_forward_signature_is_the_reference_keyword_call():
    """targetdiff.py:158-162: self.denoiser(batch_idx=batch_idx, **context_composed), keys {'x','h','gen_flag','lig_flag'}; the
    reference's own forward is (x, h, batch_idx, lig_flag, gen_flag) positionally (unitransformer.py:102)"""
    from cbgbench_amd.unitransformer import UniTransformer as CbgxUniTransformer
    import repo.modules.e3nn.unitransformer as U
    ours = inspect.signature(CbgxUniTransformer.forward)
    theirs = inspect.signature(U.UniTransformer.forward)
    lead = list(theirs.parameters)                             # self, x, h, batch_idx, lig_flag, gen_flag
    assert list(ours.parameters)[:len(lead)] == lead
    for extra in list(ours.parameters)[len(lead):]:            # everything libcbgx adds is optional
        assert ours.parameters[extra].default is not inspect.Parameter.empty, extra
    ours.bind(None, batch_idx=0, x=1, h=2, gen_flag=3, lig_flag=4)


def test_reference_sample_loop_reaches_the_libcbgx_forward_with_the_composed_tensors(ref, patched_factory, monkeypatch):
    """the reference's TargetDiff.sample (targetdiff.py:127-184), unmodified, with the swapped encoder: it composes the context and
    makes the keyword call; on this GPU-less container our forward must refuse loudly (no CPU fallback), and what it was handed must
    be what the oracle composes for the same batch"""
    from oracle import make_golden as MG, targetdiff as OT, weights as W
    import torch.nn.functional as F
    model = ref.get_model(_cfg("targetdiff")).eval()
    model.load_state_dict(W.synthetic_state_dict(13, 9, seed=0), strict=True)
    batch = MG.small_batch([(40, 6), (33, 8)], seed=5)
    seen = {}
    den = model.denoiser
    real_forward = type(den).forward

    def spy(self, *a, **kw):
        seen["args"], seen["kw"] = a, dict(kw)
        return real_forward(self, *a, **kw)

    monkeypatch.setattr(type(den), "forward", spy)
    with pytest.raises(RuntimeError, match="no CPU fallback exists"):
        model.sample(dict(batch))
    assert not seen["args"] and sorted(seen["kw"]) == ["batch_idx", "gen_flag", "h", "lig_flag", "x"]
    kw = seen["kw"]
    n = batch["protein_pos"].shape[0] + batch["ligand_pos"].shape[0]
    assert kw["x"].shape == (n, 3) and kw["h"].shape == (n, 128) and kw["x"].dtype == kw["h"].dtype == torch.float32
    assert kw["batch_idx"].dtype == torch.int64 and kw["lig_flag"].dtype == torch.bool and kw["gen_flag"].dtype == torch.bool
    assert bool((kw["batch_idx"][1:] >= kw["batch_idx"][:-1]).all())          # sorted by graph: what graph_ptr_from_batch needs
    # the same rows as the oracle's compose of the same batch (protein atoms first, then ligand atoms, per graph: common.py:189-214)
    sd = W.synthetic_state_dict(13, 9, seed=0)
    c0 = F.one_hot(batch["ligand_atom_type"], 13).float()
    aa = F.one_hot(batch["protein_aa_type"], 20).float()
    h_lig, h_rec = OT.context_embed(sd, c0, batch["protein_atom_feature"], aa)
    sort_idx, bidx = OT.compose(batch["ligand_element_batch"], batch["protein_element_batch"])
    n_rec, n_lig = batch["protein_pos"].shape[0], batch["ligand_pos"].shape[0]
    lig = torch.cat([torch.zeros(n_rec, dtype=torch.bool), torch.ones(n_lig, dtype=torch.bool)])[sort_idx]
    assert torch.equal(bidx, kw["batch_idx"]) and torch.equal(lig, kw["lig_flag"]) and torch.equal(lig, kw["gen_flag"])
    assert torch.equal(torch.cat([batch["protein_pos"], batch["ligand_pos"]])[sort_idx], kw["x"])
    assert torch.equal(torch.cat([h_rec, h_lig])[sort_idx], kw["h"])


@pytest.mark.parametrize("override,text", [
    (dict(cutoff_mode="radius"), "Not supported cutoff mode"),           # the reference's own text (unitransformer.py:85)
    (dict(cutoff_mode="hybrid"), "Not supported cutoff mode"),
    (dict(num_x2h=2), "num_x2h/num_h2x != 1"), (dict(num_h2x=2), "num_x2h/num_h2x != 1"), (dict(num_blocks=2), "num_blocks=2"),
    (dict(ew_type="r"), "ew_type=r"), (dict(ew_type="m"), "ew_type=m"), (dict(x2h_out_fc=True), "x2h_out_fc=True"),
    (dict(n_heads=8), "n_heads=8"), (dict(node_feat_dim=64), "node_feat_dim=64"), (dict(k=16), "k=16"),
])
def test_unsupported_encoder_options_raise_valueerror_through_the_reference_factory(ref, patched_factory, override, text):
    """VERDICT r5 missing #4: options of unitransformer.py:17-39 that no shipped config sets are rejected at construction -- through
    the reference's own model class, as ValueError (the reference's convention, e3nn/__init__.py:18), naming the option"""
    from oracle import ref_shim
    with pytest.raises(ValueError, match=text):
        ref.get_model(ref_shim.targetdiff_config(13, 9, **override))


def test_time_embedding_is_rejected_by_the_host_model_class():
    """context_emb.py:190-195: a `time:` embedder key (set by no shipped config; its 'sin' branch mis-broadcasts in the reference,
    SURVEY.md A.5) is refused by cbgbench_amd's model class instead of being silently ignored"""
    import cbgbench_amd as CB
    cfg = CB.default_targetdiff_config(13)
    cfg.embedder.time = CB.Config(type="sin")
    with pytest.raises(ValueError, match="time"):
        CB.get_model(cfg)

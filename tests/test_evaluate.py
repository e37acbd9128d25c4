"""Validation metrics (cbgbench_amd/evaluate.py) against sklearn's roc_auc_score and, in the build container, against the reference's
own Evaluator / AUROC classes (repo/utils/evaluate.py:13-73) on the same result dicts."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from cbgbench_amd import evaluate as EV

REF = "/root/reference/repo/utils/evaluate.py"


def _results(seed, n=400, C=13, ties=False, one_class=False):
    g = torch.Generator().manual_seed(seed)
    v0 = torch.randint(0, 2 if one_class else C, (n,), generator=g)
    if one_class:
        v0[:] = 3
    logits = torch.randn(n, C, generator=g) + 1.5 * torch.nn.functional.one_hot(v0, C)
    p = torch.softmax(logits, -1)
    if ties:
        p = (p * 8).round() / 8          # heavy ties: the mid-rank rule matters
    return {"v0": v0, "c_pred": p, "mask_gen": torch.rand(n, generator=g) < 0.8, "vt": v0.clone()}


def test_binary_auroc_equals_sklearn_with_and_without_ties():
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(0)
    for n, levels in ((50, None), (500, None), (500, 5), (64, 2)):
        y = rng.random(n) < 0.3
        s = rng.standard_normal(n) + y
        if levels:
            s = np.round(s * levels) / levels
        assert abs(EV.binary_auroc(y, s) - sk.roc_auc_score(y, s)) < 1e-12
    with pytest.raises(ValueError):
        EV.binary_auroc(np.ones(5, bool), rng.random(5))
    with pytest.raises(ValueError):
        EV.binary_auroc(np.array([0, 1, 1], bool), np.array([0.1, np.nan, 0.3]))


def test_auroc_semantics():
    ev = EV.Evaluator([{"name": "auroc", "true_key": "v0", "pred_key": "c_pred", "mask_key": "mask_gen"}])
    assert list(ev.evaluators) == ["auroc_atom"]
    r = _results(1)
    a = ev(r)["auroc_atom"]
    assert 0.5 < a <= 1.0
    # perfect scores -> 1; a single class present -> every one-vs-rest AUROC is undefined -> 0 (the reference's bare except)
    perfect = dict(r, c_pred=torch.nn.functional.one_hot(r["v0"], 13).float())
    assert ev(perfect)["auroc_atom"] == pytest.approx(1.0)
    assert ev(_results(2, one_class=True))["auroc_atom"] == 0.0
    # evaluation mode hands over a list of result dicts (one per evaluation time): concatenated
    r2 = _results(3)
    both = {k: torch.cat([r[k], r2[k]]) for k in r}
    assert ev([r, r2])["auroc_atom"] == pytest.approx(ev(both)["auroc_atom"])
    with pytest.raises(KeyError):
        EV.Evaluator([{"name": "nope", "true_key": "a", "pred_key": "b"}])


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_auroc_equals_the_reference_class():
    pytest.importorskip("sklearn.metrics")
    spec = importlib.util.spec_from_file_location("ref_evaluate", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from cbgbench_amd.config import load_config
    cfg, _ = load_config("/root/reference/configs/denovo/train/targetdiff.yml")
    mine, theirs = EV.Evaluator(cfg.eval.metrics), ref.Evaluator(cfg.eval.metrics)
    assert list(mine.evaluators) == list(theirs.evaluators)
    for seed, kw in ((0, {}), (1, {"ties": True}), (2, {"one_class": True}), (3, {"n": 37})):
        r = _results(seed, **kw)
        a, b = mine(r), theirs(r)
        for k in b:
            if kw.get("one_class"):
                # undefined AUROC: sklearn < 1.5 (the reference's environment) raises and the reference's `except` scores 0;
                # sklearn >= 1.5 returns nan with a warning instead.  cbgbench_amd keeps the reference's 0.
                assert a[k] == 0.0 and (float(b[k]) == 0.0 or np.isnan(float(b[k])))
                continue
            assert a[k] == pytest.approx(float(b[k]), abs=1e-12), (seed, k)
    rs = [_results(5), _results(6, ties=True)]
    assert mine(rs)["auroc_atom"] == pytest.approx(float(theirs(rs)["auroc_atom"]), abs=1e-12)


def test_validate_reports_the_metrics_world1():
    """train.validate with an evaluator on a stub model: loss and metric are graph-count-weighted means over the batches"""
    from cbgbench_amd import train as TRN

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, batch):
            return {"pos": batch["loss"] + 0 * self.w.sum()}, [batch["res"]]

    ev = EV.Evaluator([{"name": "auroc", "true_key": "v0", "pred_key": "c_pred", "mask_key": "mask_gen"}])
    r1, r2 = _results(1), _results(2)
    batches = [{"loss": torch.tensor(1.0), "res": r1, "num_graphs": 3}, {"loss": torch.tensor(4.0), "res": r2, "num_graphs": 1}]
    avg, met = TRN.validate(Stub(), batches, None, ev)
    assert avg == pytest.approx((1.0 * 3 + 4.0) / 4)
    assert met["auroc_atom"] == pytest.approx((ev(r1)["auroc_atom"] * 3 + ev(r2)["auroc_atom"]) / 4)
    assert isinstance(TRN.validate(Stub(), batches, None), float)

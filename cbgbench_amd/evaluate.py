"""Validation metrics of the training driver: the reference's ``repo/utils/evaluate.py`` (``Evaluator`` :13-26, ``AUROC`` :35-73),
driven by the ``eval.metrics`` list of the train configs (``configs/denovo/train/targetdiff.yml:54-60``: auroc of the predicted
atom-type distribution ``c_pred`` against the true types ``v0`` on the generated atoms ``mask_gen``), called from the validation
loop exactly where ``train.py:235-238`` calls it.

No sklearn in the product path: ``binary_auroc`` is the Mann-Whitney statistic with mid-ranks for ties, which is what
``sklearn.metrics.roc_auc_score`` computes for a binary target (tests/test_evaluate.py checks both against each other and against
the reference's class when /root/reference is present).
"""
import numpy as np
import torch

METRIC_DICT = {}


def register_transform(name):
    def decorator(cls):
        METRIC_DICT[name] = cls
        return cls
    return decorator


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class Evaluator:
    """evaluate.py:13-26: one metric object per ``eval.metrics`` entry, reported under ``{name}_{tag}`` (tag defaults to 'atom')."""

    def __init__(self, cfg):
        self.cfg = cfg or []
        self.evaluators = {}
        for eval_cfg in self.cfg:
            name = _get(eval_cfg, "name")
            if name not in METRIC_DICT:
                raise KeyError(f"unknown metric {name!r} (known: {sorted(METRIC_DICT)})")
            kw = dict(eval_cfg) if isinstance(eval_cfg, dict) else dict(eval_cfg.items())
            self.evaluators[name + "_" + str(_get(eval_cfg, "tag", "atom"))] = METRIC_DICT[name](**kw)

    def __call__(self, results):
        return {name: fn(results) for name, fn in self.evaluators.items()}

    def __len__(self):
        return len(self.evaluators)


def merge_list_of_dict(results):
    """evaluate.py:27-32: evaluation mode returns one result dict per evaluation time; they are concatenated along dim 0"""
    return {k: torch.cat([r[k] for r in results], dim=0) for k in results[0].keys()}


def binary_auroc(y_true, y_score):
    """Area under the ROC curve of a binary target (bool array) -- P(score of a positive > score of a negative), ties counted half:
    (sum of the positives' mid-ranks - n_pos (n_pos + 1) / 2) / (n_pos n_neg).  Raises ValueError when only one class is present
    or a score is not finite, like roc_auc_score."""
    y_true = np.asarray(y_true, dtype=bool)
    y_score = np.asarray(y_score, dtype=np.float64)
    n_pos = int(y_true.sum())
    n_neg = int(y_true.size - n_pos)
    if n_pos == 0 or n_neg == 0:
        raise ValueError("only one class present in y_true: the ROC AUC is not defined")
    if not np.isfinite(y_score).all():
        raise ValueError("y_score contains NaN or infinity")
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    # mid-ranks: every group of tied scores gets the mean of the 1-based positions it occupies
    first = np.r_[True, s[1:] != s[:-1]]
    start = np.flatnonzero(first)
    end = np.r_[start[1:], s.size]
    mid = 0.5 * (start + end + 1)                       # mean of positions start+1 .. end
    ranks = np.empty(s.size, dtype=np.float64)
    ranks[order] = np.repeat(mid, end - start)
    return float((ranks[y_true].sum() - 0.5 * n_pos * (n_pos + 1)) / (n_pos * float(n_neg)))


@register_transform("auroc")
class AUROC:
    """evaluate.py:35-73: one-vs-rest AUROC per class that occurs in ``y_true``, weighted by the class counts; a class whose AUROC
    is undefined (it is the only class present, non-finite scores) contributes 0, as the reference's bare ``except`` does."""

    def __init__(self, true_key, pred_key, mask_key=None, **kwargs):
        self.true_key, self.pred_key, self.mask_key = true_key, pred_key, mask_key

    def __call__(self, results):
        if isinstance(results, (list, tuple)):
            results = merge_list_of_dict(results)
        return self.cal_auroc(results)

    def cal_auroc(self, results):
        y_true = results[self.true_key]
        y_pred = results[self.pred_key]
        mask = results.get(self.mask_key) if self.mask_key is not None else None
        if mask is None:
            mask = torch.ones_like(y_true, dtype=torch.bool)
        mask = mask.bool()
        y_true = y_true[mask].detach().cpu().numpy()
        y_pred = y_pred[mask].detach().float().cpu().numpy()
        if y_true.size == 0:
            return float("nan")                          # np.divide(0., 0) in the reference
        total = 0.0
        for c in sorted(set(y_true.tolist())):
            pos = y_true == c
            try:
                auroc = binary_auroc(pos, y_pred[:, c])
            except (ValueError, IndexError):
                auroc = 0.0
            total += auroc * float(pos.sum())
        return total / float(y_true.size)

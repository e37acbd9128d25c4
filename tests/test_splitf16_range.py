"""Range safety of the split-f16 arithmetic, on the CPU: the numpy models of the node GEMMs
(tests/lanesim_node.py) and of the fused edge kernel (tests/lanesim.py) against an fp64 evaluation of the reference formulas,
over weight scales 1e-4 .. 30 and activations from 1e-3 to beyond the f16 range.  The yardstick is what fp32 itself does (a plain
fp32 FMA chain / the fp32 oracle): the HIP arithmetic has to be as good, at every scale.  The GPU twin of this file is
tests/test_gpu_range.py (the real kernels through the C ABI)."""
import os

import numpy as np
import pytest
import torch

from oracle import unitransformer as OU
from tests import lanesim as LS
from tests import lanesim_node as M

W_SCALES = (1e-4, 1e-3, 1e-2, 1.0, 30.0)
H_SCALES = (1e-3, 1.0, 1e4, 1e5)      # 1e5 > 65504: outside the f16 range before the per-row scaling


def _rel_err(out, ref, den):
    if not np.isfinite(out).all():
        return float("inf")
    return float(np.max(np.abs(out.astype(np.float64) - ref) / den))


@pytest.mark.parametrize("ws", W_SCALES)
@pytest.mark.parametrize("hs", H_SCALES)
def test_node_gemm_model_is_fp32_grade_at_every_scale(ws, hs):
    rng = np.random.default_rng(int(1e6 * ws) + int(hs))
    W = (rng.standard_normal((128, 64)) / np.sqrt(128) * ws).astype(np.float32)
    h = (rng.standard_normal((32, 128)) * hs).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1 * ws * hs).astype(np.float32)
    ref = h.astype(np.float64) @ W.astype(np.float64) + b
    den = np.abs(h).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(b)
    new = _rel_err(M.split_gemm(h, W, b), ref, den)
    chain = _rel_err(M.fp32_chain(h, W, b), ref, den)
    assert new <= 2.5e-7 and new <= 1.5 * chain, (new, chain)


def test_unscaled_split_had_the_hole():
    """documents what the scaling fixes: without it the error is 1e-4 of sum|a||b| at weights x 1e-3 and inf beyond 65504."""
    rng = np.random.default_rng(7)
    h = rng.standard_normal((32, 128)).astype(np.float32)
    W = (rng.standard_normal((128, 64)) / np.sqrt(128) * 1e-3).astype(np.float32)
    ref = h.astype(np.float64) @ W.astype(np.float64)
    den = np.abs(h).astype(np.float64) @ np.abs(W).astype(np.float64)
    assert _rel_err(M.split_gemm(h, W, scaled=False), ref, den) > 2e-5
    assert _rel_err(M.split_gemm(h, W), ref, den) < 2e-7
    big = (h * 1e5).astype(np.float32)
    assert not np.isfinite(M.split_gemm(big, W, scaled=False)).all()
    assert np.isfinite(M.split_gemm(big, W)).all()


def test_mixed_rows_and_columns():
    """one tile with rows of very different magnitude and columns of very different magnitude, zero rows / columns included"""
    rng = np.random.default_rng(11)
    h = rng.standard_normal((16, 128)).astype(np.float32) * np.logspace(-6, 5, 16)[:, None].astype(np.float32)
    h[3] = 0.0
    W = (rng.standard_normal((128, 32)) * np.logspace(-5, 2, 32)[None, :]).astype(np.float32)
    W[:, 5] = 0.0
    out = M.split_gemm(h, W)
    ref = h.astype(np.float64) @ W.astype(np.float64)
    den = np.abs(h).astype(np.float64) @ np.abs(W).astype(np.float64)
    ok = den > 0
    assert np.all(out[~ok] == 0)
    assert float(np.max(np.abs(out.astype(np.float64) - ref)[ok] / den[ok])) < 2.5e-7


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: z[k] for k in z.files}


def _adjacency(g):
    N = g["x"].shape[0]
    nbr = np.full((N, 32), -1, np.int64)
    deg = np.zeros(N, np.int64)
    ew = np.zeros((N, 32), np.float32)
    for k, (s, d) in enumerate(g["edge_index"].T):
        nbr[d, deg[d]] = s
        ew[d, deg[d]] = g["e_w"][k, 0]
        deg[d] += 1
    return nbr, deg, ew


def scaled_first_linears(sd, prefix, fns, scale):
    """a copy of the state dict with the first Linear (weight and bias) of the given MLPs multiplied by `scale`"""
    out = dict(sd)
    for fn in fns:
        for k in ("net.0.weight", "net.0.bias"):
            out[f"{prefix}.{fn}.{k}"] = sd[f"{prefix}.{fn}.{k}"] * scale
    return out


@pytest.mark.parametrize("scale", [1e-4, 1e-3, 1e-2, 30.0])
def test_edge_kernel_lane_model_with_scaled_first_linears(golden_dir, synthetic_sd, scale):
    """the first Linears of k and v (whose rbf columns are the split-f16 tables) at other magnitudes: the lane model's h' error against
    fp64 must stay at the fp32 oracle's own error level (LayerNorm makes the block scale-invariant, so the outputs stay O(1))"""
    g = _golden(golden_dir, "denoiser_2graphs")
    nbr, deg, ew = _adjacency(g)
    x, h, lig = g["x"], g["h"], g["lig_flag"]
    pre = "denoiser.blocks.0.x2h_layers.0"
    sd = scaled_first_linears(synthetic_sd, pre, ("hk_func", "hv_func"), scale)
    Wx = LS.Weights(sd, pre, True)
    assert Wx.kw_k == LS.rbf_kw(Wx.Wr_k) and 2 ** 14 <= np.abs(Wx.Wr_k).max() * 2.0 ** Wx.kw_k < 2 ** 15 or Wx.kw_k == LS.RBF_KW_MAX
    tabs = Wx.node_tables(h, lig)
    tx = lambda a, dt: torch.from_numpy(np.asarray(a)).to(dt)
    ei = torch.from_numpy(g["edge_index"]).long()
    et = OU.build_edge_type(ei, tx(lig, torch.bool))
    ref = {}
    for dt in (torch.float32, torch.float64):
        sdd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items() if k.startswith(pre)}
        ref[dt] = OU.x2h_attention(sdd, pre, tx(x, dt), tx(h, dt), et, ei, tx(g["e_w"], dt)).numpy()
    nodes = [0, 5, 69, 70, 78, 146]
    err_model = max(np.abs(LS.simulate_node(Wx, True, i, x, h, nbr, deg, lig, ew, tabs) - ref[torch.float64][i]).max() for i in nodes)
    err_fp32 = max(np.abs(ref[torch.float32][i] - ref[torch.float64][i]).max() for i in nodes)
    assert err_model < 2e-5 and err_model < 4 * err_fp32 + 2e-6, (err_model, err_fp32)

"""The lane-level model of the x2h edge backward's tile layouts (tests/lanesim_bwd.py) against torch.autograd of the
same node function (CPU only): pins the operand layouts of every product of the backward -- both
tile labelings, the transposed softmax-gradient tile, the transposed rbf operand, the transposed d(rbf) tile."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import lanesim as LS
from tests import lanesim_bwd as LB
from tests.test_lanesim import _case, _nbr


def reference_node(W, i, x, nbr, deg, lig, e_w, tables, G):
    PDk, PDv, PSk, PSv, Qt = tables
    d, lig_i = int(deg[i]), int(lig[i])
    j = nbr[i, :d]
    src_lig = torch.from_numpy(lig[j].astype(bool))
    etype = lambda s: (0 if lig_i else 1) if s else (2 if lig_i else 3)
    ty = torch.tensor([etype(bool(s)) for s in src_lig], dtype=torch.long)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, requires_grad=True)
    dist = np.sqrt(((x[i] - x[j]) ** 2).sum(-1)).astype(np.float32)
    leaves = dict(PDk=t(PDk[i]), PDv=t(PDv[i]), PSk=t(PSk[j]), PSv=t(PSv[j]), Qt=t(Qt[i]), ew=t(e_w[i, :d]),
                  rbf=t(np.exp(-0.5 * (dist[:, None] - LS.MU[None]) ** 2)), Wr_k=t(W.Wr_k), Wr_v=t(W.Wr_v),
                  g_k=t(W.g_k), b_k=t(W.be_k), g_v=t(W.g_v), b_v=t(W.be_v))
    p = leaves

    def hidden(PD, PS, Wt, Wr, g, b):
        dWt = torch.from_numpy(Wt[etype(True)] - Wt[etype(False)])
        pre = PD + PS + src_lig[:, None].float() * dWt + torch.einsum("eg,egm->em", p["rbf"], Wr[ty])
        return torch.relu(F.layer_norm(pre, (128,), g, b, eps=1e-5))

    hk = hidden(p["PDk"], p["PSk"], W.Wt_k, p["Wr_k"], p["g_k"], p["b_k"])
    hv = hidden(p["PDv"], p["PSv"], W.Wt_v, p["Wr_v"], p["g_v"], p["b_v"])
    alpha = torch.softmax(hk @ p["Qt"].T, dim=0)
    w = alpha * p["ew"][:, None]
    Sagg = torch.einsum("ea,em->am", w, hv)
    sw = w.sum(0)
    out = torch.einsum("acm,am->ac", torch.from_numpy(W.Wb_v).reshape(16, 8, 128), Sagg).reshape(128) \
        + torch.from_numpy(W.bb_v) * sw.repeat_interleave(8)
    (torch.from_numpy(G) * out).sum().backward()
    g = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape, np.float32)) for k, v in leaves.items()}
    return g, Sagg.detach().numpy(), sw.detach().numpy(), d


@pytest.mark.parametrize("case,nodes", [("denoiser_2graphs", [0, 69, 70, 146]),
                                         ("denoiser_small_graphs", [24, 25, 58, 60, 94]),
                                         ("denoiser_linker", [3, 59, 60, 73])])
def test_two_wave_backward_model_matches_autograd(golden_dir, synthetic_sd, case, nodes):
    g = _case(golden_dir, case)
    nbr, deg, ew = _nbr(g)
    x, h, lig = g["x"], g["h"], g["lig_flag"]
    W = LS.Weights(synthetic_sd, "denoiser.blocks.0.x2h_layers.0", True)
    tabs = W.node_tables(h, lig)
    rng = np.random.default_rng(0)
    for i in nodes:
        G = rng.standard_normal(128).astype(np.float32)
        got = LB.simulate_node_backward(W, i, x, nbr, deg, lig, ew, tabs, G)
        if deg[i] == 0:        # a single-atom graph: no edge, nothing flows
            assert all(np.isfinite(v).all() and not np.asarray(v).any() for v in got.values()), (case, i)
            continue
        ref, Sagg, sw, d = reference_node(W, i, x, nbr, deg, lig, ew, tabs, G)

        def close(a, b, what, rtol=2e-4):
            scale = max(float(np.abs(b).max()), 1e-6)
            assert np.allclose(a, b, rtol=rtol, atol=2e-5 * scale + 1e-7), (case, i, what, float(np.abs(a - b).max()), scale)

        close(got["S"], Sagg, "S")
        close(got["sw"], sw, "sw")
        close(got["T"], ref["Qt"], "T = dL/dQt")
        close(got["de_w"][:d], ref["ew"], "d e_w")
        close(got["dPDk"], ref["PDk"], "d PD_k")
        close(got["dPDv"], ref["PDv"], "d PD_v")
        close(got["dPSk"][:d], ref["PSk"], "d PS_k")
        close(got["dPSv"][:d], ref["PSv"], "d PS_v")
        assert not got["dPSk"][d:].any() and not got["dPSv"][d:].any()
        close(got["dWr_k"], ref["Wr_k"], "d Wr_k")
        close(got["dWr_v"], ref["Wr_v"], "d Wr_v")
        close(got["drbf"][:d], ref["rbf"], "d rbf")
        close(got["dgam_k"], ref["g_k"], "d gamma_k")
        close(got["dbet_k"], ref["b_k"], "d beta_k")
        close(got["dgam_v"], ref["g_v"], "d gamma_v")
        close(got["dbet_v"], ref["b_v"], "d beta_v")

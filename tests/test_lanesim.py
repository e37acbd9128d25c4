"""The lane-level model of the MFMA edge kernels reproduces the reference on the golden fixtures
(CPU only): pins fragment layouts / K-permutations / fragment-ordered weight tables of edge_mfma.hip."""
import os

import numpy as np
import pytest
import torch

from oracle import unitransformer as OU
from tests import lanesim as LS


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: z[k] for k in z.files}


def _nbr(g):
    N = g["x"].shape[0]
    nbr = np.full((N, 32), -1, np.int64)
    deg = np.zeros(N, np.int64)
    for s, d in g["edge_index"].T:
        nbr[d, deg[d]] = s
        deg[d] += 1
    ew = np.zeros((N, 32), np.float32)
    pos = np.zeros(N, np.int64)
    for k, (s, d) in enumerate(g["edge_index"].T):
        ew[d, pos[d]] = g["e_w"][k, 0]
        pos[d] += 1
    return nbr, deg, ew


@pytest.mark.parametrize("case,nodes", [("denoiser_2graphs", [0, 5, 69, 70, 78, 146]),
                                         ("denoiser_small_graphs", [0, 24, 25, 57, 58, 59, 60, 93, 94]),
                                         ("denoiser_linker", [3, 59, 60, 73, 132])])
def test_lane_model_matches_reference_layer0(golden_dir, synthetic_sd, case, nodes):
    g = _case(golden_dir, case)
    nbr, deg, ew = _nbr(g)
    x, h, lig = g["x"], g["h"], g["lig_flag"]
    Wx = LS.Weights(synthetic_sd, "denoiser.blocks.0.x2h_layers.0", True)
    tabs = Wx.node_tables(h, lig)
    Wh = LS.Weights(synthetic_sd, "denoiser.blocks.0.h2x_layers.0", False)
    h1 = g["h_layer0"]
    tabs_h = Wh.node_tables(h1, lig)
    for i in nodes:
        out = LS.simulate_node(Wx, True, i, x, h, nbr, deg, lig, ew, tabs)
        assert np.allclose(out, h1[i], rtol=1e-4, atol=2e-5), (i, np.abs(out - h1[i]).max())
        dx = LS.simulate_node(Wh, False, i, x, h1, nbr, deg, lig, ew, tabs_h)
        ref_dx = g["x_layer0"][i] - x[i]
        if g["gen_flag"][i]:
            assert np.allclose(dx, ref_dx, rtol=1e-3, atol=2e-5), (i, dx, ref_dx)

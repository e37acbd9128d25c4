"""Lane-level model of the tile layouts of the x2h edge backward (docs/x2h_backward.md), written for the round-1 two-wave plan:
the kernel that was built (train_bwd_x2h.hip) runs both paths in ONE wave, with the same operand layouts per product.

Same idea as tests/lanesim.py for the forward kernel: a numpy re-enactment of what each of the 64 lanes of the key-path
wave and of the value-path wave would hold and feed to ``v_mfma_f32_16x16x4_f32``, so that the operand layouts of every
product of the backward (both tile labelings of the hidden activations, the 16 x 16 transpose of the softmax-gradient tile,
the transposed rbf operand of the weight-gradient product, the transposed d(rbf) tile) are pinned on the CPU against
torch.autograd before any HIP is written.  tests/test_lanesim_bwd.py runs it.

Per destination node i with G = dL/dh_out[i]:
    key wave   : T[a][m] = dL/dQt[i][a][m], d PD_k[i], d PS_k[j_e], d Wr_k[type], d rbf (k part), LayerNorm affine grads
    value wave : S[a][m] = sum_e alpha e_w hid_v[e][m], sw[a] = sum_e alpha e_w, d e_w[e], d PD_v[i], d PS_v[j_e],
                 d Wr_v[type], d rbf (v part)
exchanged between the two waves (through LDS in the kernel): alpha (key -> value), d alpha (value -> key).
"""
import numpy as np

from tests.lanesim import C_, L, MU, Q_, frag_wr_chan, m_chan, m_edge, mfma


def xq(v):
    """sum across the four q rows of a lane column (v_permlane16/32_swap in the kernel): every lane gets the total"""
    v = v + v[..., L ^ 16]
    return v + v[..., L ^ 32]


def xc(v):
    """sum across the 16 lanes of a row (DPP row reductions): every lane gets the total"""
    for o in (1, 2, 4, 8):
        v = v + v[..., L ^ o]
    return v


def transpose_e1(tile):
    """E1 tile [hf][r][lane (c = a, q)] <-> edge 4q + r + 16hf   ->   [hf][s][lane (c = edge16, q)] = tile[e = c][a = 4s + q]
    (the 16 x 16 transpose per half that goes through LDS in the kernel)."""
    out = np.zeros((2, 4, 64), np.float32)
    for hf in range(2):
        for s in range(4):
            out[hf, s] = tile[hf, C_ & 3, (4 * s + Q_) + 16 * (C_ >> 2)]
    return out


def simulate_node_backward(W, i, x, nbr, deg, lig, e_w, tables, G):
    PDk, PDv, PSk, PSv, Qt = tables
    d = int(deg[i])
    lig_i = int(lig[i])
    f32 = np.float32
    # ---- geometry, E0 mapping: lane (c, q) <-> edge c + 16hf (as the forward kernel) --------------------------------
    e0 = [C_ + 16 * hf for hf in range(2)]
    valid0 = [e < d for e in e0]
    j0 = [np.where(v, nbr[i, e], i) for v, e in zip(valid0, e0)]
    lg0 = [lig[j].astype(bool) & v for j, v in zip(j0, valid0)]
    dist0 = [np.sqrt(((x[i] - x[j]) ** 2).sum(-1)).astype(f32) for j in j0]
    R = [[np.where(valid0[hf], np.exp(-0.5 * (dist0[hf] - MU[4 * s + Q_]) ** 2), 0).astype(f32) for s in range(5)]
         for hf in range(2)]
    # E1 mapping: lane (c = head a, q), register (hf, r) <-> edge 4q + r + 16hf
    e1 = np.stack([[4 * Q_ + r + 16 * hf for r in range(4)] for hf in range(2)])
    valid1 = e1 < d
    j1 = np.where(valid1, nbr[i, np.minimum(e1, 31)], i)
    lg1 = lig[j1].astype(bool) & valid1
    dist1 = np.sqrt(((x[i] - x[j1]) ** 2).sum(-1)).astype(f32)           # [2][4][64]  (ds_bpermute of dist0 in the kernel)
    ew1 = np.where(valid1, e_w[i, np.minimum(e1, 31)], 0).astype(f32)
    passes = [p for p in (False, True) if any((lg0[hf] == p)[valid0[hf]].any() for hf in range(2))] or [False]

    def etype(src_lig):
        return (0 if lig_i else 1) if src_lig else (2 if lig_i else 3)

    def pre_both(PD, PS, Wt, Wr):
        """pre-activation in both tile labelings.
        edge-major   E[t][hf][r][lane (c = edge16, q)]    <-> channel m_edge(t, 4q + r)
        channel-major C[t][hf][r][lane (c, q)]            <-> edge 4q + r + 16hf, channel m_chan(t, c)"""
        fa = np.zeros((4, 8, 5, 64), f32)
        for t in range(8):
            for s in range(5):
                fa[:, t, s, :] = Wr[:, 4 * s + Q_, m_edge(t, C_)]
        fb = frag_wr_chan(Wr)
        dWt = Wt[etype(True)] - Wt[etype(False)]
        E = np.zeros((8, 2, 4, 64), f32)
        Cm = np.zeros((8, 2, 4, 64), f32)
        for t in range(8):
            for hf in range(2):
                for r in range(4):
                    m = m_edge(t, 4 * Q_ + r)
                    E[t, hf, r] = PD[i, m] + PS[j0[hf], m] + np.where(lg0[hf], dWt[m], 0)
                    mc = m_chan(t, C_)
                    Cm[t, hf, r] = PD[i, mc] + PS[j1[hf, r], mc] + np.where(lg1[hf, r], dWt[mc], 0)
                for p in passes:
                    for s in range(5):
                        Rm = np.where(lg0[hf] == p, R[hf][s], 0).astype(f32)
                        E[t, hf] = mfma(fa[etype(p), t, s], Rm, E[t, hf])
                        Cm[t, hf] = mfma(Rm, fb[etype(p), t, s], Cm[t, hf])
        return E, Cm

    def ln_fwd(E, Cm, gamma, beta):
        """-> (n, rstd, hid) in both labelings (zero mean by construction: centred first Linear)"""
        vE = xq((E * E).sum((0, 2)))                                    # [hf][lane]: per edge c + 16hf
        rE = (1.0 / np.sqrt(vE / 128 + 1e-5)).astype(f32)
        nE = E * rE[None, :, None, :]
        vC = xc((Cm * Cm).sum(0))                                       # [hf][r][lane]: per edge 4q + r + 16hf
        rC = (1.0 / np.sqrt(vC / 128 + 1e-5)).astype(f32)
        nC = Cm * rC[None]
        gE = np.stack([np.stack([gamma[m_edge(t, 4 * Q_ + r)] for r in range(4)]) for t in range(8)])   # [t][r][lane]
        bE = np.stack([np.stack([beta[m_edge(t, 4 * Q_ + r)] for r in range(4)]) for t in range(8)])
        gC = np.stack([gamma[m_chan(t, C_)] for t in range(8)])                                          # [t][lane]
        bC = np.stack([beta[m_chan(t, C_)] for t in range(8)])
        hE = np.maximum(nE * gE[:, None] + bE[:, None], 0).astype(f32)
        hC = np.maximum(nC * gC[:, None, None] + bC[:, None, None], 0).astype(f32)
        return (nE.astype(f32), rE, hE, gE), (nC.astype(f32), rC, hC, gC)

    def contract_channels(hE, Bsrc):
        """[hf][r][lane (c = a, q)] = sum_m hid[e = 4q + r + 16hf][m] Bsrc[a][m]   (scores / gv: the forward's product)"""
        out = np.zeros((2, 4, 64), f32)
        for hf in range(2):
            for t in range(8):
                for r in range(4):
                    out[hf] = mfma(hE[t, hf, r], Bsrc[C_, m_edge(t, 4 * Q_ + r)], out[hf])
        return out

    def fold_edges(hC, wE1):
        """F[a][m] = sum_e w[e][a] hid[e][m]: hid channel-major as A, the E1 tile as B (the forward's aggregation product).
        D tile t: lane (c = a, q) reg r' <-> channel m_chan(t, 4q + r')."""
        F = np.zeros((16, 128), f32)
        for t in range(8):
            acc = np.zeros((4, 64), f32)
            for hf in range(2):
                for r in range(4):
                    acc = mfma(hC[t, hf, r], wE1[hf, r], acc)
            for rp in range(4):
                F[C_, m_chan(t, 4 * Q_ + rp)] = acc[rp]
        return F

    def d_hidden(wE1, Bsrc):
        """d hid[e][m] = sum_a w[e][a] Bsrc[a][m] in both labelings; w enters transposed (transpose_e1)."""
        wT = transpose_e1(wE1)
        dE = np.zeros((8, 2, 4, 64), f32)
        dC = np.zeros((8, 2, 4, 64), f32)
        for t in range(8):
            for hf in range(2):
                for s in range(4):
                    a = 4 * s + Q_
                    dE[t, hf] = mfma(Bsrc[a, m_edge(t, C_)], wT[hf, s], dE[t, hf])      # D[row = channel][col = edge]
                    dC[t, hf] = mfma(wT[hf, s], Bsrc[a, m_chan(t, C_)], dC[t, hf])      # D[row = edge][col = channel]
        return dE, dC

    def ln_bwd(dE, dC, fE, fC):
        (nE, rE, hE, gE), (nC, rC, hC, gC) = fE, fC
        dnE = np.where(hE > 0, dE * gE[:, None], 0).astype(f32)
        s1 = xq(dnE.sum((0, 2))) / 128
        s2 = xq((dnE * nE).sum((0, 2))) / 128
        dpE = rE[None, :, None, :] * (dnE - s1[None, :, None, :] - nE * s2[None, :, None, :])
        dnC = np.where(hC > 0, dC * gC[:, None, None], 0).astype(f32)
        c1 = xc(dnC.sum(0)) / 128
        c2 = xc((dnC * nC).sum(0)) / 128
        dpC = rC[None] * (dnC - c1[None] - nC * c2[None])
        # affine gradients from the edge-major copy: per lane [t][r] <-> channel m_edge(t, 4q + r), summed over the 16 edge lanes
        dy = np.where(hE > 0, dE, 0).astype(f32)
        dgam, dbet = np.zeros(128, f32), np.zeros(128, f32)
        for t in range(8):
            for r in range(4):
                m = m_edge(t, 4 * Q_ + r)
                np.add.at(dgam, m, (dy[t, :, r] * nE[t, :, r]).sum(0))
                np.add.at(dbet, m, dy[t, :, r].sum(0))
        return dpE.astype(f32), dpC.astype(f32), dgam, dbet

    def scatter_dP(dpE):
        """d PD[i][m] = sum_e dpre[e][m] ; d PS[slot e][m] = dpre[e][m]  (edge-major: lane c = edge, 4 consecutive channels)"""
        dPD = np.zeros(128, f32)
        dPS = np.zeros((32, 128), f32)
        for t in range(8):
            for hf in range(2):
                for r in range(4):
                    m = m_edge(t, 4 * Q_ + r)
                    np.add.at(dPD, m, np.where(valid0[hf], dpE[t, hf, r], 0))
                    dPS[e0[hf], m] = np.where(valid0[hf], dpE[t, hf, r], 0)
        return dPD, dPS

    def d_weights_rbf(dpC):
        """d Wr[type][g][m] = sum_{e of that type} rbf_g(d_e) dpre[e][m]: contraction over edges, so the rbf enters transposed,
        A[i = g][k = edge]: lane (c = g16, q) holds rbf_{c + 16gt}(d_{4q + r + 16hf}); B = dpre channel-major."""
        dWr = np.zeros((4, 20, 128), f32)
        for p in passes:
            for gt in range(2):
                g = C_ + 16 * gt
                acc = np.zeros((8, 4, 64), f32)
                for hf in range(2):
                    for r in range(4):
                        rT = np.where(valid1[hf, r] & (lg1[hf, r] == p) & (g < 20),
                                      np.exp(-0.5 * (dist1[hf, r] - MU[np.minimum(g, 19)]) ** 2), 0).astype(f32)
                        for t in range(8):
                            acc[t] = mfma(rT, dpC[t, hf, r], acc[t])            # D[row = g 4q'+r'][col = channel c']
                for t in range(8):
                    for rp in range(4):
                        gg = 4 * Q_ + rp + 16 * gt
                        ok = gg < 20
                        dWr[etype(p), gg[ok], m_chan(t, C_)[ok]] += acc[t, rp][ok]
        return dWr

    def d_rbf(dpE, Wr):
        """d rbf[e][g] = sum_m dpre[e][m] Wr[type_e][g][m], as the transposed tile D[row = g][col = edge]:
        A = Wr^T fragment (lane (c = g16, q) <-> channel), B = dpre edge-major masked by source class."""
        out = np.zeros((32, 20), f32)
        for hf in range(2):
            for gt in range(2):
                g = np.minimum(C_ + 16 * gt, 19)
                acc = np.zeros((4, 64), f32)
                for p in passes:
                    for t in range(8):
                        for r in range(4):
                            a = np.where(C_ + 16 * gt < 20, Wr[etype(p), g, m_edge(t, 4 * Q_ + r)], 0).astype(f32)
                            b = np.where(lg0[hf] == p, dpE[t, hf, r], 0).astype(f32)
                            acc = mfma(a, b, acc)
                for rp in range(4):
                    gg = 4 * Q_ + rp + 16 * gt
                    ok = (gg < 20) & valid0[hf]
                    out[e0[hf][ok], gg[ok]] = acc[rp][ok]
        return out

    # =================================== key wave ===================================================================
    Ek, Ck = pre_both(PDk, PSk, W.Wt_k, W.Wr_k)
    fEk, fCk = ln_fwd(Ek, Ck, W.g_k, W.be_k)
    sc = np.where(valid1, contract_channels(fEk[2], Qt[i]), -np.inf)
    mx = sc.max((0, 1)); mx = np.maximum(mx, mx[L ^ 16]); mx = np.maximum(mx, mx[L ^ 32])
    ex = np.where(valid1, np.exp(np.where(valid1, sc - np.where(np.isfinite(mx), mx, 0), 0)), 0).astype(f32)
    den = xq(ex.sum((0, 1)))
    alpha = np.where(valid1, ex / np.where(den > 0, den, 1), 0).astype(f32)          # -> value wave
    # =================================== value wave ================================================================
    Gt = np.einsum("ac,acm->am", G.reshape(16, 8), W.Wb_v.reshape(16, 8, 128)).astype(f32)   # fold_grad_kernel
    gb = (G.reshape(16, 8) * W.bb_v.reshape(16, 8)).sum(-1).astype(f32)
    Ev, Cv = pre_both(PDv, PSv, W.Wt_v, W.Wr_v)
    fEv, fCv = ln_fwd(Ev, Cv, W.g_v, W.be_v)
    gv = contract_channels(fEv[2], Gt) + gb[C_]
    dalpha = (ew1 * gv).astype(f32)                                                    # -> key wave
    de_w = np.zeros(32, f32)
    tot = xc(alpha * gv)                                                               # sum over heads (the 16 lanes of a row)
    for hf in range(2):
        for r in range(4):
            ok = valid1[hf, r]
            de_w[e1[hf, r][ok]] = tot[hf, r][ok]
    w = (alpha * ew1).astype(f32)
    swl = xq(w.sum((0, 1)))
    sw = np.array([swl[a] for a in range(16)], f32)
    Sagg = fold_edges(fCv[2], w)
    dEv, dCv = d_hidden(w, Gt)
    dpEv, dpCv, dgam_v, dbet_v = ln_bwd(dEv, dCv, fEv, fCv)
    dPDv, dPSv = scatter_dP(dpEv)
    dWr_v = d_weights_rbf(dpCv)
    drbf_v = d_rbf(dpEv, W.Wr_v)
    # =================================== key wave, continued ==========================================================
    dot = xq((alpha * dalpha).sum((0, 1)))
    ds = (alpha * (dalpha - dot)).astype(f32)
    T = fold_edges(fCk[2], ds)
    dEk, dCk = d_hidden(ds, Qt[i])
    dpEk, dpCk, dgam_k, dbet_k = ln_bwd(dEk, dCk, fEk, fCk)
    dPDk, dPSk = scatter_dP(dpEk)
    dWr_k = d_weights_rbf(dpCk)
    drbf_k = d_rbf(dpEk, W.Wr_k)
    return dict(T=T, S=Sagg, sw=sw, de_w=de_w, dPDk=dPDk, dPDv=dPDv, dPSk=dPSk, dPSv=dPSv, dWr_k=dWr_k, dWr_v=dWr_v,
                drbf=drbf_k + drbf_v, dgam_k=dgam_k, dbet_k=dbet_k, dgam_v=dgam_v, dbet_v=dbet_v)

"""Lane-level model of the MFMA edge kernels (cbgbench_amd/csrc/edge_mfma.hip).

A numpy re-enactment of what each of the 64 lanes of a wavefront holds and feeds to
``v_mfma_f32_16x16x4_f32`` in the fused x2h / h2x kernels.  It exists to pin the *index math* of the
kernel (fragment layouts, K-permutation tricks, fragment-ordered weight tables) on the CPU, where it can
be compared with the oracle; the HIP kernel mirrors it line by line.  ``tests/test_lanesim.py`` runs it.

MFMA 16x16x4 f32 operand maps (cdna_hip_programming.md section 3): lane l, c = l & 15, q = l >> 4
    A[i = c][k = q]      B[k = q][j = c]      C/D reg r: [row = 4q + r][col = c]
v_mfma_f32_16x16x16_f16 (the rbf pre-activation of protein destinations, "split-f16" arithmetic): four f16 per lane,
    A[i = c][k = 4q + j]   B[k = 4q + j][j = c]   same C/D map (scripts/ubench/mfma_f16_check.hip probes the K = 32 sibling)

Split-f16: an fp32 value v is carried as hi = f16(v) (round to nearest) and lo = f16(v - hi); a product of two such values is
hi*hi + hi*lo + lo*hi accumulated in fp32 (the lo*lo term, 2^-22 relative, is dropped).  The rbf weights are split at pack
time, the 5 rbf values of a lane per node.  Per (type, tile t) a lane owns five dwords of weight pieces in LDS
    d0 = [h0 h1]  d1 = [h2 h3]  d2 = [h4 l0]  d3 = [l1 l2]  d4 = [l3 l4]          (index = s of g = 4s + q)
-- 20 bytes, exactly the fp32 footprint -- read as the tuples T1 = (d0, d1), T2 = (d2, d3), T3 = (d4, d2), which four MFMAs
contract with the rbf tuples  [rh0..3], [rl0..3], [rh4 rh0 rh1 rh2], [rh3 rh4 rl4 0]  (T1 is used twice, d2 is read twice).

Range safety (csrc/layout.h A_RBF_SC): the weight table of a path holds Wr 2^kw with kw chosen at pack time so that the largest
entry lies in [2^14, 2^15) (clamped to RBF_KW_MAX), the rbf values are produced times 2^RBF_EXP, and the pre-activation tile is
carried scaled by S = 2^(kw + RBF_EXP) -- PD + PS and the type column enter it times S, LayerNorm divides S out again.  All
factors are powers of two: the only effect is that hi / lo never fall into the f16 subnormal range.
"""
import numpy as np

L = np.arange(64)
C_ = L & 15
Q_ = L >> 4
MU = np.array([0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10], np.float32)


def mfma(a, b, c):
    """a[64], b[64], c[4][64] -> d[4][64]  (D = A.B + C, fp32)."""
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[C_, Q_] = a
    B[Q_, C_] = b
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    return np.stack([c[r] + D[4 * Q_ + r, C_] for r in range(4)])


def mfma_f16(a4, b4, c):
    """v_mfma_f32_16x16x16_f16: a4[4 slots][64], b4[4][64] (float16), c[4][64] -> d[4][64]; products exact, fp32 accumulate."""
    A = np.zeros((16, 16), np.float64)
    B = np.zeros((16, 16), np.float64)
    for j in range(4):
        A[C_, 4 * Q_ + j] = a4[j].astype(np.float64)
        B[4 * Q_ + j, C_] = b4[j].astype(np.float64)
    D = (A @ B).astype(np.float32)
    return np.stack([c[r] + D[4 * Q_ + r, C_] for r in range(4)])


RBF_EXP, RBF_KW_MAX = 12, 20


def rbf_kw(Wr):
    """pack_rbf_scale_kernel: exponent kw of a path's rbf tables (all four edge types share it)."""
    mx = np.float32(np.abs(Wr).max())
    E = (mx.view(np.uint32) >> 23) & 0xff
    return int(max(-80, min(RBF_KW_MAX, 141 - int(E))))


def split_f16(v):
    hi = v.astype(np.float16)
    lo = (v.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def rbf_tuples(R5):
    """R5[5][64] fp32 (index s, g = 4s + q) -> the four B tuples [4][4 slots][64] float16."""
    h, l = zip(*[split_f16(r) for r in R5])
    z = np.zeros(64, np.float16)
    return [np.stack([h[0], h[1], h[2], h[3]]), np.stack([l[0], l[1], l[2], l[3]]),
            np.stack([h[4], h[0], h[1], h[2]]), np.stack([h[3], h[4], l[4], z])]


def frag16(Wr, labeling):
    """Wr[4 types][20 g][128 m] -> [type][t][tuple 3][slot 4][lane] float16: the weight tuples of the split-f16 rbf MFMAs
    (lane (c, q): channel labeling(t, c), g = 4s + q), i.e. (d0, d1), (d2, d3), (d4, d2) of the LDS record."""
    out = np.zeros((4, 8, 3, 4, 64), np.float16)
    for t in range(8):
        m = labeling(t, C_)
        h, l = zip(*[split_f16(Wr[:, 4 * s + Q_, m]) for s in range(5)])     # each [4 types][64]
        z = np.zeros_like(h[0])
        out[:, t, 0] = np.stack([h[0], h[1], h[2], h[3]], 1)
        out[:, t, 1] = np.stack([h[4], l[0], l[1], l[2]], 1)
        out[:, t, 2] = np.stack([l[3], l[4], h[4], l[0]], 1)
    return out


# ---- hidden-channel labelings ---------------------------------------------------------------------
def m_edge(t, rho):
    """'edge-major' tiles (lane column = edge): C row rho = 4q+r of tile t  <->  m = 16t + rho."""
    return 16 * t + rho


def m_chan(t, c):
    """'channel-major' tiles (lane column = channel): column c of tile t  <->  m = 64(t>>2) + 4c + (t&3)."""
    return 64 * (t >> 2) + 4 * c + (t & 3)


# ---- fragment-ordered weight tables (what cbgx_pack_weights writes, what the kernel copies to LDS) ----
def frag_wr_edge(Wr):
    """Wr[4 types][20 g][128 m] -> [type][t][s][lane]: A operand of the edge-major pre-activation MFMA.
    lane (c = rho, q = kk):  Wr[type][4s + kk][m_edge(t, rho)]."""
    out = np.zeros((4, 8, 5, 64), np.float32)
    for t in range(8):
        for s in range(5):
            out[:, t, s, :] = Wr[:, 4 * s + Q_, m_edge(t, C_)]
    return out


def frag_wr_chan(Wr):
    """-> [type][t][s][lane]: B operand of the channel-major MFMA. lane (c, q = kk): Wr[type][4s+kk][m_chan(t,c)]."""
    out = np.zeros((4, 8, 5, 64), np.float32)
    for t in range(8):
        for s in range(5):
            out[:, t, s, :] = Wr[:, 4 * s + Q_, m_chan(t, C_)]
    return out


def ln_relu(v, gamma, beta):
    mean = v.mean(-1, keepdims=True, dtype=np.float32)
    var = ((v - mean) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return np.maximum((v - mean) / np.sqrt(var + np.float32(1e-5)) * gamma + beta, 0).astype(np.float32)


class Weights:
    """Per-attention weights in the packed (factored) form of csrc/layout.h, as numpy arrays."""

    def __init__(self, sd, prefix, x2h):
        kf, vf, qf = ("hk_func", "hv_func", "hq_func") if x2h else ("xk_func", "xv_func", "xq_func")
        g = lambda n: sd[f"{prefix}.{n}"].numpy()
        wk0, wv0 = g(f"{kf}.net.0.weight"), g(f"{vf}.net.0.weight")
        # pack-time centring over the 128 output channels (cbgx_pack_weights / center_linear_kernel)
        wk0 = (wk0 - wk0.mean(0, keepdims=True, dtype=np.float32)).astype(np.float32)
        wv0 = (wv0 - wv0.mean(0, keepdims=True, dtype=np.float32)).astype(np.float32)
        self.Wt_k, self.Wt_v = wk0[:, :4].T.copy(), wv0[:, :4].T.copy()                 # [4][128]
        self.Wr_k = wk0[:, 4:84].T.reshape(4, 20, 128).copy()                            # [type][g][m]
        self.Wr_v = wv0[:, 4:84].T.reshape(4, 20, 128).copy()
        self.Wd_k, self.Ws_k = wk0[:, 84:212], wk0[:, 212:340]
        self.Wd_v, self.Ws_v = wv0[:, 84:212], wv0[:, 212:340]
        self.b_k0, self.b_v0 = g(f"{kf}.net.0.bias"), g(f"{vf}.net.0.bias")
        self.b_k0 = self.b_k0 - self.b_k0.mean(dtype=np.float32)
        self.b_v0 = self.b_v0 - self.b_v0.mean(dtype=np.float32)
        self.g_k, self.be_k = g(f"{kf}.net.1.weight"), g(f"{kf}.net.1.bias")
        self.g_v, self.be_v = g(f"{vf}.net.1.weight"), g(f"{vf}.net.1.bias")
        self.Wb_k = g(f"{kf}.net.3.weight")
        self.Wb_v, self.bb_v = g(f"{vf}.net.3.weight"), g(f"{vf}.net.3.bias")
        self.q = [g(f"{qf}.net.{i}.{w}") for i in (0, 1, 3) for w in ("weight", "bias")]
        self.fragA_k = frag_wr_edge(self.Wr_k)
        self.fragA_v = frag_wr_edge(self.Wr_v)
        self.fragB_v = frag_wr_chan(self.Wr_v)
        # split-f16 tuples (protein destinations): A operand edge-major (k; h2x v), B operand channel-major (x2h v)
        self.kw_k, self.kw_v = rbf_kw(self.Wr_k), rbf_kw(self.Wr_v)
        self.S_k, self.S_v = np.float32(2.0 ** (self.kw_k + RBF_EXP)), np.float32(2.0 ** (self.kw_v + RBF_EXP))
        up = lambda w, kw: np.ldexp(w, kw).astype(np.float32)
        self.frag16A_k = frag16(up(self.Wr_k, self.kw_k), m_edge)
        self.frag16A_v = frag16(up(self.Wr_v, self.kw_v), m_edge)
        self.frag16B_v = frag16(up(self.Wr_v, self.kw_v), m_chan)

    def node_tables(self, h, lig):
        """What the node kernels produce: P = [PDk | PDv | PSk | PSv] and the folded query Qt.
        PD carries the centred bias and the type column of a protein source for the node's class."""
        tp = np.where(lig.astype(bool), 2, 3)
        PDk = h @ self.Wd_k.T + self.b_k0 + self.Wt_k[tp]
        PDv = h @ self.Wd_v.T + self.b_v0 + self.Wt_v[tp]
        PSk, PSv = h @ self.Ws_k.T, h @ self.Ws_v.T
        w0, b0, gq, bq, w1, b1 = self.q
        qv = ln_relu(h @ w0.T + b0, gq, bq) @ w1.T + b1                                   # [N,128]
        Qt = np.einsum("nac,acm->nam", qv.reshape(-1, 16, 8), self.Wb_k.reshape(16, 8, 128)) / np.sqrt(8.0)
        return (PDk.astype(np.float32), PDv.astype(np.float32), PSk.astype(np.float32), PSv.astype(np.float32),
                Qt.astype(np.float32))


def simulate_node(W, x2h, i, x, h, nbr, deg, lig, e_w, tables):
    """One wavefront = one destination node i.  Returns h_out[i] (x2h) or delta_x[i] (h2x)."""
    PDk, PDv, PSk, PSv, Qt = tables
    d = int(deg[i])
    lig_i = int(lig[i])
    # ---- stage 0: edge geometry in the E0 mapping: lane (c, q) <-> edges e = c + 16 hf -------------------
    e0 = [C_ + 16 * hf for hf in range(2)]
    valid0 = [e < d for e in e0]
    j0 = [np.where(v, nbr[i, e], i) for v, e in zip(valid0, e0)]
    lg0 = [lig[j].astype(bool) & v for j, v in zip(j0, valid0)]
    dist0 = [np.sqrt(((x[i] - x[j]) ** 2).sum(-1)).astype(np.float32) for j in j0]
    # R[hf][s][lane] = rbf_{4s+q}(d_e): B operand (edge-major MFMA) and A operand (channel-major MFMA)
    R = [[np.where(valid0[hf], np.exp(-0.5 * (dist0[hf] - MU[4 * s + Q_]) ** 2).astype(np.float32) * np.float32(2 ** RBF_EXP), 0)
          .astype(np.float32) for s in range(5)] for hf in range(2)]     # times 2^RBF_EXP: the validity factor is 0 / 2^RBF_EXP
    mask_lig = 0
    for hf in range(2):
        for c in range(16):
            if lg0[hf][c]:
                mask_lig |= 1 << (c + 16 * hf)
    mask_valid = (1 << d) - 1
    passes = []
    if (~mask_lig) & mask_valid or d == 0:
        passes.append(False)
    if mask_lig & mask_valid:
        passes.append(True)

    def etype(src_lig):
        return (0 if lig_i else 1) if src_lig else (2 if lig_i else 3)

    def rbf_masked(hf, src_lig):
        return [np.where(lg0[hf] == src_lig, R[hf][s], 0).astype(np.float32) for s in range(5)]

    def pre_edge_major(PD, PS, Wt, frag, frag16_, S):
        """C[t][hf][r][lane]: lane (c = e16, q), m = 32q + 4t + r (edge-major); the rbf term in split-f16 MFMAs.
        The tile is carried scaled by S."""
        Cacc = np.zeros((8, 2, 4, 64), np.float32)
        for t in range(8):
            for hf in range(2):
                dWt = Wt[etype(True)] - Wt[etype(False)]
                for r in range(4):
                    m = m_edge(t, 4 * Q_ + r)
                    Cacc[t, hf, r] = (PS[j0[hf], m] * S + PD[i, m] * S) + np.where(lg0[hf], dWt[m] * S, 0)
                for src_lig in passes:
                    T, Bt = frag16_[etype(src_lig), t], rbf_tuples(rbf_masked(hf, src_lig))
                    for tu, b in ((0, 0), (0, 1), (1, 2), (2, 3)):
                        Cacc[t, hf] = mfma_f16(T[tu], Bt[b], Cacc[t, hf])
        return Cacc

    def ln_edge_major(Cacc, gamma, beta, S):
        c1, c2 = np.float32(1.0) / (np.float32(128) * S * S), np.float32(1.0) / S
        out = np.zeros_like(Cacc)
        for hf in range(2):
            dv = Cacc[:, hf]                                 # zero mean by construction (centred weights)
            v = (dv * dv).sum((0, 1))                        # in-lane over (t, r), then across q
            v = v + v[L ^ 16]; v = v + v[L ^ 32]
            rstd = (1.0 / np.sqrt(v * c1 + 1e-5)) * c2
            for t in range(8):
                for r in range(4):
                    m = m_edge(t, 4 * Q_ + r)
                    out[t, hf, r] = np.maximum(dv[t, r] * rstd * gamma[m] + beta[m], 0)
        return out.astype(np.float32)

    def contract_channels(Hd, Bsrc):
        """C[hf][r][lane (c = a, q)] = sum_m hid[e = 4q + r + 16hf][m] * Bsrc[a][m]."""
        out = np.zeros((2, 4, 64), np.float32)
        for hf in range(2):
            for t in range(8):
                for r in range(4):
                    out[hf] = mfma(Hd[t, hf, r], Bsrc[C_, m_edge(t, 4 * Q_ + r)], out[hf])
        return out

    # ---- k path (edge-major) -----------------------------------------------------------------------
    Hk = ln_edge_major(pre_edge_major(PDk, PSk, W.Wt_k, W.fragA_k, W.frag16A_k, W.S_k), W.g_k, W.be_k, W.S_k)
    S = contract_channels(Hk, Qt[i])                           # scores: lane (a, q) reg r <-> e = 4q + r + 16hf
    e1 = np.stack([[4 * Q_ + r + 16 * hf for r in range(4)] for hf in range(2)])   # [2][4][64]
    valid1 = e1 < d
    S = np.where(valid1, S, -np.inf)
    mx = S.max((0, 1)); mx = np.maximum(mx, mx[L ^ 16]); mx = np.maximum(mx, mx[L ^ 32])
    ex = np.where(valid1, np.exp(S - mx), 0).astype(np.float32)
    den = ex.sum((0, 1)); den = den + den[L ^ 16]; den = den + den[L ^ 32]
    alpha = np.where(valid1, ex / np.where(den > 0, den, 1), 0).astype(np.float32)
    ew1 = np.where(valid1, e_w[i, np.minimum(e1, 31)], 0).astype(np.float32)

    if x2h:
        w = alpha * ew1
        sw = w.sum((0, 1)); sw = sw + sw[L ^ 16]; sw = sw + sw[L ^ 32]           # sum_e alpha e_w, per head a = c
        # ---- v path (channel-major): lane (c = m16, q) reg r <-> e = 4q + r + 16hf, m = 8c + t
        j1 = np.where(valid1, nbr[i, np.minimum(e1, 31)], i)
        lg1 = ((mask_lig >> e1) & 1).astype(bool)
        Cv = np.zeros((8, 2, 4, 64), np.float32)
        for t in range(8):
            m = m_chan(t, C_)
            for hf in range(2):
                dWt = W.Wt_v[etype(True)] - W.Wt_v[etype(False)]
                for r in range(4):
                    Cv[t, hf, r] = (PSv[j1[hf, r], m] * W.S_v + PDv[i, m] * W.S_v) + np.where(lg1[hf, r], dWt[m] * W.S_v, 0)
                for src_lig in passes:
                    # rbf tuples as the A operand, weight tuples as B
                    T, Bt = W.frag16B_v[etype(src_lig), t], rbf_tuples(rbf_masked(hf, src_lig))
                    for tu, b in ((0, 0), (0, 1), (1, 2), (2, 3)):
                        Cv[t, hf] = mfma_f16(Bt[b], T[tu], Cv[t, hf])
        # LN over m = (t, c): in-lane over t, across the 16 lanes of a row (xor 1,2,4,8)
        Hv = np.zeros_like(Cv)
        for hf in range(2):
            for r in range(4):
                dv = Cv[:, hf, r]                             # zero mean by construction
                v = (dv * dv).sum(0)
                for o in (1, 2, 4, 8): v = v + v[L ^ o]
                rstd = (1.0 / np.sqrt(v * (np.float32(1.0) / (np.float32(128) * W.S_v * W.S_v)) + 1e-5)) * (np.float32(1.0) / W.S_v)
                for t in range(8):
                    m = m_chan(t, C_)
                    Hv[t, hf, r] = np.maximum(dv[t] * rstd * W.g_v[m] + W.be_v[m], 0)
        # v-agg, transposed (hid_v as A operand, w as B): S2[t][r'][lane (c = a, q)] <-> channel m_chan(t, 4q + r')
        S2 = np.zeros((8, 4, 64), np.float32)
        for t in range(8):
            for hf in range(2):
                for r in range(4):
                    S2[t] = mfma(Hv[t, hf, r], w[hf, r], S2[t])
        # epilogue: lane (a, q) holds S[a][32q .. 32q+31]; Wbv image has its 16-byte chunks XOR-swizzled by the head
        img = np.zeros(128 * 128, np.float32)
        n_, m_ = np.meshgrid(np.arange(128), np.arange(128), indexing="ij")
        img[n_ * 128 + ((((m_ >> 2) ^ ((n_ >> 3) & 15)) << 2) | (m_ & 3))] = W.Wb_v
        out = np.zeros(128, np.float32)
        for lane in range(64):
            c, q = lane & 15, lane >> 4
            for cc in range(8):
                acc = 0.0
                for rp in range(4):
                    for t in range(8):                                  # S2[t][rp] <-> channel m_chan(t, 4q + rp)
                        chunk = (16 * (t >> 2) + 4 * q + rp) ^ c
                        acc += img[(8 * c + cc) * 128 + (chunk << 2) + (t & 3)] * S2[t, rp, lane]
                out[8 * c + cc] += acc                                  # summed over q by xrow_sum in the kernel
        sw_head = np.array([sw[a] for a in range(16)])          # lane c = a (any q) holds sw[a]
        out = out + W.bb_v * np.repeat(sw_head, 8)
        return h[i] + out
    # ---- h2x: v path edge-major as well; wv[e, a] = Wbv[a] . hid_v[e] + bbv[a] --------------------------
    Hv = ln_edge_major(pre_edge_major(PDv, PSv, W.Wt_v, W.fragA_v, W.frag16A_v, W.S_v), W.g_v, W.be_v, W.S_v)
    WV = contract_channels(Hv, W.Wb_v) + W.bb_v[C_]
    j1 = np.where(valid1, nbr[i, np.minimum(e1, 31)], i)
    rel = x[i] - x[j1]                                           # [2][4][64][3]
    coef = alpha * WV * ew1
    dx = (coef[..., None] * rel).sum((0, 1))                     # per-lane partial [64][3]
    return dx.sum(0) / 16.0                                      # wave reduction over heads and q

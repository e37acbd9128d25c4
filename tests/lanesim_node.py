"""Arithmetic model of the split-f16 node GEMMs (cbgbench_amd/csrc/node_mfma.hip: node_proj_kernel, node_qmlp_kernel,
node_stage_kernel) -- numpy, CPU only.

What is modelled is the ARITHMETIC, not the lane layout (the K permutation of those kernels cancels between the A and B operands):
an fp32 row of activations is multiplied by the power of two that puts its largest magnitude into [2^14, 2^15) (``row_pow2``),
split into hi = f16(v), lo = f16(v - hi); every weight column was scaled the same way at pack time (``col_pow2_inv``); the
product is hi*lo + lo*hi + hi*hi per K = 32 chunk on ``v_mfma_f32_16x16x32_f16`` (exact f16 products, fp32 accumulation) and the
accumulator is multiplied by the two inverse powers of two, then the bias is added with one fma.  ``scaled=False`` gives the
arithmetic of the previous round (no scaling): kept to show the hole the scaling closes (tests/test_splitf16_range.py).
"""
import numpy as np


def _biased_exp(mx):
    return (np.asarray(mx, np.float32).view(np.uint32) >> 23) & 0xff


def row_pow2(mx):
    """exponent ka of the per-row up-scale 2^ka (node_mfma.hip row_pow2): mx 2^ka in [2^14, 2^15), clamped to +-100."""
    return np.clip(141 - _biased_exp(mx).astype(np.int64), -100, 100)


def col_pow2(mx):
    """exponent kc of the per-column weight scale (pack_colscale_kernel): clamped to [-100, 60]."""
    return np.clip(141 - _biased_exp(mx).astype(np.int64), -100, 60)


def split_f16(v):
    with np.errstate(over="ignore", invalid="ignore"):
        hi = v.astype(np.float16)
        lo = (v.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def split_gemm(h, W, bias=None, scaled=True):
    """h [N,K] fp32, W [K,M] fp32 (K a multiple of 32) -> [N,M] fp32, the kernels' arithmetic."""
    h, W = np.asarray(h, np.float32), np.asarray(W, np.float32)
    N, K = h.shape
    if scaled:
        ka = row_pow2(np.abs(h).max(1))
        kc = col_pow2(np.abs(W).max(0))
    else:
        ka, kc = np.zeros(N, np.int64), np.zeros(W.shape[1], np.int64)
    hs = np.ldexp(h, ka[:, None].astype(np.int32)).astype(np.float32)
    Ws = np.ldexp(W, kc[None, :].astype(np.int32)).astype(np.float32)
    ah, al = split_f16(hs)
    bh, bl = split_f16(Ws)
    acc = np.zeros((N, W.shape[1]), np.float32)
    f8 = lambda a: a.astype(np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        for u in range(K // 32):
            sl = slice(32 * u, 32 * u + 32)
            for a, b in ((ah, bl), (al, bh), (ah, bh)):
                acc = (acc.astype(np.float64) + f8(a[:, sl]) @ f8(b[sl])).astype(np.float32)
        rinv = np.ldexp(np.float32(1), (-ka).astype(np.int32)).astype(np.float32)
        cinv = np.ldexp(np.float32(1), (-kc).astype(np.int32)).astype(np.float32)
        out = (acc * rinv[:, None]).astype(np.float32)
        b = np.zeros(W.shape[1], np.float32) if bias is None else np.asarray(bias, np.float32)
        out = (out.astype(np.float64) * cinv[None, :] + b[None, :]).astype(np.float32)     # one fma
    return out


def fp32_chain(h, W, bias=None):
    """a plain fp32 FMA chain over k (what an fp32 GEMM does, up to the summation order): the yardstick."""
    h, W = np.asarray(h, np.float32), np.asarray(W, np.float32)
    acc = np.zeros((h.shape[0], W.shape[1]), np.float32) if bias is None else np.broadcast_to(
        np.asarray(bias, np.float32), (h.shape[0], W.shape[1])).copy()
    for k in range(h.shape[1]):
        acc = (acc.astype(np.float64) + h[:, k:k + 1].astype(np.float64) * W[k:k + 1].astype(np.float64)).astype(np.float32)
    return acc

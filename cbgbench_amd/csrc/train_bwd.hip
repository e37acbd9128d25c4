// libcbgx -- backward kernels of the denoiser (training).
//
// First-generation (VALU, LDS-staged) kernels: the backward of one attention block recomputes the forward
// of a node's 32 incoming edges in LDS (nothing per-edge is ever stored in HBM by the forward) and produces
//   * gradients of the node projection  dP [N,640]  (own part written, neighbour part by fp32 atomics),
//   * the per-node folds  T_i[a][m] = sum_e ds[e][a] hid_k[e][m]   and   S_i[a][m] = sum_e alpha e_w hid_v[e][m]
//     from which the second-Linear gradients follow at node level (the same folding the forward uses),
//   * coordinate gradients (atomics), gate gradients de_w, and per-workgroup partial sums of the
//     edge-indexed weight gradients (rbf / type columns of the first Linear, LayerNorm affine).
// Node-level kernels then finish the q MLP, the second Linears and the dense projections.
//
// Math follows the reference modules (autograd of x2h_attention.py:43-97, h2x_attention.py:34-73,
// common.py:151-171); oracle/training.py + torch.autograd is the checker (tests/test_gpu_training.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

__constant__ float c_mu_b[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

constexpr int EP = H + 4;  // LDS row pitch (floats): conflict-free for both row- and column-wise walks

struct EdgeBwdLds {
    float N[2][KNN][EP];    // pre-activation -> normalised -> dpre        (path 0 = k, 1 = v)
    float U[2][KNN][EP];    // hidden (post ReLU) -> d(normalised)
    float Qt[HEADS][EP];    // folded query of the node
    float Gt[HEADS][EP];    // x2h: folded output gradient; h2x: second v Linear [16][128]
    float rbf[KNN][G];
    float rel[KNN][4];      // x_i - x_j, |.|
    float ew[KNN];
    float sc[KNN][HEADS];   // scores -> alpha
    float gv[KNN][HEADS];   // x2h: G_i,a . vraw_e,a ; h2x: raw per-head value
    float ds[KNN][HEADS];
    float w2[KNN][HEADS];   // x2h: alpha e_w ; h2x: d(raw value)
    float stat[2][KNN][2];  // mean, rstd
    float stat2[2][KNN][2]; // mean(dn), mean(dn n)
    float me[KNN];          // h2x: sum_a alpha w / 16
    float D[4];             // h2x: gen_i * dL/dx_out_i
    int nb[KNN];
    int ty[KNN];
};

template <bool X2H>
__global__ __launch_bounds__(256) void edge_backward_kernel(
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ P,
    const float* __restrict__ Qt, const float* __restrict__ Gt, const float* __restrict__ gb,
    const float* __restrict__ gx_out, const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg,
    const uint8_t* __restrict__ lig, const float* __restrict__ e_w, const int* __restrict__ rows,
    const int* __restrict__ n_rows_ptr, int n_nodes, float* __restrict__ T, float* __restrict__ S,
    float* __restrict__ sw, float* __restrict__ dP, float* __restrict__ dx, float* __restrict__ de_w,
    float* __restrict__ partial) {
    __shared__ EdgeBwdLds L;
    const int t = threadIdx.x;
    const int p = t >> 7, m = t & 127, c = t;   // path, hidden channel, column of the k|v pair
    const float gamma = att[(p == 0 ? A_LNK_G : A_LNV_G) + m];
    const float beta = att[(p == 0 ? A_LNK_B : A_LNV_B) + m];

    float aWr[NT][G];
    float aWt[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        aWt[a] = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) aWr[a][g] = 0.f;
    }
    float aG = 0.f, aB = 0.f, aBb = 0.f;
    float aV16[HEADS];
#pragma unroll
    for (int a = 0; a < HEADS; ++a) aV16[a] = 0.f;

    if (!X2H) {
        for (int u = t; u < HEADS * H; u += 256) L.Gt[u >> 7][u & 127] = att[A_WBV + u];
    }
    const int count = rows ? *n_rows_ptr : n_nodes;
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const int i = rows ? rows[it] : it;
        const int d = deg[i];
        const int lig_i = lig[i];
        __syncthreads();
        if (t < KNN) {
            const int j = t < d ? nbr[(size_t)i * KNN + t] : -1;
            L.nb[t] = j;
            if (j >= 0) {
                const float rx = x[3 * i] - x[3 * j], ry = x[3 * i + 1] - x[3 * j + 1], rz = x[3 * i + 2] - x[3 * j + 2];
                const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
                L.rel[t][0] = rx; L.rel[t][1] = ry; L.rel[t][2] = rz; L.rel[t][3] = dist;
#pragma unroll
                for (int g = 0; g < G; ++g) { const float u = dist - c_mu_b[g]; L.rbf[t][g] = expf(-0.5f * (u * u)); }
                const int lig_j = lig[j];
                L.ty[t] = lig_j ? (lig_i ? 0 : 1) : (lig_i ? 2 : 3);
                L.ew[t] = e_w[(size_t)i * KNN + t];
            }
        }
        for (int u = t; u < HEADS * H; u += 256) {
            L.Qt[u >> 7][u & 127] = Qt[(size_t)i * HEADS * H + u];
            if (X2H) L.Gt[u >> 7][u & 127] = Gt[(size_t)i * HEADS * H + u];
        }
        if (!X2H && t < 3) L.D[t] = gx_out[3 * i + t];
        __syncthreads();

        {   // 1. pre-activations of the first Linear (factored form)
            const float pd = P[(size_t)i * PROW + c];
            for (int e = 0; e < d; ++e) {
                const int j = L.nb[e], ty = L.ty[e];
                float v = pd + P[(size_t)j * PROW + 2 * H + c] + att[A_WT + ty * 2 * H + c];
                const float* wr = att + A_WR + (size_t)ty * G * 2 * H + c;
#pragma unroll
                for (int g = 0; g < G; ++g) v = fmaf(wr[g * 2 * H], L.rbf[e][g], v);
                L.N[p][e][m] = v;
            }
        }
        __syncthreads();
        {   // 2. LayerNorm statistics: 4 threads per (path, edge)
            const int pe = t >> 2, part = t & 3, pp = pe >> 5, e = pe & 31;
            float s = 0.f;
            if (e < d) for (int u = 0; u < 32; ++u) s += L.N[pp][e][part + 4 * u];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            const float mean = s * (1.f / H);
            float q = 0.f;
            if (e < d) for (int u = 0; u < 32; ++u) { const float a = L.N[pp][e][part + 4 * u] - mean; q += a * a; }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
            if (part == 0) { L.stat[pp][e][0] = mean; L.stat[pp][e][1] = 1.f / sqrtf(q * (1.f / H) + 1e-5f); }
        }
        __syncthreads();
        for (int e = 0; e < d; ++e) {   // 3. normalise, affine, ReLU
            const float n = (L.N[p][e][m] - L.stat[p][e][0]) * L.stat[p][e][1];
            L.N[p][e][m] = n;
            L.U[p][e][m] = fmaxf(n * gamma + beta, 0.f);
        }
        __syncthreads();
        for (int pr = t; pr < KNN * HEADS; pr += 256) {   // 4. scores and per-head values
            const int e = pr >> 4, a = pr & 15;
            if (e < d) {
                float s = 0.f, v = 0.f;
                for (int k = 0; k < H; ++k) {
                    s = fmaf(L.Qt[a][k], L.U[0][e][k], s);
                    v = fmaf(L.Gt[a][k], L.U[1][e][k], v);
                }
                L.sc[e][a] = s;
                L.gv[e][a] = v + (X2H ? gb[(size_t)i * HEADS + a] : att[A_BBV + a]);
            }
        }
        __syncthreads();
        if (t < HEADS) {   // 5. softmax over the incoming edges of head t and its backward
            const int a = t;
            float mx = -INFINITY;
            for (int e = 0; e < d; ++e) mx = fmaxf(mx, L.sc[e][a]);
            float den = 0.f;
            for (int e = 0; e < d; ++e) { const float ex = expf(L.sc[e][a] - mx); L.sc[e][a] = ex; den += ex; }
            float cacc = 0.f;
            for (int e = 0; e < d; ++e) {
                const float al = L.sc[e][a] / den;
                L.sc[e][a] = al;
                float dal;
                if (X2H) {
                    dal = L.ew[e] * L.gv[e][a];
                } else {
                    const float rho = L.D[0] * L.rel[e][0] + L.D[1] * L.rel[e][1] + L.D[2] * L.rel[e][2];
                    dal = rho * L.gv[e][a] * L.ew[e] * (1.f / HEADS);
                }
                L.ds[e][a] = dal;
                cacc = fmaf(al, dal, cacc);
            }
            float swacc = 0.f;
            for (int e = 0; e < d; ++e) {
                const float al = L.sc[e][a];
                L.ds[e][a] = al * (L.ds[e][a] - cacc);
                if (X2H) {
                    const float w = al * L.ew[e];
                    L.w2[e][a] = w;
                    swacc += w;
                } else {
                    const float rho = L.D[0] * L.rel[e][0] + L.D[1] * L.rel[e][1] + L.D[2] * L.rel[e][2];
                    const float dv = rho * al * (1.f / HEADS) * L.ew[e];   // d(raw value)
                    L.w2[e][a] = dv;
                    swacc += dv;
                }
            }
            if (X2H) sw[(size_t)i * HEADS + a] = swacc; else aBb += swacc;
        }
        __syncthreads();
        if (t < KNN && t < d) {   // gate gradient and (h2x) the scalar coefficient of rel in delta_x
            const int e = t;
            float de = 0.f, me = 0.f;
            if (X2H) {
                for (int a = 0; a < HEADS; ++a) de = fmaf(L.sc[e][a], L.gv[e][a], de);
            } else {
                const float rho = L.D[0] * L.rel[e][0] + L.D[1] * L.rel[e][1] + L.D[2] * L.rel[e][2];
                float av = 0.f;
                for (int a = 0; a < HEADS; ++a) av = fmaf(L.sc[e][a], L.gv[e][a], av);
                de = rho * av * (1.f / HEADS);
                me = av * L.ew[e] * (1.f / HEADS);
            }
            de_w[(size_t)i * KNN + e] += de;
            L.me[e] = me;
        }
        {   // 6. folds over the edges: T (k path), S / second-Linear gradient (v path)
            float acc[HEADS];
#pragma unroll
            for (int a = 0; a < HEADS; ++a) acc[a] = 0.f;
            for (int e = 0; e < d; ++e) {
                const float u = L.U[p][e][m];
#pragma unroll
                for (int a = 0; a < HEADS; ++a) acc[a] = fmaf(p == 0 ? L.ds[e][a] : L.w2[e][a], u, acc[a]);
            }
            if (p == 0) {
#pragma unroll
                for (int a = 0; a < HEADS; ++a) T[((size_t)i * HEADS + a) * H + m] = acc[a];
            } else if (X2H) {
#pragma unroll
                for (int a = 0; a < HEADS; ++a) S[((size_t)i * HEADS + a) * H + m] = acc[a];
            } else {
#pragma unroll
                for (int a = 0; a < HEADS; ++a) aV16[a] += acc[a];
            }
        }
        {   // 7. gradient of the hidden activations -> gradient of the normalised pre-activations
            float qc[HEADS];
#pragma unroll
            for (int a = 0; a < HEADS; ++a) qc[a] = p == 0 ? L.Qt[a][m] : L.Gt[a][m];
            for (int e = 0; e < d; ++e) {
                float du = 0.f;
#pragma unroll
                for (int a = 0; a < HEADS; ++a) du = fmaf(p == 0 ? L.ds[e][a] : L.w2[e][a], qc[a], du);
                const float dy = L.U[p][e][m] > 0.f ? du : 0.f;
                aG = fmaf(dy, L.N[p][e][m], aG);
                aB += dy;
                L.U[p][e][m] = dy * gamma;
            }
        }
        __syncthreads();
        {   // 8. LayerNorm backward statistics
            const int pe = t >> 2, part = t & 3, pp = pe >> 5, e = pe & 31;
            float s1 = 0.f, s2 = 0.f;
            if (e < d) for (int u = 0; u < 32; ++u) {
                const float dn = L.U[pp][e][part + 4 * u];
                s1 += dn;
                s2 = fmaf(dn, L.N[pp][e][part + 4 * u], s2);
            }
            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
            if (part == 0) { L.stat2[pp][e][0] = s1 * (1.f / H); L.stat2[pp][e][1] = s2 * (1.f / H); }
        }
        __syncthreads();
        {   // 9. dpre; its sums: own projection row, neighbour rows (atomics), type / rbf columns of the first Linear
            float accpd = 0.f;
            for (int e = 0; e < d; ++e) {
                const float n = L.N[p][e][m];
                const float dp = L.stat[p][e][1] * (L.U[p][e][m] - L.stat2[p][e][0] - n * L.stat2[p][e][1]);
                L.N[p][e][m] = dp;
                accpd += dp;
                atomicAdd(&dP[(size_t)L.nb[e] * PROW + 2 * H + c], dp);
                const float* r = L.rbf[e];
                switch (L.ty[e]) {
#define CBGX_ACC_TYPE(TY)                                                        \
    case TY:                                                                     \
        aWt[TY] += dp;                                                           \
        _Pragma("unroll") for (int g = 0; g < G; ++g) aWr[TY][g] = fmaf(r[g], dp, aWr[TY][g]); \
        break;
                    CBGX_ACC_TYPE(0)
                    CBGX_ACC_TYPE(1)
                    CBGX_ACC_TYPE(2)
                    CBGX_ACC_TYPE(3)
#undef CBGX_ACC_TYPE
                }
            }
            dP[(size_t)i * PROW + c] = accpd;
        }
        __syncthreads();
        {   // 10. through the radial basis to the coordinates: 8 threads per edge
            const int e = t >> 3, part = t & 7;
            float acc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = 0.f;
            if (e < d) {
                const float* wr = att + A_WR + (size_t)L.ty[e] * G * 2 * H;
                for (int k = 0; k < 32; ++k) {
                    const int cc = part + 8 * k;
                    const float dpv = L.N[cc >> 7][e][cc & 127];
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = fmaf(wr[g * 2 * H + cc], dpv, acc[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[g] += __shfl_xor(acc[g], 1, 64);
                acc[g] += __shfl_xor(acc[g], 2, 64);
                acc[g] += __shfl_xor(acc[g], 4, 64);
            }
            if (part == 0 && e < d) {
                const float dist = L.rel[e][3];
                float dd = 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dd = fmaf(acc[g], -(dist - c_mu_b[g]) * L.rbf[e][g], dd);
                const float coef = dist > 0.f ? dd / dist : 0.f;
                const int j = L.nb[e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float g3 = coef * L.rel[e][k];
                    if (!X2H) g3 = fmaf(L.me[e], L.D[k], g3);
                    atomicAdd(&dx[3 * i + k], g3);
                    atomicAdd(&dx[3 * j + k], -g3);
                }
            }
        }
    }
    // per-workgroup partial sums of the edge-indexed weight gradients
    float* slab = partial + (size_t)blockIdx.x * PB_SIZE;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        slab[PB_WT + a * 2 * H + c] = aWt[a];
#pragma unroll
        for (int g = 0; g < G; ++g) slab[PB_WR + (a * G + g) * 2 * H + c] = aWr[a][g];
    }
    slab[PB_LNG + c] = aG;
    slab[PB_LNB + c] = aB;
    if (!X2H) {
        if (p == 1) {
#pragma unroll
            for (int a = 0; a < HEADS; ++a) slab[PB_WBV16 + a * H + m] = aV16[a];
        }
        if (t < HEADS) slab[PB_BBV16 + t] = aBb;
    }
}

// ------------------------------------------------------------------------------------------------
// x2h: fold the output gradient through the second v Linear (the mirror image of the query fold):
//   Gt[i][a][m] = sum_c G[i][8a+c] Wbv[8a+c][m],   gb[i][a] = sum_c G[i][8a+c] bbv[8a+c]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void fold_grad_kernel(const float* __restrict__ att, const float* __restrict__ Gr,
                                                        int n_nodes, float* __restrict__ Gt, float* __restrict__ gb) {
    __shared__ float sG[16][H];
    const int row0 = blockIdx.x * 16, m = threadIdx.x;
    for (int u = m; u < 16 * H; u += 128) {
        const int r = u >> 7;
        sG[r][u & 127] = row0 + r < n_nodes ? Gr[(size_t)(row0 + r) * H + (u & 127)] : 0.f;
    }
    __syncthreads();
    for (int a = 0; a < HEADS; ++a) {
        float w[DH];
#pragma unroll
        for (int cc = 0; cc < DH; ++cc) w[cc] = att[A_WBV + (size_t)m * H + a * DH + cc];   // x2h layout [m][n]
        for (int r = 0; r < 16 && row0 + r < n_nodes; ++r) {
            float s = 0.f;
#pragma unroll
            for (int cc = 0; cc < DH; ++cc) s = fmaf(sG[r][a * DH + cc], w[cc], s);
            Gt[((size_t)(row0 + r) * HEADS + a) * H + m] = s;
        }
    }
    if (m < HEADS) {
        for (int r = 0; r < 16 && row0 + r < n_nodes; ++r) {
            float s = 0.f;
#pragma unroll
            for (int cc = 0; cc < DH; ++cc) s = fmaf(sG[r][m * DH + cc], att[A_BBV + m * DH + cc], s);
            gb[(size_t)(row0 + r) * HEADS + m] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// query path backward, one node at a time per 128-thread workgroup (persistent):
//   recompute z = ReLU(LN(P[:,512:640])), q = Wq1 z + bq1;  dq = (1/sqrt 8) Wbk_a^T-fold of T;
//   outputs qs = q/sqrt(8), dq, z (for the outer-product weight gradients) and dP[:,512:640].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void q_backward_kernel(const float* __restrict__ att, const float* __restrict__ P,
                                                         const float* __restrict__ T, const int* __restrict__ rows,
                                                         const int* __restrict__ n_rows_ptr, int n_nodes,
                                                         float* __restrict__ qs, float* __restrict__ dqb,
                                                         float* __restrict__ zb, float* __restrict__ dP,
                                                         float* __restrict__ partial) {
    // 16 nodes per tile, 256 threads.  Phases: LayerNorm (16 threads per row) -> q (thread = column, 8 rows) ->
    // dq (T streamed through LDS in chunks of 32 m) -> dz -> LayerNorm backward.
    __shared__ float sZ[H][17];            // hidden (post ReLU), transposed [k][row]
    __shared__ float sNq[16][H + 1];       // normalised pre-activation
    __shared__ float sDq[16][H + 1];       // dq, then d(normalised)
    __shared__ float sT[16][HEADS][17];    // chunk of the per-node fold T: [row][head][16 m]
    __shared__ float sRstd[16];
    __shared__ int sRow[16];
    const int t = threadIdx.x;
    const int n = t & 127, half = t >> 7;
    const float s8 = 0.35355339059327376220f;
    const float gq = att[A_LNQ_G + n], b1 = att[A_BQ1 + n];
    float aG = 0.f, aB = 0.f;
    const int count = rows ? *n_rows_ptr : n_nodes;
    const int tiles = (count + 15) / 16;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        if (t < 16) { const int it = tile * 16 + t; sRow[t] = it < count ? (rows ? rows[it] : it) : -1; }
        __syncthreads();
        {
            const int r = t >> 4, part = t & 15, i = sRow[r];
            float v[8], s = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = i >= 0 ? P[(size_t)i * PROW + 4 * H + part + 16 * u] : 0.f; s += v[u]; }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            const float mean = s * (1.f / H);
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) q += (v[u] - mean) * (v[u] - mean);
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
            const float rstd = 1.f / sqrtf(q * (1.f / H) + 1e-5f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = part + 16 * u;
                const float nq = (v[u] - mean) * rstd;
                const float z = fmaxf(nq * att[A_LNQ_G + k] + att[A_LNQ_B + k], 0.f);
                sNq[r][k] = nq;
                sZ[k][r] = z;
                if (i >= 0) zb[(size_t)i * H + k] = z;
            }
            if (part == 0) sRstd[r] = rstd;
        }
        __syncthreads();
        {   // q = Wq1 z + bq1 (stored pre-scaled by 1/sqrt(8) for the Wbk gradient)
            float acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = b1;
#pragma unroll 16
            for (int k = 0; k < H; ++k) {
                const float w = att[A_WQ1T + (size_t)k * H + n];
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = fmaf(sZ[k][half * 8 + r], w, acc[r]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int i = sRow[half * 8 + r]; if (i >= 0) qs[(size_t)i * H + n] = acc[r] * s8; }
        }
        {   // dq[n] = (1/sqrt 8) sum_m Wbk[n][m] T[n >> 3][m]
            float acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.f;
            const int a = n >> 3;
            for (int m0 = 0; m0 < H; m0 += 16) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int idx = t + 256 * j, mm = idx & 15, aa = (idx >> 4) & 15, rr = idx >> 8;
                    const int i = sRow[rr];
                    sT[rr][aa][mm] = i >= 0 ? T[((size_t)i * HEADS + aa) * H + m0 + mm] : 0.f;
                }
                __syncthreads();
#pragma unroll
                for (int mm = 0; mm < 16; ++mm) {
                    const float w = att[A_WBKT + (size_t)(m0 + mm) * H + n];
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = fmaf(w, sT[half * 8 + r][a][mm], acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float dq = acc[r] * s8;
                sDq[half * 8 + r][n] = dq;
                const int i = sRow[half * 8 + r];
                if (i >= 0) dqb[(size_t)i * H + n] = dq;
            }
        }
        __syncthreads();
        {   // dz[k] = sum_n dq[n] Wq1[n][k]  ->  d(normalised)
            float acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll 16
            for (int nn = 0; nn < H; ++nn) {
                const float w = att[A_WQ1O + (size_t)nn * H + n];
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = fmaf(sDq[half * 8 + r][nn], w, acc[r]);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = half * 8 + r;
                const float dy = sZ[n][rr] > 0.f ? acc[r] : 0.f;
                aG = fmaf(dy, sNq[rr][n], aG);
                aB += dy;
                sDq[rr][n] = dy * gq;
            }
        }
        __syncthreads();
        {
            const int r = t >> 4, part = t & 15, i = sRow[r];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dn = sDq[r][part + 16 * u];
                s1 += dn;
                s2 = fmaf(dn, sNq[r][part + 16 * u], s2);
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
            const float m1 = s1 * (1.f / H), m2 = s2 * (1.f / H), rstd = sRstd[r];
            if (i >= 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = part + 16 * u;
                    dP[(size_t)i * PROW + 4 * H + k] = rstd * (sDq[r][k] - m1 - sNq[r][k] * m2);
                }
            }
        }
    }
    // LayerNorm affine gradients: [gamma | beta], accumulated over workgroups (partial must be zeroed by the caller)
    atomicAdd(&partial[n], aG);
    atomicAdd(&partial[H + n], aB);
}

// ------------------------------------------------------------------------------------------------
// outer-product accumulation over nodes:  dW[n][m] = sum_i L[i][n] * R[i][(HEADED ? n>>3 : 0)][m]
// (HEADED: R is [N,16,128]; otherwise [N,128]).  One 128x128 partial per workgroup.
// ------------------------------------------------------------------------------------------------
template <bool HEADED>
__global__ __launch_bounds__(256) void outer_accum_kernel(const float* __restrict__ Lm, const float* __restrict__ R,
                                                          const int* __restrict__ rows, const int* __restrict__ n_rows_ptr,
                                                          int n_nodes, float* __restrict__ partial, size_t slab_stride) {
    __shared__ float sL[4][H];
    __shared__ int sI[4];
    const int t = threadIdx.x, m = t & 127, nh = t >> 7;
    float acc[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) acc[k] = 0.f;
    const int count = rows ? *n_rows_ptr : n_nodes;
    for (int base = blockIdx.x * 4; base < count; base += gridDim.x * 4) {
        __syncthreads();
        if (t < 4) { const int it = base + t; sI[t] = it < count ? (rows ? rows[it] : it) : -1; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = t + 256 * j, qn = idx >> 7, i = sI[qn];
            sL[qn][idx & 127] = i >= 0 ? Lm[(size_t)i * H + (idx & 127)] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) {
            const int i = sI[qn];
            if (i < 0) continue;
            if (HEADED) {
                float r[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) r[a] = R[((size_t)i * HEADS + nh * 8 + a) * H + m];
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int cc = 0; cc < DH; ++cc)
                        acc[a * DH + cc] = fmaf(sL[qn][nh * 64 + a * DH + cc], r[a], acc[a * DH + cc]);
            } else {
                const float r = R[(size_t)i * H + m];
#pragma unroll
                for (int k = 0; k < 64; ++k) acc[k] = fmaf(sL[qn][nh * 64 + k], r, acc[k]);
            }
        }
    }
    float* slab = partial + (size_t)blockIdx.x * slab_stride;
#pragma unroll
    for (int k = 0; k < 64; ++k) slab[(size_t)(nh * 64 + k) * H + m] = acc[k];
}

// column sums of A[rows, cols] (optionally of A[i][c] * scale[i][c >> 3]) -> one partial row per workgroup
template <bool LISTED, bool SCALED>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int lda, int cols,
                                                     const float* __restrict__ scale, const int* __restrict__ rows,
                                                     const int* __restrict__ n_rows_ptr, int n_rows,
                                                     float* __restrict__ partial, size_t slab_stride) {
    const int count = LISTED ? *n_rows_ptr : n_rows;
    for (int c0 = threadIdx.x; c0 < cols; c0 += 256) {
        float acc = 0.f;
        // eight rows per pass, loads unconditional (rows past the end read row 0 and are weighted by zero): predicated
        // loads compile to one branch + wait per element and leave the kernel latency bound
        for (int it = blockIdx.x; it < count; it += 8 * gridDim.x) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int itj = it + j * gridDim.x;
                const bool ok = itj < count;
                const int itc = ok ? itj : 0;
                const int i = LISTED ? rows[itc] : itc;
                v[j] = A[(size_t)i * lda + c0];
                if (SCALED) v[j] *= scale[(size_t)i * HEADS + (c0 >> 3)];
                v[j] = ok ? v[j] : 0.f;
            }
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        partial[(size_t)blockIdx.x * slab_stride + c0] = acc;
    }
}

// first level of a two-level slab reduction: dst[g][c] = sum over slabs s = g (mod groups) of src[s][c]
__global__ __launch_bounds__(256) void slab_fold_kernel(const float* __restrict__ src, int n_slabs, size_t slab_stride,
                                                        int size, int groups, float* __restrict__ dst) {
    const int c = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (c >= size) return;
    float acc = 0.f;
    for (int s = g; s < n_slabs; s += groups) acc += src[(size_t)s * slab_stride + c];
    dst[(size_t)g * size + c] = acc;
}

// dst[r][c] (or dst[c][r] if transpose) = sum_s src[s * slab_stride + r * src_ld + c]
__global__ void reduce_store_kernel(const float* __restrict__ src, int n_slabs, size_t slab_stride, int src_ld,
                                    int rows, int cols, float* __restrict__ dst, int dst_ld, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx % cols;
    float acc = 0.f;
    for (int s = 0; s < n_slabs; ++s) acc += src[(size_t)s * slab_stride + (size_t)r * src_ld + c];
    if (transpose) dst[(size_t)c * dst_ld + r] = acc; else dst[(size_t)r * dst_ld + c] = acc;
}

// several reduce_store pieces in one launch (blockIdx.y = piece)
__global__ void reduce_store_multi_kernel(RsBatch b) {
    const RsPiece& pc = b.p[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pc.rows * pc.cols) return;
    const int r = idx / pc.cols, c = idx % pc.cols;
    float acc = 0.f;
    for (int s = 0; s < pc.n_slabs; ++s) acc += pc.src[(size_t)s * pc.stride + (size_t)r * pc.src_ld + c];
    if (pc.transpose) pc.dst[(size_t)c * pc.dst_ld + r] = acc; else pc.dst[(size_t)r * pc.dst_ld + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// generic fp32 GEMM for the node-level products of the backward (sizes are small: N_nodes x 640 x 128):
//   C[z] (+)= op(A) op(B) over the K range of split z.  64x64 tile, 256 threads, 4x4 per thread.
// ------------------------------------------------------------------------------------------------
typedef float floatx4_t __attribute__((ext_vector_type(4)));

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                    int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                    int k_chunk, size_t c_split_stride, int accumulate) {
    // 64 x 64 output tile per workgroup, K staged 16 at a time through LDS; wave (wr, wc) of the 2 x 2 wave grid owns a
    // 32 x 32 sub-tile = 2 x 2 MFMA tiles (v_mfma_f32_16x16x4_f32: lane (li, kq) feeds A[row li][k kq], B[k kq][col li]).
    __shared__ float sA[16][68];
    __shared__ float sB[16][68];
    const int bm = blockIdx.y * 64, bn = blockIdx.x * 64, z = blockIdx.z;
    const int k0 = z * k_chunk, k1 = min(K, k0 + k_chunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    floatx4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (floatx4_t){0.f, 0.f, 0.f, 0.f};
    for (int kb = k0; kb < k1; kb += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = threadIdx.x + 256 * r;
            {
                int row, kk;
                if (!TA) { row = idx >> 4; kk = idx & 15; } else { kk = idx >> 6; row = idx & 63; }
                const bool ok = bm + row < M && kb + kk < k1;
                float v = 0.f;
                if (ok) v = TA ? A[(size_t)(kb + kk) * lda + bm + row] : A[(size_t)(bm + row) * lda + kb + kk];
                sA[kk][row] = v;
            }
            {
                int col, kk;
                if (!TB) { kk = idx >> 6; col = idx & 63; } else { col = idx >> 4; kk = idx & 15; }
                const bool ok = bn + col < N && kb + kk < k1;
                float v = 0.f;
                if (ok) v = TB ? B[(size_t)(bn + col) * ldb + kb + kk] : B[(size_t)(kb + kk) * ldb + bn + col];
                sB[kk][col] = v;
            }
        }
        __syncthreads();
        float a[4][2], b[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[kk][i] = sA[4 * kk + kq][32 * wr + 16 * i + li];
                b[kk][i] = sB[4 * kk + kq][32 * wc + 16 * i + li];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
    float* Cz = C + (size_t)z * c_split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + 32 * wr + 16 * i + 4 * kq + r, col = bn + 32 * wc + 16 * j + li;
                if (row < M && col < N) {
                    float* o = Cz + (size_t)row * ldc + col;
                    *o = accumulate ? *o + acc[i][j][r] : acc[i][j][r];
                }
            }
}

// ------------------------------------------------------------------------------------------------
// distance gate backward (unitransformer.py:109-112): pass 1 per edge (scalars of the LayerNorm
// backward), pass 2 per hidden unit (weight gradients).  The gate reads the *input* coordinates, which
// are data, so no coordinate gradient is produced.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_bwd_edge_kernel(const float* __restrict__ wts, const float* __restrict__ x,
                                                            const int32_t* __restrict__ nbr,
                                                            const int32_t* __restrict__ deg, int n_nodes,
                                                            const float* __restrict__ de_w, float* __restrict__ E8) {
    __shared__ float sW1[GH * G];
    __shared__ float sB1[GH], sG[GH], sBe[GH], sW2[GH];
    for (int t = threadIdx.x; t < GH * G; t += blockDim.x) sW1[t] = wts[GATE_W1 + t];
    for (int t = threadIdx.x; t < GH; t += blockDim.x) {
        sB1[t] = wts[GATE_B1 + t]; sG[t] = wts[GATE_LNG + t]; sBe[t] = wts[GATE_LNB + t]; sW2[t] = wts[GATE_W2 + t];
    }
    __syncthreads();
    const float b2 = wts[GATE_B2];
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n_nodes * KNN) return;
    const int i = (int)(e >> 5), s = (int)(e & 31);
    float* o = E8 + (size_t)e * 8;
    if (s >= deg[i]) { o[5] = 0.f; o[0] = 0.f; o[1] = 0.f; o[2] = 1.f; o[3] = 0.f; o[4] = 0.f; return; }
    const int j = nbr[e];
    const float dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    float r[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { const float t = dist - c_mu_b[g]; r[g] = expf(-0.5f * (t * t)); }
    float sum = 0.f;
    for (int u = 0; u < GH; ++u) {
        float y = sB1[u];
#pragma unroll
        for (int g = 0; g < G; ++g) y = fmaf(sW1[u * G + g], r[g], y);
        sum += y;
    }
    const float mean = sum * (1.f / GH);
    float var = 0.f;
    for (int u = 0; u < GH; ++u) {
        float y = sB1[u];
#pragma unroll
        for (int g = 0; g < G; ++g) y = fmaf(sW1[u * G + g], r[g], y);
        var += (y - mean) * (y - mean);
    }
    const float rstd = 1.f / sqrtf(var * (1.f / GH) + 1e-5f);
    float acc = b2;
    for (int u = 0; u < GH; ++u) {
        float y = sB1[u];
#pragma unroll
        for (int g = 0; g < G; ++g) y = fmaf(sW1[u * G + g], r[g], y);
        acc = fmaf(sW2[u], fmaxf((y - mean) * rstd * sG[u] + sBe[u], 0.f), acc);
    }
    const float ew = 1.f / (1.f + expf(-acc));
    const float dacc = de_w[e] * ew * (1.f - ew);
    float s1 = 0.f, s2 = 0.f;
    for (int u = 0; u < GH; ++u) {
        float y = sB1[u];
#pragma unroll
        for (int g = 0; g < G; ++g) y = fmaf(sW1[u * G + g], r[g], y);
        const float n = (y - mean) * rstd;
        const float dn = (n * sG[u] + sBe[u] > 0.f) ? dacc * sW2[u] * sG[u] : 0.f;
        s1 += dn;
        s2 = fmaf(dn, n, s2);
    }
    o[0] = dist; o[1] = mean; o[2] = rstd; o[3] = s1 * (1.f / GH); o[4] = s2 * (1.f / GH); o[5] = dacc;
}

constexpr int GATE_TILE = 64;

__global__ __launch_bounds__(GH) void gate_bwd_weight_kernel(const float* __restrict__ wts, const float* __restrict__ E8,
                                                             long n_edges, float* __restrict__ partial) {
    __shared__ float sR[GATE_TILE][G];
    __shared__ float sE[GATE_TILE][8];
    const int u = threadIdx.x;
    float w1[G], aW1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { w1[g] = wts[GATE_W1 + u * G + g]; aW1[g] = 0.f; }
    const float b1 = wts[GATE_B1 + u], gam = wts[GATE_LNG + u], bet = wts[GATE_LNB + u], w2 = wts[GATE_W2 + u];
    float aB1 = 0.f, aG = 0.f, aBe = 0.f, aW2 = 0.f, aB2 = 0.f;
    const long tiles = (n_edges + GATE_TILE - 1) / GATE_TILE;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long e0 = tile * GATE_TILE;
        __syncthreads();
        for (int k = u; k < GATE_TILE * 8; k += GH) {
            const long e = e0 + (k >> 3);
            sE[k >> 3][k & 7] = e < n_edges ? E8[(size_t)e * 8 + (k & 7)] : 0.f;
        }
        __syncthreads();
        for (int k = u; k < GATE_TILE * G; k += GH) {
            const int ee = k / G, g = k % G;
            const float t = sE[ee][0] - c_mu_b[g];
            sR[ee][g] = expf(-0.5f * (t * t));
        }
        __syncthreads();
        for (int ee = 0; ee < GATE_TILE; ++ee) {
            const float dacc = sE[ee][5];
            if (dacc == 0.f) continue;   // padded slot or zero upstream gradient (uniform across the workgroup)
            float y = b1;
#pragma unroll
            for (int g = 0; g < G; ++g) y = fmaf(w1[g], sR[ee][g], y);
            const float n = (y - sE[ee][1]) * sE[ee][2];
            const float ya = n * gam + bet;
            aW2 = fmaf(dacc, fmaxf(ya, 0.f), aW2);
            aB2 += dacc;
            const float dy = ya > 0.f ? dacc * w2 : 0.f;
            aG = fmaf(dy, n, aG);
            aBe += dy;
            const float dn = dy * gam;
            const float dp = sE[ee][2] * (dn - sE[ee][3] - n * sE[ee][4]);
            aB1 += dp;
#pragma unroll
            for (int g = 0; g < G; ++g) aW1[g] = fmaf(dp, sR[ee][g], aW1[g]);
        }
    }
    float* slab = partial + (size_t)blockIdx.x * GB_SIZE;
#pragma unroll
    for (int g = 0; g < G; ++g) slab[GB_W1 + u * G + g] = aW1[g];
    slab[GB_B1 + u] = aB1;
    slab[GB_LNG + u] = aG;
    slab[GB_LNB + u] = aBe;
    slab[GB_W2 + u] = aW2;
    if (u == 0) slab[GB_B2] = aB2;
}

// classifier: d(pre) = d(act) * sigmoid(pre)   (derivative of softplus(x) - ln 2)
__global__ void ssp_backward_kernel(const float* __restrict__ pre, const float* __restrict__ dact, long n,
                                    float* __restrict__ dpre) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dpre[idx] = dact[idx] / (1.f + expf(-pre[idx]));
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dst[idx] += src[idx];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define CBGX_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return _e;               \
    } while (0)

hipError_t launch_edge_backward(bool x2h, const float* att, const float* x, const float* P, const float* Qt,
                                const float* Gt, const float* gb, const float* gx_out, const int32_t* nbr,
                                const int32_t* deg, const uint8_t* lig, const float* e_w, const int* rows,
                                const int* n_rows, int n_nodes, float* T, float* S, float* sw, float* dP, float* dx,
                                float* de_w, float* partial, int grid, hipStream_t s) {
    profile_mark_begin(x2h ? (rows ? K_EDGE_X2H_BWD_LISTED : K_EDGE_X2H_BWD) : K_EDGE_H2X_BWD, s);
    if (x2h)
        hipLaunchKernelGGL(edge_backward_kernel<true>, dim3(grid), dim3(256), 0, s, att, x, P, Qt, Gt, gb, gx_out, nbr,
                           deg, lig, e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial);
    else
        hipLaunchKernelGGL(edge_backward_kernel<false>, dim3(grid), dim3(256), 0, s, att, x, P, Qt, Gt, gb, gx_out, nbr,
                           deg, lig, e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_fold_grad(const float* att, const float* Gr, int n_nodes, float* Gt, float* gb, hipStream_t s) {
    hipLaunchKernelGGL(fold_grad_kernel, dim3((n_nodes + 15) / 16), dim3(128), 0, s, att, Gr, n_nodes, Gt, gb);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_q_backward(const float* att, const float* P, const float* T, const int* rows, const int* n_rows,
                             int n_nodes, float* qs, float* dqb, float* zb, float* dP, float* partial, int grid,
                             hipStream_t s) {
    hipLaunchKernelGGL(q_backward_kernel, dim3(grid), dim3(256), 0, s, att, P, T, rows, n_rows, n_nodes, qs, dqb, zb, dP,
                       partial);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_outer_accum(bool headed, const float* Lm, const float* R, const int* rows, const int* n_rows,
                              int n_nodes, float* partial, size_t slab_stride, int grid, hipStream_t s) {
    if (headed)
        hipLaunchKernelGGL(outer_accum_kernel<true>, dim3(grid), dim3(256), 0, s, Lm, R, rows, n_rows, n_nodes, partial,
                           slab_stride);
    else
        hipLaunchKernelGGL(outer_accum_kernel<false>, dim3(grid), dim3(256), 0, s, Lm, R, rows, n_rows, n_nodes, partial,
                           slab_stride);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_colsum(const float* A, int lda, int cols, const float* scale, const int* rows, const int* n_rows_ptr,
                         int n_rows, float* partial, size_t slab_stride, int grid, hipStream_t s) {
#define CBGX_COLSUM(L, S)                                                                                             \
    hipLaunchKernelGGL((colsum_kernel<L, S>), dim3(grid), dim3(256), 0, s, A, lda, cols, scale, rows, n_rows_ptr, n_rows, \
                       partial, slab_stride)
    if (rows) { if (scale) CBGX_COLSUM(true, true); else CBGX_COLSUM(true, false); }
    else      { if (scale) CBGX_COLSUM(false, true); else CBGX_COLSUM(false, false); }
#undef CBGX_COLSUM
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_slab_fold(const float* src, int n_slabs, size_t slab_stride, int size, int groups, float* dst,
                            hipStream_t s) {
    hipLaunchKernelGGL(slab_fold_kernel, dim3((size + 255) / 256, groups), dim3(256), 0, s, src, n_slabs, slab_stride, size,
                       groups, dst);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_reduce_store(const float* src, int n_slabs, size_t slab_stride, int src_ld, int rows, int cols,
                               float* dst, int dst_ld, int transpose, hipStream_t s) {
    const int total = rows * cols;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_store_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, n_slabs, slab_stride, src_ld,
                       rows, cols, dst, dst_ld, transpose);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_reduce_store_multi(const RsBatch& b, hipStream_t s) {
    if (b.n == 0) return hipSuccess;
    int mx = 0;
    for (int k = 0; k < b.n; ++k) mx = b.p[k].rows * b.p[k].cols > mx ? b.p[k].rows * b.p[k].cols : mx;
    hipLaunchKernelGGL(reduce_store_multi_kernel, dim3((mx + 255) / 256, b.n), dim3(256), 0, s, b);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sgemm(bool ta, bool tb, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                        int N, int K, int splits, size_t c_split_stride, int accumulate, hipStream_t s) {
    if (M == 0 || N == 0) return hipSuccess;
    if (splits < 1) splits = 1;
    int k_chunk = (K + splits - 1) / splits;
    k_chunk = (k_chunk + 15) / 16 * 16;
    dim3 grid((N + 63) / 64, (M + 63) / 64, splits), block(256);
    profile_mark_begin(K_TRAIN_GEMM, s);
    if (!ta && !tb) hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk, c_split_stride, accumulate);
    else if (!ta && tb) hipLaunchKernelGGL((sgemm_kernel<false, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk, c_split_stride, accumulate);
    else if (ta && !tb) hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk, c_split_stride, accumulate);
    else hipLaunchKernelGGL((sgemm_kernel<true, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk, c_split_stride, accumulate);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gate_backward(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                                const float* de_w, float* E8, float* partial, int grid, hipStream_t s) {
    const long total = (long)n_nodes * KNN;
    hipLaunchKernelGGL(gate_bwd_edge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, packed, x, nbr, deg,
                       n_nodes, de_w, E8);
    CBGX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gate_bwd_weight_kernel, dim3(grid), dim3(GH), 0, s, packed, E8, total, partial);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_ssp_backward(const float* pre, const float* dact, long n, float* dpre, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ssp_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, dact, n, dpre);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_add_inplace(float* dst, const float* src, long n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cbgx

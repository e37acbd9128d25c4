// libcbgx -- FIRST-GENERATION backward kernels (VALU, LDS-staged): test-only, compiled into libcbgx_xcheck.so (-DCBGX_XCHECK).
//
// The backward of one attention block recomputes the forward of a node's 32 incoming edges in LDS and produces dP, the
// per-node folds T / S, coordinate and gate gradients and per-workgroup partial sums of the edge-indexed weight gradients,
// exactly like the MFMA generation (train_bwd_mfma.hip) that replaced it; cbgx_debug_set_edge_kernel(1) routes the
// backward entry points through these kernels as an independent on-device cross-check (tests/test_gpu_training.py).
// Math follows the reference modules (autograd of x2h_attention.py:43-97, h2x_attention.py:34-73, common.py:151-171).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

#define CBGX_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return _e;               \
    } while (0)

__constant__ float c_mu_b1[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
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
                for (int g = 0; g < G; ++g) { const float u = dist - c_mu_b1[g]; L.rbf[t][g] = expf(-0.5f * (u * u)); }
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
                for (int g = 0; g < G; ++g) dd = fmaf(acc[g], -(dist - c_mu_b1[g]) * L.rbf[e][g], dd);
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
}  // namespace cbgx

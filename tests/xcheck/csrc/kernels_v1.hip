// libcbgx -- first-generation gfx950 kernels (VALU, LDS-staged, one workgroup per destination node).
// These are correct-by-construction baselines of the fused/factored formulation; the MFMA kernels
// replace them stage by stage.  Wave = 64 lanes throughout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

__constant__ float c_rbf_mu[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                  3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

// ------------------------------------------------------------------------------------------------
// kNN graph: one wave per centre node; 32 rounds of "smallest key greater than the previous one",
// key = (bits(d2) << 32) | j, which orders by (squared distance, index) exactly like the oracle.
// d2 = ((dx*dx)+(dy*dy))+(dz*dz) with contraction off so it is bit-identical to the CPU oracle.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float s = dx * dx;
    float t = dy * dy;
    s = s + t;
    t = dz * dz;
    s = s + t;
    return s;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void knn_graph_kernel(const float* __restrict__ x,
                                                        const int32_t* __restrict__ graph_ptr, int n_graphs,
                                                        int n_nodes, int32_t* __restrict__ nbr,
                                                        int32_t* __restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_nodes) return;
    // binary search: largest g with graph_ptr[g] <= i
    int lo = 0, hi = n_graphs;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (graph_ptr[mid] <= i) lo = mid; else hi = mid;
    }
    const int gs = graph_ptr[lo], ge = graph_ptr[lo + 1];
    const int n = ge - gs;
    const int d = min(KNN, n - 1);
    const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    unsigned long long prev = 0ull;
    bool first = true;
    for (int r = 0; r < KNN; ++r) {
        int out = -1;
        if (r < d) {
            unsigned long long best = ~0ull;
            for (int j = gs + lane; j < ge; j += 64) {
                if (j == i) continue;
                float d2 = dist2_exact(xi, yi, zi, x[3 * j], x[3 * j + 1], x[3 * j + 2]);
                unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
                if ((first || key > prev) && key < best) best = key;
            }
            best = wave_min_u64(best);
            prev = best;
            first = false;
            out = (int)(unsigned)(best & 0xffffffffull);
        }
        if (lane == 0) nbr[(size_t)i * KNN + r] = out;
    }
    if (lane == 0) deg[i] = d < 0 ? 0 : d;
}

// ------------------------------------------------------------------------------------------------
// global distance gate: one thread per (node, slot); MLP 20 -> 160 -> LN -> ReLU -> 1 -> sigmoid,
// three passes over the 160 hidden units (mean, centred variance, output) so nothing spills.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_gate_kernel(const float* __restrict__ wts, const float* __restrict__ x,
                                                        const int32_t* __restrict__ nbr,
                                                        const int32_t* __restrict__ deg, int n_nodes,
                                                        float* __restrict__ e_w) {
    __shared__ float sW1[GH * G];
    __shared__ float sB1[GH], sG[GH], sBe[GH], sW2[GH];
    for (int t = threadIdx.x; t < GH * G; t += blockDim.x) sW1[t] = wts[GATE_W1 + t];
    for (int t = threadIdx.x; t < GH; t += blockDim.x) {
        sB1[t] = wts[GATE_B1 + t];
        sG[t] = wts[GATE_LNG + t];
        sBe[t] = wts[GATE_LNB + t];
        sW2[t] = wts[GATE_W2 + t];
    }
    __syncthreads();
    const float b2 = wts[GATE_B2];
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n_nodes * KNN) return;
    const int i = (int)(e >> 5), s = (int)(e & 31);
    if (s >= deg[i]) { e_w[e] = 0.f; return; }
    const int j = nbr[e];
    const float dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    float r[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { float t = dist - c_rbf_mu[g]; r[g] = expf(-0.5f * (t * t)); }
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
        float z = fmaxf((y - mean) * rstd * sG[u] + sBe[u], 0.f);
        acc = fmaf(sW2[u], z, acc);
    }
    e_w[e] = 1.f / (1.f + expf(-acc));
}

// ------------------------------------------------------------------------------------------------
// node GEMM: C[M, nout] = act(A[M,128] @ Wt[128, nout] + bias).  16 rows per block, A tile
// transposed in LDS so each k step reads the 16 row values with 4 ds_read_b128 broadcasts.
// ACT: 0 none, 1 softplus(x) - ln 2 (ShiftedSoftplus, repo/modules/common.py:174-180).
// ------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256) void node_gemm_kernel(const float* __restrict__ A, int lda,
                                                        const float* __restrict__ Wt, const float* __restrict__ bias,
                                                        float* __restrict__ C, int ldc, int M_all, int nout,
                                                        const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    __shared__ __attribute__((aligned(16))) float sA[H][16];
    __shared__ int sRow[16];
    const int M = rows ? *n_rows_ptr : M_all;   // optional row list: A / C rows rows[k], k < *n_rows_ptr
    const int row0 = blockIdx.x * 16;
    if (row0 >= M) return;
    if (threadIdx.x < 16) sRow[threadIdx.x] = row0 + threadIdx.x < M ? (rows ? rows[row0 + threadIdx.x] : row0 + threadIdx.x) : -1;
    __syncthreads();
    for (int t = threadIdx.x; t < 16 * H; t += 256) {
        int r = t >> 7, k = t & 127;
        sA[k][r] = (sRow[r] >= 0) ? A[(size_t)sRow[r] * lda + k] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < nout; c += 256) {
        float acc[16];
        const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
        for (int k = 0; k < H; ++k) {
            const float w = Wt[(size_t)k * nout + c];
            const float4* a4 = reinterpret_cast<const float4*>(&sA[k][0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 a = a4[q];
                acc[4 * q + 0] = fmaf(a.x, w, acc[4 * q + 0]);
                acc[4 * q + 1] = fmaf(a.y, w, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(a.z, w, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(a.w, w, acc[4 * q + 3]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (sRow[r] >= 0) {
                float v = acc[r];
                if (ACT == 1) v = (v > 20.f ? v : log1pf(expf(v))) - 0.69314718055994530942f;
                C[(size_t)sRow[r] * ldc + c] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// query path: q = Linear(ReLU(LN(P[:,512:640]))) then the key's second Linear is folded into the
// query (k only ever appears in q.k, and its bias cancels in the softmax over incoming edges):
//   Qt[i][a][m] = (1/sqrt(8)) * sum_c q[i][8a+c] * Wbk[8a+c][m]     so   q_i,a . k_e,a / sqrt(8) = Qt[i][a] . hid_k[e]
// 16 rows per block, 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void node_query_kernel(const float* __restrict__ att, const float* __restrict__ P,
                                                         float* __restrict__ Qt, int M) {
    __shared__ __attribute__((aligned(16))) float sZ[H][16];  // LN+ReLU(q hidden), transposed
    __shared__ float sQ[16][H];
    const int row0 = blockIdx.x * 16;
    const int tid = threadIdx.x;
    {   // LayerNorm: 16 threads per row
        const int r = tid >> 4, part = tid & 15;
        const bool ok = row0 + r < M;
        const float* p = P + (size_t)(row0 + r) * PROW + 4 * H;
        float v[8];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { v[u] = ok ? p[part + 16 * u] : 0.f; s += v[u]; }
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
            int k = part + 16 * u;
            sZ[k][r] = fmaxf((v[u] - mean) * rstd * att[A_LNQ_G + k] + att[A_LNQ_B + k], 0.f);
        }
    }
    __syncthreads();
    {   // q = Z @ Wq1t + bq1 : thread -> column c = tid & 127, rows half*8..+7
        const int c = tid & 127, half = tid >> 7;
        float acc[8];
        const float b = att[A_BQ1 + c];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = b;
        for (int k = 0; k < H; ++k) {
            const float w = att[A_WQ1T + (size_t)k * H + c];
            const float4* a4 = reinterpret_cast<const float4*>(&sZ[k][half * 8]);
            float4 a0 = a4[0], a1 = a4[1];
            acc[0] = fmaf(a0.x, w, acc[0]); acc[1] = fmaf(a0.y, w, acc[1]);
            acc[2] = fmaf(a0.z, w, acc[2]); acc[3] = fmaf(a0.w, w, acc[3]);
            acc[4] = fmaf(a1.x, w, acc[4]); acc[5] = fmaf(a1.y, w, acc[5]);
            acc[6] = fmaf(a1.z, w, acc[6]); acc[7] = fmaf(a1.w, w, acc[7]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) sQ[half * 8 + r][c] = acc[r];
    }
    __syncthreads();
    {   // Qt: thread -> m = tid & 127, heads half*8..+7, all 16 rows
        const int m = tid & 127, half = tid >> 7;
        const float scale = 0.35355339059327376220f;  // 1/sqrt(8)
        for (int a = half * 8; a < half * 8 + 8; ++a) {
            float w[DH];
#pragma unroll
            for (int c = 0; c < DH; ++c) w[c] = att[A_WBK + (size_t)(a * DH + c) * H + m];
            for (int r = 0; r < 16; ++r) {
                if (row0 + r >= M) break;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < DH; ++c) s = fmaf(sQ[r][a * DH + c], w[c], s);
                Qt[((size_t)(row0 + r) * HEADS + a) * H + m] = s * scale;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fused edge kernel (x2h / h2x): one 128-thread workgroup per destination node i.
//   pre[e]   = PD[i] + PS[j_e] + Wt[type_e] + Wr[type_e] . rbf(|x_i - x_j|)      (k and v, 128 each)
//   hid[e]   = ReLU(LN(pre[e]))
//   score    = Qt[i][a] . hid_k[e]  ->  softmax over the node's incoming edges (per head)
//   x2h: S[a] = sum_e alpha[e][a] e_w[e] hid_v[e];  h_out = h + Wbv_a S[a] + bbv * sum_e alpha e_w
//   h2x: w[e][a] = (Wbv[a] . hid_v[e] + bbv[a]) e_w[e];  dx = 1/16 sum_a sum_e alpha w (x_i - x_j)
// ------------------------------------------------------------------------------------------------
constexpr int HP = H + 4;  // LDS row pitch (floats)

template <bool X2H>
__global__ __launch_bounds__(128) void edge_attention_kernel(
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ h,
    const float* __restrict__ P, const float* __restrict__ Qt, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ deg, const uint8_t* __restrict__ lig, const uint8_t* __restrict__ gen,
    const float* __restrict__ e_w, int n_nodes, float* __restrict__ out, float* __restrict__ dx_out) {
    __shared__ __attribute__((aligned(16))) float sHk[KNN][HP];
    __shared__ __attribute__((aligned(16))) float sHv[KNN][HP];
    __shared__ __attribute__((aligned(16))) float sQt[HEADS][HP];
    __shared__ __attribute__((aligned(16))) float sS[HEADS][HP];
    __shared__ float sRbf[KNN][G];
    __shared__ float sRel[KNN][3];
    __shared__ float sEw[KNN];
    __shared__ int sNb[KNN], sTy[KNN];
    __shared__ float sScore[KNN][HEADS];  // scores, then alpha*e_w (x2h) / alpha (h2x)
    __shared__ float sWv[KNN][HEADS];     // h2x: per-edge per-head value
    __shared__ float sSw[HEADS];
    __shared__ float sStat[KNN][4];

    const int i = blockIdx.x;
    const int t = threadIdx.x;
    const int d = deg[i];
    const int lig_i = lig[i];

    if (t < KNN) {
        int j = t < d ? nbr[(size_t)i * KNN + t] : -1;
        sNb[t] = j;
        if (j >= 0) {
            float rx = x[3 * i] - x[3 * j], ry = x[3 * i + 1] - x[3 * j + 1], rz = x[3 * i + 2] - x[3 * j + 2];
            sRel[t][0] = rx; sRel[t][1] = ry; sRel[t][2] = rz;
            float dist = sqrtf(rx * rx + ry * ry + rz * rz);
#pragma unroll
            for (int g = 0; g < G; ++g) { float u = dist - c_rbf_mu[g]; sRbf[t][g] = expf(-0.5f * (u * u)); }
            // unitransformer.py:92-97: (src lig, dst lig)->0, (lig, prot)->1, (prot, lig)->2, (prot, prot)->3
            int lig_j = lig[j];
            sTy[t] = lig_j ? (lig_i ? 0 : 1) : (lig_i ? 2 : 3);
            sEw[t] = e_w[(size_t)i * KNN + t];
        }
    }
    for (int u = t; u < HEADS * H; u += 128) sQt[u >> 7][u & 127] = Qt[(size_t)i * HEADS * H + u];
    __syncthreads();

    {   // pre-activations, thread = hidden channel
        const float pdk = P[(size_t)i * PROW + t], pdv = P[(size_t)i * PROW + H + t];
        for (int e = 0; e < d; ++e) {
            const int j = sNb[e], ty = sTy[e];
            float pk = pdk + P[(size_t)j * PROW + 2 * H + t] + att[A_WT + ty * 2 * H + t];
            float pv = pdv + P[(size_t)j * PROW + 3 * H + t] + att[A_WT + ty * 2 * H + H + t];
            const float* wr = att + A_WR + (size_t)ty * G * 2 * H;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float r = sRbf[e][g];
                pk = fmaf(wr[g * 2 * H + t], r, pk);
                pv = fmaf(wr[g * 2 * H + H + t], r, pv);
            }
            sHk[e][t] = pk;
            sHv[e][t] = pv;
        }
    }
    __syncthreads();
    {   // LayerNorm statistics: 4 threads per edge
        const int e = t >> 2, part = t & 3;
        float sk = 0.f, sv = 0.f;
        if (e < d) for (int u = 0; u < 32; ++u) { sk += sHk[e][part + 4 * u]; sv += sHv[e][part + 4 * u]; }
        sk += __shfl_xor(sk, 1, 64); sk += __shfl_xor(sk, 2, 64);
        sv += __shfl_xor(sv, 1, 64); sv += __shfl_xor(sv, 2, 64);
        const float mk = sk * (1.f / H), mv = sv * (1.f / H);
        float qk = 0.f, qv = 0.f;
        if (e < d) for (int u = 0; u < 32; ++u) {
            float a = sHk[e][part + 4 * u] - mk, b = sHv[e][part + 4 * u] - mv;
            qk += a * a; qv += b * b;
        }
        qk += __shfl_xor(qk, 1, 64); qk += __shfl_xor(qk, 2, 64);
        qv += __shfl_xor(qv, 1, 64); qv += __shfl_xor(qv, 2, 64);
        if (part == 0) {
            sStat[e][0] = mk; sStat[e][1] = 1.f / sqrtf(qk * (1.f / H) + 1e-5f);
            sStat[e][2] = mv; sStat[e][3] = 1.f / sqrtf(qv * (1.f / H) + 1e-5f);
        }
    }
    __syncthreads();
    {
        const float gk = att[A_LNK_G + t], bk = att[A_LNK_B + t], gv = att[A_LNV_G + t], bv = att[A_LNV_B + t];
        for (int e = 0; e < d; ++e) {
            sHk[e][t] = fmaxf((sHk[e][t] - sStat[e][0]) * sStat[e][1] * gk + bk, 0.f);
            sHv[e][t] = fmaxf((sHv[e][t] - sStat[e][2]) * sStat[e][3] * gv + bv, 0.f);
        }
    }
    __syncthreads();
    // scores (and, for h2x, the per-edge per-head values): (e, a) pairs, 4 per thread
    for (int p = t; p < KNN * HEADS; p += 128) {
        const int e = p >> 4, a = p & 15;
        if (e < d) {
            float s = 0.f;
            for (int m = 0; m < H; ++m) s = fmaf(sQt[a][m], sHk[e][m], s);
            sScore[e][a] = s;
            if (!X2H) {
                float w = att[A_BBV + a];
                const float* wb = att + A_WBV + (size_t)a * H;
                for (int m = 0; m < H; ++m) w = fmaf(wb[m], sHv[e][m], w);
                sWv[e][a] = w * sEw[e];
            }
        }
    }
    __syncthreads();
    if (t < HEADS) {  // softmax over incoming edges, per head (torch_scatter.scatter_softmax semantics)
        float mx = -INFINITY;
        for (int e = 0; e < d; ++e) mx = fmaxf(mx, sScore[e][t]);
        float den = 0.f;
        for (int e = 0; e < d; ++e) { float ex = expf(sScore[e][t] - mx); sScore[e][t] = ex; den += ex; }
        float sw = 0.f;
        for (int e = 0; e < d; ++e) {
            float al = sScore[e][t] / den;
            if (X2H) { al *= sEw[e]; sw += al; }
            sScore[e][t] = al;
        }
        sSw[t] = sw;
    }
    __syncthreads();
    if (X2H) {
        float acc[HEADS];
#pragma unroll
        for (int a = 0; a < HEADS; ++a) acc[a] = 0.f;
        for (int e = 0; e < d; ++e) {
            const float v = sHv[e][t];
#pragma unroll
            for (int a = 0; a < HEADS; ++a) acc[a] = fmaf(sScore[e][a], v, acc[a]);
        }
#pragma unroll
        for (int a = 0; a < HEADS; ++a) sS[a][t] = acc[a];
        __syncthreads();
        const int a = t >> 3;
        float o = att[A_BBV + t] * sSw[a];
        for (int m = 0; m < H; ++m) o = fmaf(att[A_WBV + (size_t)m * H + t], sS[a][m], o);
        out[(size_t)i * H + t] = h[(size_t)i * H + t] + o;
    } else {
        if (t < 3) {
            float acc = 0.f;
            for (int e = 0; e < d; ++e) {
                float s = 0.f;
#pragma unroll
                for (int a = 0; a < HEADS; ++a) s = fmaf(sScore[e][a], sWv[e][a], s);
                acc = fmaf(s, sRel[e][t], acc);
            }
            acc *= (1.f / HEADS);
            if (dx_out) dx_out[3 * i + t] = acc;
            out[3 * i + t] = x[3 * i + t] + (gen[i] ? acc : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers of the first-generation kernels (test-only library libcbgx_xcheck.so, see dispatch.hip)
// ------------------------------------------------------------------------------------------------
#define CBGX_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return _e;               \
    } while (0)

hipError_t launch_knn_v1(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int32_t* nbr,
                         int32_t* deg, hipStream_t s) {
    profile_mark_begin(K_KNN, s);
    hipLaunchKernelGGL(knn_graph_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, s, x, graph_ptr, n_graphs, n_nodes,
                       nbr, deg);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gate_v1(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                          float* e_w, hipStream_t s) {
    long total = (long)n_nodes * KNN;
    profile_mark_begin(K_GATE, s);
    hipLaunchKernelGGL(edge_gate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, packed, x, nbr, deg,
                       n_nodes, e_w);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_node_gemm_v1(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M,
                               int nout, int act, hipStream_t s, const int* rows, const int* n_rows) {
    if (M == 0) return hipSuccess;
    dim3 grid((M + 15) / 16), block(256);
    profile_mark_begin(K_NODE_GEMM, s);
    if (act == 0)
        hipLaunchKernelGGL(node_gemm_kernel<0>, grid, block, 0, s, A, lda, Wt, bias, C, ldc, M, nout, rows, n_rows);
    else
        hipLaunchKernelGGL(node_gemm_kernel<1>, grid, block, 0, s, A, lda, Wt, bias, C, ldc, M, nout, rows, n_rows);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_node_query_v1(const float* att, const float* P, float* Qt, int n_nodes, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    profile_mark_begin(K_NODE_QUERY, s);
    hipLaunchKernelGGL(node_query_kernel, dim3((n_nodes + 15) / 16), dim3(256), 0, s, att, P, Qt, n_nodes);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_attention_v1(bool x2h, const float* att, const float* x, const float* h, const int32_t* nbr,
                               const int32_t* deg, const uint8_t* lig, const uint8_t* gen, const float* e_w, int n_nodes,
                               float* P, float* Qt, float* out, float* dx_out, hipStream_t s) {
    hipError_t e = launch_node_gemm_v1(h, H, att + A_WN, att + A_BN, P, PROW, n_nodes, PROW, 0, s, nullptr, nullptr);
    if (e != hipSuccess) return e;
    e = launch_node_query_v1(att, P, Qt, n_nodes, s);
    if (e != hipSuccess) return e;
    profile_mark_begin(x2h ? K_EDGE_X2H : K_EDGE_H2X, s);
    if (x2h)
        hipLaunchKernelGGL(edge_attention_kernel<true>, dim3(n_nodes), dim3(128), 0, s, att, x, h, P, Qt, nbr, deg,
                           lig, gen, e_w, n_nodes, out, dx_out);
    else
        hipLaunchKernelGGL(edge_attention_kernel<false>, dim3(n_nodes), dim3(128), 0, s, att, x, h, P, Qt, nbr, deg,
                           lig, gen, e_w, n_nodes, out, dx_out);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cbgx

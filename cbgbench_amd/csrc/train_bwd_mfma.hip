// libcbgx -- second-generation backward of one attention block: same algorithm and LDS staging as
// edge_backward_kernel (train_bwd.hip), with the six inner products that carry ~98 % of its FLOPs issued as
// v_mfma_f32_16x16x4_f32 tiles fed from LDS.  One workgroup = 8 waves (two per SIMD, so one wave's LDS / memory waits
// overlap the other's matrix work) = one destination node at a time; wave w owns the 32 k|v columns [32 w, 32 w + 32).
//   1  pre-activations      [32 e x 20 g] x [20 g x 256 c]    per source class present  (B = rbf columns, from L2)
//   4  scores / values      [32 e x 128 m] x [128 m x 16 a]   (k path: Qt, v path: Gt or the h2x value matrix)
//   6  folds T / S          [16 a x 32 e] x [32 e x 128 m]
//   7  d(hidden)            [32 e x 16 a] x [16 a x 128 m]
//   9  d(rbf columns)       [20 g x 32 e] x [32 e x 256 c]    accumulated in registers across the nodes of a workgroup
//   10 d(rbf)               [32 e x 256 c] x [256 c x 20 g]
// LayerNorm statistics, softmax, the dpre pass (atomics to the neighbours' projection rows) and the final coordinate
// scatter stay on the vector ALU: they are O(32 x 256) per node.  Operands of every MFMA block are loaded into
// registers first (explicit scheduling fences): the compiler otherwise serialises load -> wait -> 2 MFMAs.
// MFMA 16x16x4 fp32 operand layout (lane l: i = l & 15, kq = l >> 4): A[i][kq], B[kq][i], D regs r: D[4 kq + r][i].
//
// The file also holds the node-level kernels of the backward (further down, each with its own header): the query-MLP
// backward (q_backward_mfma_kernel), the weight-gradient outer products over nodes (outer_accum_mfma_kernel,
// wgrad_mfma_kernel) and the input-gradient product (dgrad_mfma_kernel).  They read both MFMA operands straight from global
// memory (no LDS staging), issue all gathers of a pass before its first MFMA and never predicate a load.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__constant__ float c_mu_m[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

constexpr int EPM = H + 4;   // pitch of the [32][128] arrays
constexpr int HP2 = 20;      // pitch of the [32][16] arrays (conflict-free when a lane group walks rows)
constexpr int BWD_THREADS = 512;

struct EdgeBwdMfmaLds {
    float N[2][KNN][EPM];      // pre-activation -> normalised -> dpre        (path 0 = k, 1 = v)
    float U[2][KNN][EPM];      // hidden (post ReLU) -> d(normalised)
    float QG[2][HEADS][EPM];   // [0] folded query; [1] x2h: folded output gradient, h2x: second v Linear [16][128]
    float rbf[KNN][G];
    float rbfc[2][KNN][33];    // rbf masked by source class (0 outside the class, in padded slots and for g >= 20)
    float rel[KNN][4];
    float ew[KNN];
    float scp[2][2][KNN][HP2]; // [scores | values][half of the m range][edge][head]: partial sums of step 4
    float sc[KNN][HP2];        // alpha
    float gv[KNN][HP2];        // x2h: G_i,a . vraw_e,a ; h2x: raw per-head value
    float cf[2][KNN][HP2];     // [0] ds (score gradient); [1] x2h: alpha e_w, h2x: d(raw value)
    float stat[2][KNN][2];
    float stat2[2][KNN][2];
    float me[KNN];
    float ddp[2][2][KNN];      // partial d(dist): [half of the column range][half of the rbf index][edge]
    float D[4];
    float bb[HEADS];
    float dwr3[G][2 * H];      // d(rbf columns) of the dominant edge type 3, accumulated across the nodes of the workgroup
    int nb[KNN];
    int nbs[KNN];              // nb, with the node itself in padded slots (safe to gather from / add zeros to)
    int ty[KNN];
    int cls[KNN];              // 1 = ligand source, 0 = protein source, -1 = padded slot
    float wrt3[2 * H][G];      // transposed rbf columns of the dominant edge type 3: B operand of step 10 from LDS
};

template <bool X2H>
__global__ __launch_bounds__(BWD_THREADS) void edge_backward_mfma_kernel(
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ P,
    const float* __restrict__ Qt, const float* __restrict__ Gt, const float* __restrict__ gb,
    const float* __restrict__ gx_out, const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg,
    const uint8_t* __restrict__ lig, const float* __restrict__ e_w, const int* __restrict__ rows,
    const int* __restrict__ n_rows_ptr, int n_nodes, float* __restrict__ T, float* __restrict__ S,
    float* __restrict__ sw, float* __restrict__ dP, float* __restrict__ dx, float* __restrict__ de_w,
    float* __restrict__ partial, int abl, int centred) {
    __shared__ EdgeBwdMfmaLds L;
    const int t = threadIdx.x;
    // vector-ALU view: column c of the k|v pair (path p, channel m), edges [16 eh, 16 eh + 16)
    const int c0 = t & 255, eh = __builtin_amdgcn_readfirstlane(t >> 8), p = __builtin_amdgcn_readfirstlane((t & 255) >> 7);
    const int m0 = c0 & 127;
    // matrix view: wave w owns columns [32 w, +32) = channels [mbase, +32) of path pw   (wave-uniform -> scalar registers)
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63, li0 = l & 15, kq0 = l >> 4;
    const int pw = w >> 2, mbase = 32 * (w & 3), cbase = 32 * w;
    int c = c0, m = m0, li = li0, kq = kq0;
    const float gamma = att[(p == 0 ? A_LNK_G : A_LNV_G) + m];
    const float beta = att[(p == 0 ? A_LNK_B : A_LNV_B) + m];
    float gam2[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) gam2[ct] = att[(pw == 0 ? A_LNK_G : A_LNV_G) + mbase + 16 * ct + li];

    // d(rbf columns) of the dominant edge type 3 (protein -> protein) is accumulated in LDS across the nodes of the
    // workgroup (each element is owned by one lane: plain read-modify-write).  The other three types (an endpoint is a
    // ligand atom, ~10 % of the edges) are added to the workgroup's private slab in memory node by node.
    float* slab = partial + (size_t)blockIdx.x * PB_SIZE;
    for (int u = t; u < 3 * G * 2 * H; u += BWD_THREADS) slab[PB_WR + u] = 0.f;     // types 0..2
    for (int u = t; u < NT * 2 * H; u += BWD_THREADS) slab[PB_WT + u] = 0.f;
    for (int u = t; u < G * 2 * H; u += BWD_THREADS) (&L.dwr3[0][0])[u] = 0.f;
    float aWt[NT] = {0.f, 0.f, 0.f, 0.f};
    float aG2[2] = {0.f, 0.f}, aB2[2] = {0.f, 0.f};
    float aBb = 0.f;
    floatx4 aV16[2];            // h2x: second v Linear gradient [16 heads][the wave's 32 channels]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) aV16[ct] = (floatx4){0.f, 0.f, 0.f, 0.f};

    for (int u = t; u < 2 * KNN * EPM; u += BWD_THREADS) { (&L.N[0][0][0])[u] = 0.f; (&L.U[0][0][0])[u] = 0.f; }
    for (int u = t; u < 2 * KNN * HP2; u += BWD_THREADS) (&L.cf[0][0][0])[u] = 0.f;
    // (LDS fills: every load of the thread goes out before the first value is stored -- as plain loops they compile to one
    // dependent L2 round trip per iteration, 14 of them at the head of every launch)
    if (!X2H) {
        float v[HEADS * H / BWD_THREADS];
#pragma unroll
        for (int k = 0; k < HEADS * H / BWD_THREADS; ++k) v[k] = att[A_WBV + t + BWD_THREADS * k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < HEADS * H / BWD_THREADS; ++k) { const int u = t + BWD_THREADS * k; L.QG[1][u >> 7][u & 127] = v[k]; }
    }
    {
        float v[G * 2 * H / BWD_THREADS];
#pragma unroll
        for (int k = 0; k < G * 2 * H / BWD_THREADS; ++k) v[k] = att[A_WR + (size_t)3 * G * 2 * H + t + BWD_THREADS * k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < G * 2 * H / BWD_THREADS; ++k) { const int u = t + BWD_THREADS * k; L.wrt3[u & 255][u >> 8] = v[k]; }
    }
    for (int u = t; u < 2 * KNN * 33; u += BWD_THREADS) (&L.rbfc[0][0][0])[u] = 0.f;
    const int count = rows ? *n_rows_ptr : n_nodes;
    // geometry of a node is built by all 512 threads: thread = (edge ge, rbf group gg); the (neighbour, coordinates, flags)
    // of the NEXT node are fetched one iteration ahead so their latency is hidden behind the current node's work
    const int ge = t & 31, gg = t >> 5;
    int pj = -1, plig = 0, pdeg = 0, pligi = 0;
    float pxj = 0.f, pyj = 0.f, pzj = 0.f, pxi = 0.f, pyi = 0.f, pzi = 0.f, pew = 0.f;
    auto prefetch_geometry = [&](int node) {
        pdeg = deg[node];
        pligi = lig[node];
        pxi = x[3 * node]; pyi = x[3 * node + 1]; pzi = x[3 * node + 2];
        pj = ge < pdeg ? nbr[(size_t)node * KNN + ge] : -1;
        const int js = pj >= 0 ? pj : node;
        pxj = x[3 * js]; pyj = x[3 * js + 1]; pzj = x[3 * js + 2];
        plig = lig[js];
        pew = e_w[(size_t)node * KNN + ge];
    };
    // the node's folded query / folded output gradient (16 KB) are fetched one iteration ahead as well
    float pq[4], pg[4];
    auto prefetch_qg = [&](int node) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pq[k] = Qt[(unsigned)node * HEADS * H + t + BWD_THREADS * k];
            if (X2H) pg[k] = Gt[(unsigned)node * HEADS * H + t + BWD_THREADS * k];
        }
    };
    if ((int)blockIdx.x < count) {
        prefetch_geometry(rows ? rows[blockIdx.x] : (int)blockIdx.x);
        prefetch_qg(rows ? rows[blockIdx.x] : (int)blockIdx.x);
    }
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const int i = rows ? rows[it] : it;
        // the lane coordinates are re-materialised every iteration: otherwise every LDS address below is loop-invariant,
        // gets hoisted out of the node loop and the ~100 hoisted addresses are spilled to scratch
        c = c0; m = m0; li = li0; kq = kq0;
        asm volatile("" : "+v"(c), "+v"(m), "+v"(li), "+v"(kq));
        const int d = pdeg;
        const int lig_i = pligi;
        const int ty_prot = lig_i ? 2 : 3, ty_lig = lig_i ? 0 : 1;   // edge type by source class (unitransformer.py:92-97)
        __syncthreads();
        {
            const int j = pj;
            const int cl = j >= 0 ? (plig ? 1 : 0) : -1;
            const float rx = pxi - pxj, ry = pyi - pyj, rz = pzi - pzj;
            const float dist = j >= 0 ? sqrtf(rx * rx + ry * ry + rz * rz) : 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int g = gg + 16 * k;
                if (g < G) {
                    const float u = dist - c_mu_m[g];
                    const float r = j >= 0 ? expf(-0.5f * (u * u)) : 0.f;
                    L.rbf[ge][g] = r;
                    L.rbfc[0][ge][g] = cl == 0 ? r : 0.f;
                    L.rbfc[1][ge][g] = cl == 1 ? r : 0.f;
                }
            }
            if (gg == 0) {
                L.nb[ge] = j;
                L.nbs[ge] = j >= 0 ? j : i;
                L.rel[ge][0] = j >= 0 ? rx : 0.f; L.rel[ge][1] = j >= 0 ? ry : 0.f; L.rel[ge][2] = j >= 0 ? rz : 0.f;
                L.rel[ge][3] = dist;
                L.ty[ge] = cl == 1 ? ty_lig : (cl == 0 ? ty_prot : 3);
                L.ew[ge] = j >= 0 ? pew : 0.f;
                L.cls[ge] = cl;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = t + BWD_THREADS * k;
            L.QG[0][u >> 7][u & 127] = pq[k];
            if (X2H) L.QG[1][u >> 7][u & 127] = pg[k];
        }
        if (!X2H && t < 3) L.D[t] = gx_out[3 * i + t];
        __syncthreads();
        if (it + (int)gridDim.x < count) {
            const int nxt = rows ? rows[it + gridDim.x] : it + (int)gridDim.x;
            prefetch_geometry(nxt);
            prefetch_qg(nxt);
        }
        // which source classes occur among the node's edges (uniform over the workgroup)
        bool has_cls[2];
        has_cls[0] = __ballot(l < KNN && L.cls[l & 31] == 0) != 0ull;
        has_cls[1] = __ballot(l < KNN && L.cls[l & 31] == 1) != 0ull;

        {   // 1. pre-activations: the wave's 32 columns x 32 edges
            floatx4 acc[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cl = 0; cl < 2; ++cl) {
                if (!has_cls[cl]) continue;
                // P from the MFMA node kernel is centred (and holds the protein-source type column): use the centred rbf columns
                const float* wr = att + (centred ? A_WRC : A_WR) + (unsigned)(cl ? ty_lig : ty_prot) * G * 2 * H + cbase + li;
                float a0[G / 4], a1[G / 4], bq[G / 4][2];
#pragma unroll
                for (int s = 0; s < G / 4; ++s) {
                    const int g = 4 * s + kq;
                    a0[s] = L.rbfc[cl][li][g];
                    a1[s] = L.rbfc[cl][16 + li][g];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) bq[s][ct] = wr[(unsigned)g * 2 * H + 16 * ct];
                }
                SCHED_FENCE();
#pragma unroll
                for (int s = 0; s < G / 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        acc[0][ct] = MFMA(a0[s], bq[s][ct], acc[0][ct]);
                        acc[1][ct] = MFMA(a1[s], bq[s][ct], acc[1][ct]);
                    }
                SCHED_FENCE();
            }
            // epilogue without data-dependent branches (so all gathers are in flight together): padded slots gather the
            // node's own row and are zeroed again in step 3
            float pd[2], wtv[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int cc = cbase + 16 * ct + li;
                pd[ct] = P[(unsigned)i * PROW + cc];
                wtv[0][ct] = centred ? 0.f : att[A_WT + ty_prot * 2 * H + cc];
                wtv[1][ct] = centred ? att[A_IMG + IMG_WT + lig_i * 2 * H + cc] : att[A_WT + ty_lig * 2 * H + cc];
            }
            float ps[2][4][2];
            int clv[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 16 * rt + 4 * kq + r;
                    const int j = L.nbs[e];
                    clv[rt][r] = L.cls[e];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        ps[rt][r][ct] = (abl & 2) ? 0.f : P[(unsigned)j * PROW + 2 * H + cbase + 16 * ct + li];
                }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 16 * rt + 4 * kq + r;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        L.N[pw][e][mbase + 16 * ct + li] =
                            acc[rt][ct][r] + pd[ct] + ps[rt][r][ct] + (clv[rt][r] == 1 ? wtv[1][ct] : wtv[0][ct]);
                }
        }
        __syncthreads();
        {   // 2. LayerNorm statistics: 8 threads per (path, edge)
            const int pe = t >> 3, part = t & 7, pp = pe >> 5, e = pe & 31;
            float s = 0.f;
            if (e < d) {
#pragma unroll
                for (int u = 0; u < 16; ++u) s += L.N[pp][e][part + 8 * u];
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.f / H);
            float q = 0.f;
            if (e < d) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { const float a = L.N[pp][e][part + 8 * u] - mean; q += a * a; }
            }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            if (part == 0) { L.stat[pp][e][0] = mean; L.stat[pp][e][1] = 1.f / sqrtf(q * (1.f / H) + 1e-5f); }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 16; ++k) {   // 3. normalise, affine, ReLU (padded slots: exact zeros, they enter the K sums below)
            const int e = 16 * eh + k;
            float n = 0.f, u = 0.f;
            if (e < d) {
                n = (L.N[p][e][m] - L.stat[p][e][0]) * L.stat[p][e][1];
                u = fmaxf(n * gamma + beta, 0.f);
            }
            L.N[p][e][m] = n;
            L.U[p][e][m] = u;
        }
        __syncthreads();
        {   // 4. scores (which = 0) and per-head values (which = 1): wave = (which, row tile, half of the m range)
            const int rt = w & 1, which = (w >> 1) & 1, kh = w >> 2;
            const float* ua = &L.U[which][16 * rt + li][64 * kh + kq];
            const float* qb = &L.QG[which][li][64 * kh + kq];
            floatx4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            float av[2][8], bv[2][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { av[0][k] = ua[4 * k]; bv[0][k] = qb[4 * k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { av[1][k] = ua[4 * (8 + k)]; bv[1][k] = qb[4 * (8 + k)]; }
            SCHED_FENCE();
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    a0 = MFMA(av[ch][k], bv[ch][k], a0);
                    a1 = MFMA(av[ch][k + 1], bv[ch][k + 1], a1);
                }
            SCHED_FENCE();
#pragma unroll
            for (int r = 0; r < 4; ++r) L.scp[which][kh][16 * rt + 4 * kq + r][li] = a0[r] + a1[r];
        }
        __syncthreads();
        if (!(abl & 8)) {   // 5. softmax over the incoming edges and its backward: thread = (head, edge), 32 lanes per head
            const int a = t >> 5, e = t & 31;
            const bool on = e < d;
            const float sv = on ? L.scp[0][0][e][a] + L.scp[0][1][e][a] : -INFINITY;
            const float gvv = L.scp[1][0][e][a] + L.scp[1][1][e][a] + (X2H ? gb[(size_t)i * HEADS + a] : att[A_BBV + a]);
            const float rho = X2H ? 1.f : (L.D[0] * L.rel[e][0] + L.D[1] * L.rel[e][1] + L.D[2] * L.rel[e][2]);
            const float ewe = L.ew[e];
            float mx = sv;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float ex = on ? expf(sv - mx) : 0.f;
            float den = ex;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) den += __shfl_xor(den, off, 64);
            const float al = d > 0 ? ex / den : 0.f;
            const float dal = X2H ? ewe * gvv : rho * gvv * ewe * (1.f / HEADS);
            float cacc = al * dal;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) cacc += __shfl_xor(cacc, off, 64);
            const float wv = X2H ? al * ewe : rho * al * (1.f / HEADS) * ewe;
            L.sc[e][a] = al;
            L.gv[e][a] = gvv;
            L.cf[0][e][a] = on ? al * (dal - cacc) : 0.f;
            L.cf[1][e][a] = on ? wv : 0.f;
            float swacc = on ? wv : 0.f;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) swacc += __shfl_xor(swacc, off, 64);
            if (e == 0) { if (X2H) sw[(size_t)i * HEADS + a] = swacc; else L.bb[a] = swacc; }
        }
        __syncthreads();
        if (!X2H && t < HEADS) aBb += L.bb[t];
        if (t < KNN && t < d) {   // gate gradient and (h2x) the scalar coefficient of rel in delta_x
            const int e = t;
            float de = 0.f, me = 0.f;
            if (X2H) {
#pragma unroll
                for (int a = 0; a < HEADS; ++a) de = fmaf(L.sc[e][a], L.gv[e][a], de);
            } else {
                const float rho = L.D[0] * L.rel[e][0] + L.D[1] * L.rel[e][1] + L.D[2] * L.rel[e][2];
                float av = 0.f;
#pragma unroll
                for (int a = 0; a < HEADS; ++a) av = fmaf(L.sc[e][a], L.gv[e][a], av);
                de = rho * av * (1.f / HEADS);
                me = av * L.ew[e] * (1.f / HEADS);
            }
            de_w[(size_t)i * KNN + e] += de;
            L.me[e] = me;
        }
        {   // 6. folds over the edges, the wave's 32 channels: T (k path) / S or the h2x value-matrix gradient (v path)
            floatx4 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = (!X2H && pw == 1) ? aV16[ct] : (floatx4){0.f, 0.f, 0.f, 0.f};
            float av[8], bv[8][2];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int e = 4 * s + kq;
                av[s] = L.cf[pw][e][li];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bv[s][ct] = L.U[pw][e][mbase + 16 * ct + li];
            }
            SCHED_FENCE();
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = MFMA(av[s], bv[s][ct], acc[ct]);
            SCHED_FENCE();
            if (!X2H && pw == 1) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) aV16[ct] = acc[ct];
            } else if (!(abl & 16)) {
                float* dst = (pw == 0 ? T : S) + (unsigned)i * HEADS * H;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(unsigned)(4 * kq + r) * H + mbase + 16 * ct + li] = acc[ct][r];
            }
        }
        {   // 7. d(hidden) -> d(normalised pre-activation), the wave's 32 channels x 32 edges
            floatx4 acc[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
            {
                float a0[4], a1[4], bv[4][2];
#pragma unroll
                for (int s = 0; s < HEADS / 4; ++s) {
                    const int a = 4 * s + kq;
                    a0[s] = L.cf[pw][li][a];
                    a1[s] = L.cf[pw][16 + li][a];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) bv[s][ct] = L.QG[pw][a][mbase + 16 * ct + li];
                }
                SCHED_FENCE();
#pragma unroll
                for (int s = 0; s < HEADS / 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        acc[0][ct] = MFMA(a0[s], bv[s][ct], acc[0][ct]);
                        acc[1][ct] = MFMA(a1[s], bv[s][ct], acc[1][ct]);
                    }
                SCHED_FENCE();
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 16 * rt + 4 * kq + r;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int mm = mbase + 16 * ct + li;
                        const float dy = L.U[pw][e][mm] > 0.f ? acc[rt][ct][r] : 0.f;
                        aG2[ct] = fmaf(dy, L.N[pw][e][mm], aG2[ct]);
                        aB2[ct] += dy;
                        L.U[pw][e][mm] = dy * gam2[ct];
                    }
                }
        }
        __syncthreads();
        {   // 8. LayerNorm backward statistics: 8 threads per (path, edge)
            const int pe = t >> 3, part = t & 7, pp = pe >> 5, e = pe & 31;
            float s1 = 0.f, s2 = 0.f;
            if (e < d) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float dn = L.U[pp][e][part + 8 * u];
                    s1 += dn;
                    s2 = fmaf(dn, L.N[pp][e][part + 8 * u], s2);
                }
            }
            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
            if (part == 0) { L.stat2[pp][e][0] = s1 * (1.f / H); L.stat2[pp][e][1] = s2 * (1.f / H); }
        }
        __syncthreads();
        {   // 9a. dpre; own projection row, neighbour rows (atomics), type columns of the first Linear.
            // branch-free: padded slots carry exact zeros (N, U, stat2) and add 0 to the node's own row
            float accpd = 0.f, acclig = 0.f;
#pragma unroll
            for (int e0 = 0; e0 < 16; e0 += 8) {
                float dpv[8];
                int jv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = 16 * eh + e0 + k;
                    const float n = L.N[p][e][m];
                    dpv[k] = L.stat[p][e][1] * (L.U[p][e][m] - L.stat2[p][e][0] - n * L.stat2[p][e][1]);
                    jv[k] = L.nbs[e];
                    acclig += L.cls[e] == 1 ? dpv[k] : 0.f;
                    accpd += dpv[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    L.N[p][16 * eh + e0 + k][m] = dpv[k];
                    if (!(abl & 1)) atomicAdd(&dP[(unsigned)jv[k] * PROW + 2 * H + c], dpv[k]);
                }
            }
            atomicAdd(&dP[(unsigned)i * PROW + c], accpd);      // two edge halves per column; dP is zeroed by the caller
            // type columns: a node's edges have one of two types, by source class (padded slots contribute zeros)
            if (lig_i) { aWt[0] += acclig; aWt[2] += accpd - acclig; } else { aWt[1] += acclig; aWt[3] += accpd - acclig; }
        }
        __syncthreads();
        // 9b. rbf columns of the first Linear: dWr[type][g][c] += sum_e rbf[e][g] dpre[e][c], per source class present
#define CBGX_ACC_WR(DST, CL)                                                                       \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                         \
            float a0[4], a1[4], bv[4][2];                                                          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                        \
                const int e = 4 * (4 * hf + s) + kq;                                               \
                a0[s] = L.rbfc[CL][e][li];                                                         \
                a1[s] = L.rbfc[CL][e][16 + li];                                                    \
                _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) bv[s][ct] = L.N[pw][e][mbase + 16 * ct + li]; \
            }                                                                                      \
            SCHED_FENCE();                                                                         \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                          \
                _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                 \
                    DST[0][ct] = MFMA(a0[s], bv[s][ct], DST[0][ct]);                               \
                    DST[1][ct] = MFMA(a1[s], bv[s][ct], DST[1][ct]);                               \
                }                                                                                  \
            SCHED_FENCE();                                                                         \
        }
        if (!(abl & 4)) {
#pragma unroll
            for (int cl = 0; cl < 2; ++cl) {
                if (!has_cls[cl]) continue;
                {
                    floatx4 tmp[2][2];
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) tmp[gt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
                    if (cl == 0) { CBGX_ACC_WR(tmp, 0) } else { CBGX_ACC_WR(tmp, 1) }
                    const int tyc = cl ? ty_lig : ty_prot;
#pragma unroll
                    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = 16 * gt + 4 * kq + r;
                            if (g < G) {
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct) {
                                    const int cc = cbase + 16 * ct + li;
                                    if (tyc == 3) L.dwr3[g][cc] += tmp[gt][ct][r];
                                    else atomicAdd(&slab[PB_WR + (tyc * G + g) * 2 * H + cc], tmp[gt][ct][r]);   // no-return atomic
                                }
                            }
                        }
                }
            }
        }
#undef CBGX_ACC_WR
        if (!(abl & 4)) {   // 10. through the radial basis: drbf[e][g] = sum_c dpre[e][c] Wr[type_e][g][c]
            // wave = (row tile, g tile, half of the column range).  The product is formed for every source class present
            // with unmasked inputs; each output row then takes the result of its own edge's class.
            const int rt = w & 1, gt = (w >> 1) & 1, kh = w >> 2, e_a = 16 * rt + li, gl = 16 * gt + li;
            floatx4 dr[2][2];
#pragma unroll
            for (int cl = 0; cl < 2; ++cl) { dr[cl][0] = (floatx4){0.f, 0.f, 0.f, 0.f}; dr[cl][1] = (floatx4){0.f, 0.f, 0.f, 0.f}; }
#define CBGX_LOAD_CH(BUF, CH, BEXPR)                                                   \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                    \
        const int c4 = 128 * kh + 4 * (8 * (CH) + k);                                  \
        av[BUF][k] = L.N[kh][e_a][4 * (8 * (CH) + k) + kq];                            \
        bv[BUF][k] = BEXPR;                                                            \
    }
#define CBGX_DRBF_CLASS(CL, BEXPR)                                                     \
    {                                                                                  \
        _Pragma("unroll 1") for (int ch = 0; ch < 4; ++ch) {                           \
            float av[1][8], bv[1][8];                                                  \
            CBGX_LOAD_CH(0, ch, BEXPR)                                                 \
            SCHED_FENCE();                                                             \
            _Pragma("unroll") for (int k = 0; k < 8; k += 2) {                         \
                dr[CL][0] = MFMA(av[0][k], bv[0][k], dr[CL][0]);                       \
                dr[CL][1] = MFMA(av[0][k + 1], bv[0][k + 1], dr[CL][1]);               \
            }                                                                          \
            SCHED_FENCE();                                                             \
        }                                                                              \
    }
            if (has_cls[0]) {
                if (!lig_i) {   // type 3 (protein -> protein): table in LDS (columns g >= 20 of the tile are discarded below)
                    const float* wl = &L.wrt3[kq][gl < G ? gl : 0];
                    CBGX_DRBF_CLASS(0, wl[(size_t)c4 * G])
                } else {
                    const float* wt = att + A_WRT + ((size_t)2 * 2 * H + kq) * 32 + gl;
                    CBGX_DRBF_CLASS(0, wt[(size_t)c4 * 32])
                }
            }
            if (has_cls[1]) {
                const float* wt = att + A_WRT + ((size_t)ty_lig * 2 * H + kq) * 32 + gl;
                CBGX_DRBF_CLASS(1, wt[(size_t)c4 * 32])
            }
#undef CBGX_DRBF_CLASS
#undef CBGX_LOAD_CH
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 16 * rt + 4 * kq + r;
                float v = 0.f;
                if (gl < G && e < d) {
                    const float dsum = L.cls[e] == 1 ? dr[1][0][r] + dr[1][1][r] : dr[0][0][r] + dr[0][1][r];
                    v = dsum * (-(L.rel[e][3] - c_mu_m[gl]) * L.rbf[e][gl]);
                }
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (li == 0) L.ddp[kh][gt][e] = v;
            }
        }
        __syncthreads();
        if (t < KNN && t < d) {
            const int e = t;
            const float dist = L.rel[e][3];
            const float dd = (L.ddp[0][0][e] + L.ddp[0][1][e]) + (L.ddp[1][0][e] + L.ddp[1][1][e]);
            const float cf = dist > 0.f ? dd / dist : 0.f;
            const int j = L.nb[e];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float g3 = cf * L.rel[e][k];
                if (!X2H) g3 = fmaf(L.me[e], L.D[k], g3);
                atomicAdd(&dx[3 * i + k], g3);
                atomicAdd(&dx[3 * j + k], -g3);
            }
        }
    }
    // per-workgroup partial sums of the edge-indexed weight gradients
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NT; ++a) atomicAdd(&slab[PB_WT + a * 2 * H + c], aWt[a]);      // two edge halves per column
    for (int u = t; u < G * 2 * H; u += BWD_THREADS) slab[PB_WR + 3 * G * 2 * H + u] = (&L.dwr3[0][0])[u];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {   // the 4 k-quarters of a column hold partial sums: fold them, lanes 0..15 store
        float g4 = aG2[ct], b4 = aB2[ct];
        g4 += __shfl_xor(g4, 16, 64); g4 += __shfl_xor(g4, 32, 64);
        b4 += __shfl_xor(b4, 16, 64); b4 += __shfl_xor(b4, 32, 64);
        if (kq == 0) { slab[PB_LNG + cbase + 16 * ct + li] = g4; slab[PB_LNB + cbase + 16 * ct + li] = b4; }
    }
    if (!X2H) {
        if (pw == 1) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[PB_WBV16 + (4 * kq + r) * H + mbase + 16 * ct + li] = aV16[ct][r];
        }
        if (t < HEADS) slab[PB_BBV16 + t] = aBb;
    }
}

// ------------------------------------------------------------------------------------------------
// Query path backward on the matrix cores.  Same contract as q_backward_kernel (train_bwd.hip, kept as the VALU
// cross-check): for every listed node recompute z = ReLU(LN(P[:,512:640])) and q = Wq1 z + bq1, take
// dq = (1/sqrt 8) fold(T, Wbk), and produce qs = q/sqrt(8), dq, z and dP[:,512:640]; the LayerNorm affine gradients
// are accumulated into partial[0:256] by atomics.
//
// One 16-node tile per 4-wave workgroup pass; the three [16 x 128] x [128 x 128] products of a tile are MFMA
// 16x16x4 chains whose A operand is read four k at a time (one ds_read_b128 / global float4 per four MFMAs): lane
// (i, kq) holds k = 16 s + 4 kq + j for j = 0..3, and the B operand is loaded for the same k, so the k permutation
// cancels.  Every load is unconditional (out-of-range rows read row 0 and are masked at the stores): predicated loads
// compile to branch + wait per element, which is what made the VALU version latency bound (~130 us per tile).
// ------------------------------------------------------------------------------------------------
constexpr int QB_LD = H + 4;   // row stride of the LDS tiles (16-byte aligned rows, conflict-light b128 reads)

#ifndef CBGX_QB_WAVES
#define CBGX_QB_WAVES 4     // waves per SIMD = workgroups per CU.  A tile's dependent chain is ~45 us whatever runs beside it, so the launch
                            // time is rounds x 45: a training batch of 16 506 nodes is 1 032 tiles on 4 x 256 = 1 024 slots -- eight
                            // workgroups run a second tile and the launch takes two rounds (88 us); 5 fits them in one (96 registers)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CBGX_QB_WAVES, CBGX_QB_WAVES))) void q_backward_mfma_kernel(const float* __restrict__ att, const float* __restrict__ P,
                                                              const float* __restrict__ T, const int* __restrict__ rows,
                                                              const int* __restrict__ n_rows_ptr, int n_nodes,
                                                              float* __restrict__ qs, float* __restrict__ dqb,
                                                              float* __restrict__ zb, float* __restrict__ dP,
                                                              float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float sZ[16][QB_LD];    // z (post ReLU); later d(normalised) in place
    __shared__ __attribute__((aligned(16))) float sNq[16][QB_LD];   // normalised pre-activation
    __shared__ __attribute__((aligned(16))) float sDq[16][QB_LD];   // dq
    __shared__ float sRstd[16];
    __shared__ int sRow[16];
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const float s8 = 0.35355339059327376220f;
    float aG[2] = {0.f, 0.f}, aB[2] = {0.f, 0.f};
    const int count = rows ? *n_rows_ptr : n_nodes;
    const int tiles = (count + 15) / 16;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        if (t < 16) { const int it = tile * 16 + t; sRow[t] = it < count ? (rows ? rows[it] : it) : -1; }
        __syncthreads();
        {   // LayerNorm + ReLU of the query hidden, 16 threads per row
            const int r = t >> 4, part = t & 15, node = sRow[r];
            const unsigned base = (unsigned)(node < 0 ? 0 : node) * PROW + 4 * H;
            float v[8], s = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = P[base + part + 16 * u]; s += v[u]; }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            const float mean = s * (1.f / H);
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) q += (v[u] - mean) * (v[u] - mean);
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
            const float rstd = 1.f / sqrtf(q * (1.f / H) + 1e-5f);
            const float live = node >= 0 ? 1.f : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = part + 16 * u;
                const float nq = (v[u] - mean) * rstd * live;
                const float z = fmaxf(nq * att[A_LNQ_G + k] + att[A_LNQ_B + k], 0.f) * live;
                sNq[r][k] = nq;
                sZ[r][k] = z;
                if (node >= 0) zb[(unsigned)node * H + k] = z;
            }
            if (part == 0) sRstd[r] = rstd;
        }
        __syncthreads();
        // ---- dq: wave w owns heads 4w .. 4w+3; A = T rows straight from global, B = 8 columns of Wbk^T ------
        // one head at a time (rolled: keeps the live set to one head's gathers), results go straight to LDS / HBM;
        // lane column c = i < 8 is n = 8a + c, register r is row 4 kq + r
        {
            const int node = sRow[i];
            const unsigned trow = (unsigned)(node < 0 ? 0 : node) * HEADS * H;
#pragma unroll 1
            for (int hh = 0; hh < 4; ++hh) {
                const int a = 4 * w + hh;
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                float4 ta[8];
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) ta[sp] = *reinterpret_cast<const float4*>(T + trow + a * H + 16 * sp + 4 * kq);
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    const float* wb = att + A_WBKT + (unsigned)(16 * sp + 4 * kq) * H + 8 * a + (i & 7);
                    const float b0 = wb[0], b1 = wb[H], b2 = wb[2 * H], b3 = wb[3 * H];
                    acc = MFMA(ta[sp].x, b0, acc);
                    acc = MFMA(ta[sp].y, b1, acc);
                    acc = MFMA(ta[sp].z, b2, acc);
                    acc = MFMA(ta[sp].w, b3, acc);
                }
                const int n = 8 * a + (i & 7);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dq = acc[r] * s8;
                    const int nd = sRow[4 * kq + r];
                    if (i < 8) {
                        sDq[4 * kq + r][n] = dq;
                        if (nd >= 0) dqb[(unsigned)nd * H + n] = dq;
                    }
                }
            }
        }
        {   // ---- q = Wq1 z + bq1: wave w owns columns 32w .. 32w+31 ------------------------------------
            floatx4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
            for (int sp = 0; sp < 8; ++sp) {
                const float4 za = *reinterpret_cast<const float4*>(&sZ[i][16 * sp + 4 * kq]);
                const float* wq = att + A_WQ1T + (unsigned)(16 * sp + 4 * kq) * H + 32 * w + i;
                float b[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { b[0][j] = wq[j * H]; b[1][j] = wq[j * H + 16]; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    acc[tt] = MFMA(za.x, b[tt][0], acc[tt]);
                    acc[tt] = MFMA(za.y, b[tt][1], acc[tt]);
                    acc[tt] = MFMA(za.z, b[tt][2], acc[tt]);
                    acc[tt] = MFMA(za.w, b[tt][3], acc[tt]);
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int n = 32 * w + 16 * tt + i;
                const float b1 = att[A_BQ1 + n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int node = sRow[4 * kq + r];
                    if (node >= 0) qs[(unsigned)node * H + n] = (acc[tt][r] + b1) * s8;
                }
            }
        }
        __syncthreads();
        {   // ---- dz = dq Wq1 -> ReLU/LayerNorm-affine backward; wave w owns hidden columns 32w .. 32w+31 ----
            floatx4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
            for (int sp = 0; sp < 8; ++sp) {
                const float4 da = *reinterpret_cast<const float4*>(&sDq[i][16 * sp + 4 * kq]);
                const float* wq = att + A_WQ1O + (unsigned)(16 * sp + 4 * kq) * H + 32 * w + i;
                float b[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { b[0][j] = wq[j * H]; b[1][j] = wq[j * H + 16]; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    acc[tt] = MFMA(da.x, b[tt][0], acc[tt]);
                    acc[tt] = MFMA(da.y, b[tt][1], acc[tt]);
                    acc[tt] = MFMA(da.z, b[tt][2], acc[tt]);
                    acc[tt] = MFMA(da.w, b[tt][3], acc[tt]);
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int k = 32 * w + 16 * tt + i;
                const float gq = att[A_LNQ_G + k];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 4 * kq + r;
                    const float dy = sZ[rr][k] > 0.f ? acc[tt][r] : 0.f;    // z = 0 on padding rows
                    aG[tt] = fmaf(dy, sNq[rr][k], aG[tt]);
                    aB[tt] += dy;
                    sZ[rr][k] = dy * gq;       // each element is read and rewritten by the same lane only
                }
            }
        }
        __syncthreads();
        {   // LayerNorm backward, 16 threads per row
            const int r = t >> 4, part = t & 15, node = sRow[r];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dn = sZ[r][part + 16 * u];
                s1 += dn;
                s2 = fmaf(dn, sNq[r][part + 16 * u], s2);
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
            const float m1 = s1 * (1.f / H), m2 = s2 * (1.f / H), rstd = sRstd[r];
            if (node >= 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = part + 16 * u;
                    dP[(unsigned)node * PROW + 4 * H + k] = rstd * (sZ[r][k] - m1 - sNq[r][k] * m2);
                }
            }
        }
    }
    // LayerNorm affine gradients [gamma | beta]: fold the four kq lanes of a column, then one atomic per column and wave
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        float g = aG[tt], b = aB[tt];
        g += __shfl_xor(g, 16, 64); g += __shfl_xor(g, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (kq == 0) {
            atomicAdd(&partial[32 * w + 16 * tt + i], g);
            atomicAdd(&partial[H + 32 * w + 16 * tt + i], b);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Outer-product accumulation over nodes on the matrix cores (same contract as outer_accum_kernel, train_bwd.hip):
//   dW[n][m] = sum_i L[i][n] * R[i][(HEADED ? n >> 3 : 0)][m],   one 128 x 128 partial per workgroup.
// The node index is the MFMA k dimension: lane (i, kq) feeds node kq of a group of four, A = L[node][16 tn + i],
// B = R[node][..][m].  Wave w owns rows n = 32w .. 32w+31 (heads 4w .. 4w+3) and all 128 columns; B comes in as two
// float4 per (node, head) with the column labelling m = 64 tmq + 4 i + j, so one load feeds four column tiles and
// the results leave as float4 stores.  HEADED row tiles span two heads: two MFMAs per tile with the other head's
// rows of A zeroed.  Eight nodes per pass, every load unconditional (padding nodes read row 0 with A = 0).
// ------------------------------------------------------------------------------------------------
template <bool HEADED>
__global__ __launch_bounds__(256) void outer_accum_mfma_kernel(const float* __restrict__ Lm, const float* __restrict__ R,
                                                               const int* __restrict__ rows,
                                                               const int* __restrict__ n_rows_ptr, int n_nodes,
                                                               float* __restrict__ partial, size_t slab_stride) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    floatx4 acc[2][8];
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[tr][c] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int count = rows ? *n_rows_ptr : n_nodes;
    const float lo = (i >> 3) == 0 ? 1.f : 0.f, hi = 1.f - lo;
    for (int base = blockIdx.x * 8; base < count; base += gridDim.x * 8) {
        float a[2][2];
        float4 b[2][HEADED ? 4 : 1][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int it = base + 4 * g + kq;
            const bool ok = it < count;
            const int itc = ok ? it : 0;
            const unsigned node = (unsigned)(rows ? rows[itc] : itc);
            const float live = ok ? 1.f : 0.f;
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) a[g][tr] = Lm[node * H + 32 * w + 16 * tr + i] * live;
            if (HEADED) {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const float* r = R + (node * HEADS + 4 * w + hh) * H + 4 * i;
                    b[g][hh][0] = *reinterpret_cast<const float4*>(r);
                    b[g][hh][1] = *reinterpret_cast<const float4*>(r + 64);
                }
            } else {
                const float* r = R + node * H + 4 * i;
                b[g][0][0] = *reinterpret_cast<const float4*>(r);
                b[g][0][1] = *reinterpret_cast<const float4*>(r + 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // all gathers of the pass in flight before the first MFMA waits on one
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) {
                if (HEADED) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float am = a[g][tr] * (hf ? hi : lo);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float4 bb = b[g][2 * tr + hf][q];
                            acc[tr][4 * q + 0] = MFMA(am, bb.x, acc[tr][4 * q + 0]);
                            acc[tr][4 * q + 1] = MFMA(am, bb.y, acc[tr][4 * q + 1]);
                            acc[tr][4 * q + 2] = MFMA(am, bb.z, acc[tr][4 * q + 2]);
                            acc[tr][4 * q + 3] = MFMA(am, bb.w, acc[tr][4 * q + 3]);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4 bb = b[g][0][q];
                        acc[tr][4 * q + 0] = MFMA(a[g][tr], bb.x, acc[tr][4 * q + 0]);
                        acc[tr][4 * q + 1] = MFMA(a[g][tr], bb.y, acc[tr][4 * q + 1]);
                        acc[tr][4 * q + 2] = MFMA(a[g][tr], bb.z, acc[tr][4 * q + 2]);
                        acc[tr][4 * q + 3] = MFMA(a[g][tr], bb.w, acc[tr][4 * q + 3]);
                    }
                }
            }
    }
    // D register r of tile (tr, q, j): row n = 32w + 16tr + 4kq + r, column m = 64q + 4i + j
    float* slab = partial + (size_t)blockIdx.x * slab_stride;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 o = make_float4(acc[tr][4 * q + 0][r], acc[tr][4 * q + 1][r], acc[tr][4 * q + 2][r],
                                             acc[tr][4 * q + 3][r]);
                *reinterpret_cast<float4*>(slab + (size_t)(32 * w + 16 * tr + 4 * kq + r) * H + 64 * q + 4 * i) = o;
            }
}

// ------------------------------------------------------------------------------------------------
// The two dense products of the backward whose long dimension is the node index, without LDS staging: both MFMA
// operands are read from global memory in the layout the instruction wants, all loads of a pass are issued before the
// first MFMA (sched_barrier) and none is predicated (rows past the end are clamped and not stored).
//
// wgrad:  C_g[128][128 cb .. +127] = sum over the nodes of group g of L[node][0:128]^T R[node][128 cb .. +127]
//         (weight gradient of a Linear applied to every node: node = MFMA k, one partial slab per node group,
//         folded by the caller).  Wave w owns rows 32w .. 32w+31; columns labelled m = 64 q + 4 i + j (float4 loads).
// dgrad:  C[row][0:128] (+)= A[row][0:K] B[col][0:K]^T   (gradient w.r.t. the Linear's input: row = node)
//         32 rows per workgroup, wave w owns columns 32w .. 32w+31; k labelled 16 s + 4 kq + j on both operands.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const float* __restrict__ Lm, int ldl, const float* __restrict__ R,
                                                         int ldr, int n_nodes, int nodes_per_group,
                                                         float* __restrict__ partial, int ldo, size_t slab_stride,
                                                         const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    // `rows` / `n_rows_ptr`: sum over the listed nodes only (device-side count; the other rows of R are known to be zero -- the
    // h2x blocks touch the movable nodes and their neighbours); the groups then split the list
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int cb = blockIdx.y;
    if (rows) {
        n_nodes = *n_rows_ptr;
        nodes_per_group = ((n_nodes + (int)gridDim.x - 1) / (int)gridDim.x + 15) / 16 * 16;
    }
    floatx4 acc[2][8];
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[tr][c] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int n0 = blockIdx.x * nodes_per_group;
    const int n1 = min(n_nodes, n0 + nodes_per_group);
    for (int base = n0; base < n1; base += 16) {
        float a[4][2];
        float4 b[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int it = base + 4 * g + kq;
            const bool ok = it < n1;
            const int sel = ok ? it : n0;
            const unsigned node = (unsigned)(rows ? rows[sel] : sel);
            const float live = ok ? 1.f : 0.f;
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) a[g][tr] = Lm[node * ldl + 32 * w + 16 * tr + i] * live;
            const float* r = R + node * ldr + 128 * cb + 4 * i;
            b[g][0] = *reinterpret_cast<const float4*>(r);
            b[g][1] = *reinterpret_cast<const float4*>(r + 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 bb = b[g][q];
                    acc[tr][4 * q + 0] = MFMA(a[g][tr], bb.x, acc[tr][4 * q + 0]);
                    acc[tr][4 * q + 1] = MFMA(a[g][tr], bb.y, acc[tr][4 * q + 1]);
                    acc[tr][4 * q + 2] = MFMA(a[g][tr], bb.z, acc[tr][4 * q + 2]);
                    acc[tr][4 * q + 3] = MFMA(a[g][tr], bb.w, acc[tr][4 * q + 3]);
                }
    }
    float* slab = partial + (size_t)blockIdx.x * slab_stride + 128 * cb;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 o = make_float4(acc[tr][4 * q + 0][r], acc[tr][4 * q + 1][r], acc[tr][4 * q + 2][r],
                                             acc[tr][4 * q + 3][r]);
                *reinterpret_cast<float4*>(slab + (size_t)(32 * w + 16 * tr + 4 * kq + r) * ldo + 64 * q + 4 * i) = o;
            }
}

// C[M x 128] (+)= A[M x K] . B^T, B = [128 x K] row-major: the input-gradient product (dh += dP . Wn^T).  LDS-tiled: a workgroup
// (4 waves) owns 64 rows x 128 columns, wave w the columns [32 w, 32 w + 32); k in chunks of 32, double-buffered through registers.
// The first version read both MFMA operands straight from global memory in fragment order -- 16 rows x 64 bytes per load
// instruction, the most expensive shape for the CU's texture addresser (scripts/ubench/vmem.hip: ~64 cycles per instruction), four
// waves each fetching the same A rows: the addresser, not the matrix pipe, set its 69 us.  Here every global load covers 128
// contiguous bytes per row and every operand byte is fetched once per workgroup.
constexpr int DG_KC = 32, DG_PITCH = DG_KC + 4;
template <int TR>      // 16 TR rows per workgroup
__global__ __launch_bounds__(256) void dgrad_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                         int ldb, float* C, int ldc, int M, int K,
                                                         int accumulate, const float* C_in, const int* __restrict__ rows,
                                                         const int* __restrict__ n_rows_ptr) {
    // `rows` / `n_rows_ptr`: only the listed rows of A are multiplied and only those rows of C updated (device-side count)
    constexpr int DG_ROWS = 16 * TR;
    if (rows) {
        M = *n_rows_ptr;
        if ((int)blockIdx.x * DG_ROWS >= M) return;
    }
    __shared__ __attribute__((aligned(16))) float sA[2][DG_ROWS][DG_PITCH];
    __shared__ __attribute__((aligned(16))) float sB[2][H][DG_PITCH];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * DG_ROWS;
    // loader view: thread = (row t >> 3 [+32 ...], four k at 4 (t & 7)): 8 lanes cover one row's 128 contiguous bytes
    const int lr = t >> 3, lk = 4 * (t & 7);
    constexpr int NA = TR / 2;      // float4 of A per thread and chunk
    unsigned ao[NA], bo[4];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int k = min(row0 + lr + 32 * u, M - 1);
        ao[u] = (unsigned)(rows ? rows[k] : k) * lda + lk;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) bo[u] = (unsigned)(lr + 32 * u) * ldb + lk;
    // Two register sets: chunk c travels in set c & 1 and is requested TWO chunks before it is staged into LDS (one chunk of
    // MFMAs, ~0.85 us, does not cover a memory round trip when every CU streams at once).
    floatx4 ga[2][NA], gb[2][4];
    auto fetch = [&](auto S, int k0) {
        constexpr int st = decltype(S)::value;
#pragma unroll
        for (int u = 0; u < NA; ++u) ga[st][u] = *reinterpret_cast<const floatx4*>(A + ao[u] + k0);
#pragma unroll
        for (int u = 0; u < 4; ++u) gb[st][u] = *reinterpret_cast<const floatx4*>(B + bo[u] + k0);
    };
    auto stage = [&](auto S) {      // chunk in register set S -> LDS buffer S
        constexpr int st = decltype(S)::value;
#pragma unroll
        for (int u = 0; u < NA; ++u) *reinterpret_cast<floatx4*>(&sA[st][lr + 32 * u][lk]) = ga[st][u];
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<floatx4*>(&sB[st][lr + 32 * u][lk]) = gb[st][u];
    };
    floatx4 acc[TR][2];
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) acc[tr][tc] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](auto S) {
        constexpr int buf = decltype(S)::value;
#pragma unroll
        for (int sp = 0; sp < DG_KC / 16; ++sp) {
            float4 a[TR], b[2];
#pragma unroll
            for (int tr = 0; tr < TR; ++tr) a[tr] = *reinterpret_cast<const float4*>(&sA[buf][16 * tr + i][16 * sp + 4 * kq]);
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) b[tc] = *reinterpret_cast<const float4*>(&sB[buf][32 * w + 16 * tc + i][16 * sp + 4 * kq]);
            // eight independent accumulators between two MFMAs of the same one (a dependent 16x16x4 issues every 40+ cycles, not 32)
#pragma unroll
            for (int comp = 0; comp < 4; ++comp)
#pragma unroll
                for (int tr = 0; tr < TR; ++tr)
#pragma unroll
                    for (int tc = 0; tc < 2; ++tc)
                        acc[tr][tc] = MFMA(reinterpret_cast<const float*>(&a[tr])[comp], reinterpret_cast<const float*>(&b[tc])[comp],
                                           acc[tr][tc]);
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    const int nch = K / DG_KC;       // even: K is a multiple of 64 (launcher)
    fetch(S0{}, 0);
    stage(S0{});
    fetch(S1{}, DG_KC);
    __syncthreads();
    // step(S): chunk ch (parity S) is in LDS buffer S, chunk ch + 1 in flight in register set 1 - S
    auto step = [&](auto S, auto Sn, int ch) {
        if (ch + 2 < nch) fetch(S, (ch + 2) * DG_KC);       // set S is free: its chunk was staged one step ago
        SCHED_FENCE();      // (without the fences the scheduler waits for loads and stages them BEFORE the MFMAs: an exposed memory
                            // round trip per chunk -- what every earlier version of this kernel spent two thirds of its time on)
        compute(S);
        SCHED_FENCE();
        if (ch + 1 < nch) {
            stage(Sn);      // LDS buffer 1 - S: last read one step ago, behind that step's barrier
            __syncthreads();
        }
    };
    for (int ch = 0; ch < nch; ch += 2) {
        step(S0{}, S1{}, ch);
        step(S1{}, S0{}, ch + 1);
    }
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) {
        float cold[2][4];
        unsigned crow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = min(row0 + 16 * tr + 4 * kq + r, M - 1);
            crow[r] = (unsigned)(rows ? rows[k] : k);
        }
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[tc][r] = accumulate ? C_in[crow[r] * ldc + 32 * w + 16 * tc + i] : 0.f;
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * tr + 4 * kq + r;
                if (row < M) C[crow[r] * ldc + 32 * w + 16 * tc + i] = cold[tc][r] + acc[tr][tc][r];
            }
    }
}

#define CBGX_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return _e;               \
    } while (0)

hipError_t launch_edge_backward_mfma(bool x2h, const float* att, const float* x, const float* P, const float* Qt,
                                     const float* Gt, const float* gb, const float* gx_out, const int32_t* nbr,
                                     const int32_t* deg, const uint8_t* lig, const float* e_w, const int* rows,
                                     const int* n_rows, int n_nodes, float* T, float* S, float* sw, float* dP, float* dx,
                                     float* de_w, float* partial, int grid, hipStream_t s, int centred) {
    if ((long)n_nodes * HEADS * H >= (1L << 32)) return hipErrorInvalidValue;   // 32-bit element offsets inside the kernel
#ifdef CBGX_ABLATE
    // timing ablations (WRONG results): only in libcbgx_ablate.so, built by scripts/abl_bwd.sh with -DCBGX_ABLATE
    static const int abl = getenv("CBGX_BWD_ABL") ? atoi(getenv("CBGX_BWD_ABL")) : 0;
#else
    constexpr int abl = 0;
#endif
    profile_mark_begin(x2h ? (rows ? K_EDGE_X2H_BWD_LISTED : K_EDGE_X2H_BWD) : K_EDGE_H2X_BWD, s);
    if (x2h) {
        // the product runs x2h blocks through train_bwd_x2h.hip; this instantiation exists in libcbgx_xcheck.so only
        // (cbgx_debug_set_edge_kernel(2)) as an independent on-device implementation of the same contract
#ifdef CBGX_XCHECK
        hipLaunchKernelGGL(edge_backward_mfma_kernel<true>, dim3(grid), dim3(BWD_THREADS), 0, s, att, x, P, Qt, Gt, gb,
                           gx_out, nbr, deg, lig, e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial, abl, centred);
#else
        profile_mark_end(s);
        return hipErrorNotSupported;
#endif
    } else
        hipLaunchKernelGGL(edge_backward_mfma_kernel<false>, dim3(grid), dim3(BWD_THREADS), 0, s, att, x, P, Qt, Gt, gb,
                           gx_out, nbr, deg, lig, e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial, abl, centred);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_q_backward_mfma(const float* att, const float* P, const float* T, const int* rows, const int* n_rows,
                                  int n_nodes, float* qs, float* dqb, float* zb, float* dP, float* partial, int grid,
                                  hipStream_t s) {
    if ((long)n_nodes * HEADS * H >= (1L << 32)) return hipErrorInvalidValue;   // 32-bit element offsets inside the kernel
    hipLaunchKernelGGL(q_backward_mfma_kernel, dim3(grid), dim3(256), 0, s, att, P, T, rows, n_rows, n_nodes, qs, dqb, zb,
                       dP, partial);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_outer_accum_mfma(bool headed, const float* Lm, const float* R, const int* rows, const int* n_rows,
                                   int n_nodes, float* partial, size_t slab_stride, int grid, hipStream_t s) {
    if ((long)n_nodes * HEADS * H >= (1L << 32)) return hipErrorInvalidValue;   // 32-bit element offsets inside the kernel
    if (headed)
        hipLaunchKernelGGL(outer_accum_mfma_kernel<true>, dim3(grid), dim3(256), 0, s, Lm, R, rows, n_rows, n_nodes,
                           partial, slab_stride);
    else
        hipLaunchKernelGGL(outer_accum_mfma_kernel<false>, dim3(grid), dim3(256), 0, s, Lm, R, rows, n_rows, n_nodes,
                           partial, slab_stride);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

// C_slabs[g][128][ldo] (column blocks of 128) = partial sums over node groups of L^T R; returns the number of slabs
hipError_t launch_wgrad_mfma(const float* Lm, int ldl, const float* R, int ldr, int n_nodes, int col_blocks, float* partial,
                             int ldo, size_t slab_stride, int groups, hipStream_t s, const int* rows, const int* n_rows) {
    if ((long)n_nodes * (ldl > ldr ? ldl : ldr) >= (1L << 32)) return hipErrorInvalidValue;
    const int per = ((n_nodes + groups - 1) / groups + 15) / 16 * 16;
    profile_mark_begin(K_TRAIN_GEMM, s);
    hipLaunchKernelGGL(wgrad_mfma_kernel, dim3(groups, col_blocks), dim3(256), 0, s, Lm, ldl, R, ldr, n_nodes, per, partial,
                       ldo, slab_stride, rows, n_rows);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

// C[M][128] (+)= A[M][K] B[128][K]^T, K a multiple of 64
hipError_t launch_dgrad_mfma(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int K,
                             int accumulate, hipStream_t s, const float* C_in, const int* rows, const int* n_rows) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || (long)M * (lda > ldc ? lda : ldc) >= (1L << 32)) return hipErrorInvalidValue;
    profile_mark_begin(K_TRAIN_GEMM, s);
#ifdef CBGX_ABLATE
    {   // timing ablations (wrong results): 1024 = every row of A is row 0 (A from cache), 2048 = no accumulate (C not read)
        static const int abl = getenv("CBGX_BWD_ABL") ? atoi(getenv("CBGX_BWD_ABL")) : 0;
        if (abl & 1024) lda = 0;
        if (abl & 2048) accumulate = 0;
    }
#endif
    // two workgroups of 32 rows per CU overlap each other's barriers and loads better than one of 64 (measured: 51.7 vs ... us)
    hipLaunchKernelGGL(dgrad_mfma_kernel<2>, dim3((M + 31) / 32), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, K, accumulate,
                       C_in ? C_in : C, rows, n_rows);
    profile_mark_end(s);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cbgx

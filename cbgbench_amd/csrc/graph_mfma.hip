// libcbgx -- once-per-step graph kernels, second generation:
//   knn_graph_reg_kernel : kNN with the wave's candidate keys cached in registers (graphs up to 768 nodes)
//   edge_gate_mfma_kernel: the global distance gate MLP (20 -> 160 -> LN -> ReLU -> 1 -> sigmoid) on MFMA
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"

namespace cbgx {

typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__constant__ float c_mu2[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                               3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float g_xrow_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// lexicographic min of (hi, lo) pairs over the wave; every lane gets the result
__device__ __forceinline__ void take_min(unsigned& hi, unsigned& lo, unsigned ohi, unsigned olo) {
    const bool t = (((unsigned long long)ohi << 32) | olo) < (((unsigned long long)hi << 32) | lo);     // one v_cmp_lt_u64
    hi = t ? ohi : hi;
    lo = t ? olo : lo;
}
__device__ __forceinline__ void wave_min_pair(unsigned& hi, unsigned& lo) {
    take_min(hi, lo, dpp_u32<0xB1>(hi), dpp_u32<0xB1>(lo));     // quad_perm [1,0,3,2]
    take_min(hi, lo, dpp_u32<0x4E>(hi), dpp_u32<0x4E>(lo));     // quad_perm [2,3,0,1]
    take_min(hi, lo, dpp_u32<0x141>(hi), dpp_u32<0x141>(lo));   // row_half_mirror
    take_min(hi, lo, dpp_u32<0x140>(hi), dpp_u32<0x140>(lo));   // row_mirror
    {
        auto a = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        unsigned h0 = a[0], l0 = b[0];
        take_min(h0, l0, a[1], b[1]);
        hi = h0; lo = l0;
    }
    {
        auto a = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        unsigned h0 = a[0], l0 = b[0];
        take_min(h0, l0, a[1], b[1]);
        hi = h0; lo = l0;
    }
}

__device__ __forceinline__ float dist2_exact2(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float s = dx * dx;
    float t = dy * dy;
    s = s + t;
    t = dz * dz;
    s = s + t;
    return s;
}

// One wave per centre node.  Each lane caches the keys (bits(d2), j) of its <= KNN_SLOTS candidates in registers;
// 32 rounds of {lane-local min, wave min, retire the winner}.  Ordering = (squared distance, index), bit-identical
// to the oracle (d2 computed with contraction off).  Graphs larger than 64 * KNN_SLOTS use knn_graph_kernel.
constexpr int KNN_SLOTS = 12;

constexpr int KNN_SLOTS_SMALL = 8;
// the <= 64 * SLOTS candidates of a graph cached in registers (keys (bits(d2), j)): d rounds of {lane-local min, wave min, retire}
template <int SLOTS>
__device__ __forceinline__ int knn_select_cached(const float* __restrict__ x, int gs, int ge, int i, int d, int lane,
                                                 float xi, float yi, float zi) {
    unsigned khi[SLOTS], klo[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int j = gs + lane + 64 * u;
        const bool ok = j < ge && j != i;
        const int jj = ok ? j : i;      // (unconditional gathers: all of a lane's candidates in flight together)
        const unsigned kh = __float_as_uint(dist2_exact2(xi, yi, zi, x[3 * jj], x[3 * jj + 1], x[3 * jj + 2]));
        khi[u] = ok ? kh : 0xffffffffu;
        klo[u] = ok ? (unsigned)j : 0xffffffffu;
    }
    int mine = -1;
    for (int r = 0; r < KNN; ++r) {
        if (r >= d) break;
        unsigned bh = 0xffffffffu, bl = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) take_min(bh, bl, khi[u], klo[u]);
        wave_min_pair(bh, bl);
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
            if (klo[u] == bl) { khi[u] = 0xffffffffu; klo[u] = 0xffffffffu; }   // indices are unique
        if (lane == r) mine = (int)bl;
    }
    return mine;
}

// the search of one centre node i by the wave it is called from: every candidate of the graph is scanned
// -> lane r < 32: the r-th neighbour (-1 past the degree)
__device__ __forceinline__ int knn_scan_node(const float* __restrict__ x, const int32_t* __restrict__ graph_ptr, int n_graphs,
                                             int i, int lane, int32_t* __restrict__ nbr, int32_t* __restrict__ deg) {
    int lo_g = 0, hi_g = n_graphs;
    while (hi_g - lo_g > 1) {
        const int mid = (lo_g + hi_g) >> 1;
        if (graph_ptr[mid] <= i) lo_g = mid; else hi_g = mid;
    }
    const int gs = graph_ptr[lo_g], ge = graph_ptr[lo_g + 1];
    const int n = ge - gs;
    const int d = min(KNN, n - 1);
    const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    int mine = -1;   // lane r keeps the r-th neighbour
    if (n <= 64 * KNN_SLOTS_SMALL) {      // (every pocket of the shipped data: 8 cached candidates per lane instead of 12)
        mine = knn_select_cached<KNN_SLOTS_SMALL>(x, gs, ge, i, d, lane, xi, yi, zi);
    } else if (n <= 64 * KNN_SLOTS) {
        mine = knn_select_cached<KNN_SLOTS>(x, gs, ge, i, d, lane, xi, yi, zi);
    } else {
        // large graph: rescan the candidates every round, keeping the smallest key greater than the previous one
        unsigned ph = 0u, pl = 0u;
        bool first = true;
        for (int r = 0; r < KNN; ++r) {
            if (r >= d) break;
            unsigned bh = 0xffffffffu, bl = 0xffffffffu;
            for (int j = gs + lane; j < ge; j += 64) {
                if (j == i) continue;
                const unsigned kh = __float_as_uint(dist2_exact2(xi, yi, zi, x[3 * j], x[3 * j + 1], x[3 * j + 2]));
                const unsigned kl = (unsigned)j;
                const bool after = first || kh > ph || (kh == ph && kl > pl);
                if (after) take_min(bh, bl, kh, kl);
            }
            wave_min_pair(bh, bl);
            ph = bh; pl = bl; first = false;
            if (lane == r) mine = (int)bl;
        }
    }
    if (lane < KNN) nbr[(size_t)i * KNN + lane] = mine;
    if (lane == 0) deg[i] = d < 0 ? 0 : d;
    return mine;
}

__global__ __launch_bounds__(256) void knn_graph_reg_kernel(const float* __restrict__ x,
                                                            const int32_t* __restrict__ graph_ptr, int n_graphs,
                                                            int n_nodes, int32_t* __restrict__ nbr,
                                                            int32_t* __restrict__ deg, const int* __restrict__ rows,
                                                            const int* __restrict__ n_rows_ptr) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (rows ? *n_rows_ptr : n_nodes)) return;
    // (wave-uniform, and told so: the graph search below is then scalar loads through the scalar cache instead of one vector
    // round trip per halving step -- nine in a row for the 340 graphs of a headline batch)
    const int i = __builtin_amdgcn_readfirstlane(rows ? rows[idx] : idx);      // optional centre list (static-context cache)
    knn_scan_node(x, graph_ptr, n_graphs, i, lane, nbr, deg);
}

// Graph-cached calls (static-context cache): the listed centres are the nodes near a ligand.  A PROTEIN centre's new neighbour
// list is the 32 smallest of (its own pocket list: 32 protein atoms, sorted by (d2, index), distances unchanged because protein
// atoms do not move) U (the ligand atoms of its graph) -- any other protein atom is beaten by all 32 of the pocket list.  Instead
// of scanning the ~550 atoms of the graph through 32 rounds of a wave-wide minimum, every candidate's RANK in the union is counted:
//   pocket entry p:   p + #{ligand keys < its key}          ligand atom:   #{pocket keys < key} + #{ligand keys < key}
// (keys = (bits(d2), index), compared lexicographically, d2 evaluated exactly as in the scan: the result is bit-identical to it --
// tests/test_gpu_parity.py::test_static_context_cache_is_exact compares the two paths on every node).  The ligand atoms close a
// graph's rows (compose_context, common.py:200); up to 128 of them are handled here.  Ligand centres, graphs with more ligand atoms
// and graphs above the register-cached size take the scan.
// (one 64-bit compare: the lexicographic order of (hi, lo) is the order of hi << 32 | lo -- five 32-bit operations otherwise, in
// loops that made the merge VALU-bound: ~1100 vector instructions per centre, 60 k centres per headline batch)
__device__ __forceinline__ bool key_less(unsigned ah, unsigned al, unsigned bh, unsigned bl) {
    return (((unsigned long long)ah << 32) | al) < (((unsigned long long)bh << 32) | bl);
}

__global__ __launch_bounds__(256) void knn_merge_kernel(const float* __restrict__ x, const int32_t* __restrict__ graph_ptr,
                                                        int n_graphs, const uint8_t* __restrict__ lig,
                                                        const int32_t* __restrict__ s_nbr, const int32_t* __restrict__ s_deg,
                                                        int32_t* __restrict__ nbr, int32_t* __restrict__ deg,
                                                        const int* __restrict__ rows, const int* __restrict__ n_rows_ptr,
                                                        const float* __restrict__ s_ew, float* __restrict__ e_w,
                                                        unsigned* __restrict__ newmask) {
    // `e_w` / `newmask` (optional, round 5): the gate values of the pocket entries that stay in the list move to their new rank
    // here (the gate is a function of the distance between two atoms that never move), padded slots get 0, and newmask[i] = the
    // ranks that hold a NEW entry -- the only ones edge_gate_mfma_kernel then evaluates (unitransformer.py:109-112)
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= *n_rows_ptr) return;
    const int i = __builtin_amdgcn_readfirstlane(rows[idx]);      // wave-uniform: scalar graph search, scalar loop bounds
    // everything that depends on the centre alone is requested before the graph search, unconditionally (clamped slots), so that
    // a centre costs three dependent round trips (list entry -> these -> positions) instead of one per `if`
    const unsigned lig_i = lig[i];
    const int sd_raw = s_deg[i];
    const int js_raw = s_nbr[(size_t)i * KNN + (lane & (KNN - 1))];
    const float ews = e_w ? s_ew[(size_t)i * KNN + (lane & (KNN - 1))] : 0.f;
    const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    int lo_g = 0, hi_g = n_graphs;
    while (hi_g - lo_g > 1) {
        const int mid = (lo_g + hi_g) >> 1;
        if (graph_ptr[mid] <= i) lo_g = mid; else hi_g = mid;
    }
    const int gs = graph_ptr[lo_g], ge = graph_ptr[lo_g + 1];
    // ligand atoms of the graph: the run of flagged rows at its end, counted over the last 128 rows
    const int r0 = ge - 1 - lane, r1 = ge - 65 - lane;
    const unsigned l0 = lig[r0 >= gs ? r0 : gs], l1 = lig[r1 >= gs ? r1 : gs];      // both in flight
    const unsigned long long b0 = __ballot((r0 >= gs) & (l0 != 0u));
    const unsigned long long b1 = __ballot((r1 >= gs) & (l1 != 0u));
    const int n0 = b0 == ~0ull ? 64 : __builtin_ctzll(~b0);
    const int n1 = n0 < 64 ? 0 : (b1 == ~0ull ? 64 : __builtin_ctzll(~b1));
    const int nl = n0 + n1;
    if (lig_i != 0u || nl >= 128 || ge - gs > 64 * KNN_SLOTS) {      // wave-uniform
        const int mine = knn_scan_node(x, graph_ptr, n_graphs, i, lane, nbr, deg);
        if (e_w) {      // every valid slot is new
            if (lane < KNN && mine < 0) e_w[(size_t)i * KNN + lane] = 0.f;
            const unsigned long long b = __ballot(lane < KNN && mine >= 0);
            if (lane == 0) newmask[i] = (unsigned)(b & 0xffffffffull);
        }
        return;
    }
    const unsigned NONE = 0xffffffffu;
    // pocket candidates: lane p < sd holds entry p of the static list
    const int sd = __builtin_amdgcn_readfirstlane(sd_raw);
    const bool vs = lane < sd && lane < KNN;
    const int js = vs ? js_raw : i;
    // ligand candidates: rows ls .. ge-1, two per lane
    const int ls = ge - nl;
    const bool v0 = lane < nl, v1 = lane + 64 < nl;
    const int j0 = v0 ? ls + lane : i, j1 = v1 ? ls + 64 + lane : i;
    // the three position gathers in flight together (padded slots read the centre itself)
    const float sx = x[3 * js], sy = x[3 * js + 1], sz = x[3 * js + 2];
    const float ax = x[3 * j0], ay = x[3 * j0 + 1], az = x[3 * j0 + 2];
    const float bx = x[3 * j1], by = x[3 * j1 + 1], bz = x[3 * j1 + 2];
    const unsigned ks_h = vs ? __float_as_uint(dist2_exact2(xi, yi, zi, sx, sy, sz)) : NONE;
    const unsigned ks_l = vs ? (unsigned)js : NONE;
    const unsigned k0_h = v0 ? __float_as_uint(dist2_exact2(xi, yi, zi, ax, ay, az)) : NONE;
    const unsigned k1_h = v1 ? __float_as_uint(dist2_exact2(xi, yi, zi, bx, by, bz)) : NONE;
    const unsigned k0_l = v0 ? (unsigned)j0 : NONE, k1_l = v1 ? (unsigned)j1 : NONE;
    int rs = lane, r0k = 0, r1k = 0;      // ranks in the union
    // every ligand key against every candidate of this lane (the loop index is wave-uniform: v_readlane broadcasts key t)
    for (int t = 0; t < min(nl, 64); ++t) {
        const unsigned bh = __builtin_amdgcn_readlane(k0_h, t), bl = __builtin_amdgcn_readlane(k0_l, t);
        rs += key_less(bh, bl, ks_h, ks_l) ? 1 : 0;
        r0k += key_less(bh, bl, k0_h, k0_l) ? 1 : 0;
        r1k += key_less(bh, bl, k1_h, k1_l) ? 1 : 0;
    }
    for (int t = 0; t < nl - 64; ++t) {
        const unsigned bh = __builtin_amdgcn_readlane(k1_h, t), bl = __builtin_amdgcn_readlane(k1_l, t);
        rs += key_less(bh, bl, ks_h, ks_l) ? 1 : 0;
        r0k += key_less(bh, bl, k0_h, k0_l) ? 1 : 0;
        r1k += key_less(bh, bl, k1_h, k1_l) ? 1 : 0;
    }
    for (int p = 0; p < min(sd, KNN); ++p) {   // every pocket key against this lane's ligand candidates
        const unsigned bh = __builtin_amdgcn_readlane(ks_h, p), bl = __builtin_amdgcn_readlane(ks_l, p);
        r0k += key_less(bh, bl, k0_h, k0_l) ? 1 : 0;
        r1k += key_less(bh, bl, k1_h, k1_l) ? 1 : 0;
    }
    const int d = min(KNN, min(sd, KNN) + nl);
    int32_t* out = nbr + (size_t)i * KNN;
    if (lane >= d && lane < KNN) out[lane] = -1;
    if (vs && rs < KNN) out[rs] = js;
    if (v0 && r0k < KNN) out[r0k] = j0;
    if (v1 && r1k < KNN) out[r1k] = j1;
    if (lane == 0) deg[i] = d;
    if (e_w) {
        float* ew_out = e_w + (size_t)i * KNN;
        if (lane >= d && lane < KNN) ew_out[lane] = 0.f;
        if (vs && rs < KNN) ew_out[rs] = ews;
        unsigned mk = ((v0 && r0k < KNN) ? 1u << r0k : 0u) | ((v1 && r1k < KNN) ? 1u << r1k : 0u);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mk |= (unsigned)__shfl_xor((int)mk, o, 64);
        if (lane == 0) newmask[i] = mk;
    }
}

// ------------------------------------------------------------------------------------------------
// gate: one wave per node; per half (16 edges) the hidden layer is 10 edge-major tiles: lane (c = edge, q),
// C row rho = 4q + r <-> hidden unit u = 16t + rho.  W1 is centred over its 160 outputs at pack time, so
// LayerNorm needs only sum(y^2).  Fragments [t 10][s 5][lane] = W1c[u = 16t + c][g = 4s + q].
// ------------------------------------------------------------------------------------------------
constexpr int GT = GH / 16;  // 10 tiles
#ifndef CBGX_GATE_WAVES_PER_EU
#define CBGX_GATE_WAVES_PER_EU 4
#endif
constexpr int GATE_WAVES_PER_EU = CBGX_GATE_WAVES_PER_EU;

// the gate value of this lane's edge (i, j) -- lane (c = edge of the tile, q), result valid in every q row -- from the LDS image
// (GATE_IMG layout); `mu` = the lane's five rbf centres 4 s + q.  The arithmetic of a column does not depend on the other columns
// of the tile, so any assignment of edges to lanes gives the same bits.
// The LDS image is the same for every tile, so the compiler hoists all of its reads out of the centre loop -- 210 registers of
// fragments and per-channel vectors, 330 in all: ONE wave per SIMD, and a centre is a chain of three or four dependent global round
// trips (list entry -> mask / degree / position -> neighbour index -> neighbour position) with nothing else to run meanwhile
// (157 us per 60 k listed centres of the headline batch, ~58 centres per wave in sequence).  With more than one wave per SIMD asked
// for (GATE_WAVES_PER_EU) every tile addresses the image through an offset the compiler cannot see through: the reads stay where
// they are used (40 KB of LDS reads per 16-edge tile) and the register budget is met without spilling.
__device__ __forceinline__ const float* gate_image_of_this_tile(const float* lds_img) {
    if (GATE_WAVES_PER_EU <= 1) return lds_img;
    int off = 0;
    asm volatile("" : "+v"(off));
    return lds_img + off;
}

__device__ __forceinline__ float gate_tile_value(const float* lds, const float (&mu)[5], float b2, int lane, int q,
                                                 float xi, float yi, float zi, float xj, float yj, float zj) {
    lds = gate_image_of_this_tile(lds);
    const float* l_frag = lds;
    const float* l_b1 = lds + GT * 5 * 64;
    const float* l_g = l_b1 + GH;
    const float* l_be = l_g + GH;
    const float* l_w2 = l_be + GH;
    const float rx = xi - xj, ry = yi - yj, rz = zi - zj;
    const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
    float R[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) { const float u = dist - mu[s]; R[s] = expf(-0.5f * (u * u)); }
    floatx4 acc[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(l_b1 + 16 * t + 4 * q);
        acc[t] = floatx4{b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int t = 0; t < GT; ++t) acc[t] = MFMA(l_frag[(t * 5 + s) * 64 + lane], R[s], acc[t]);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < GT; ++t)
        v += (acc[t].x * acc[t].x + acc[t].y * acc[t].y) + (acc[t].z * acc[t].z + acc[t].w * acc[t].w);
    v = g_xrow_sum(v);
    const float rstd = 1.f / sqrtf(v * (1.f / GH) + 1e-5f);
    float z = 0.f;
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        const float4 g = *reinterpret_cast<const float4*>(l_g + 16 * t + 4 * q);
        const float4 be = *reinterpret_cast<const float4*>(l_be + 16 * t + 4 * q);
        const float4 w2 = *reinterpret_cast<const float4*>(l_w2 + 16 * t + 4 * q);
        z = fmaf(fmaxf((acc[t].x * rstd) * g.x + be.x, 0.f), w2.x, z);
        z = fmaf(fmaxf((acc[t].y * rstd) * g.y + be.y, 0.f), w2.y, z);
        z = fmaf(fmaxf((acc[t].z * rstd) * g.z + be.z, 0.f), w2.z, z);
        z = fmaf(fmaxf((acc[t].w * rstd) * g.w + be.w, 0.f), w2.w, z);
    }
    z = g_xrow_sum(z) + b2;
    return 1.f / (1.f + expf(-z));
}

__global__ __launch_bounds__(256, GATE_WAVES_PER_EU) void edge_gate_mfma_kernel(const float* __restrict__ wts,
                                                             const float* __restrict__ x,
                                                             const int32_t* __restrict__ nbr,
                                                             const int32_t* __restrict__ deg, int n_nodes,
                                                             float* __restrict__ e_w, const int* __restrict__ rows,
                                                             const int* __restrict__ n_rows_ptr,
                                                             const unsigned* __restrict__ newmask) {
    __shared__ __attribute__((aligned(16))) float lds[GATE_IMG_SIZE];
    __shared__ int s_rank[4][KNN];      // newmask mode: the ranks to evaluate, compacted
    {   // LDS fill, all loads of the thread in flight together (a plain loop compiles to one dependent round trip per iteration)
        typedef float fx4 __attribute__((ext_vector_type(4)));
        constexpr int NV = ((int)GATE_IMG_SIZE / 4 + 255) / 256;
        fx4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int t = threadIdx.x + 256 * u;
            v[u] = reinterpret_cast<const fx4*>(wts + GATE_IMG)[t < (int)GATE_IMG_SIZE / 4 ? t : (int)GATE_IMG_SIZE / 4 - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int t = threadIdx.x + 256 * u;
            if (t < (int)GATE_IMG_SIZE / 4) reinterpret_cast<fx4*>(lds)[t] = v[u];
        }
    }
    __syncthreads();
    const float b2 = wts[GATE_B2];      // (image layout: frag [10][5][64] | centred bias | gamma | beta | w2: gate_tile_value)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    float mu[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) mu[s] = c_mu2[4 * s + q];
    const int count = rows ? *n_rows_ptr : n_nodes;
    for (int idx = blockIdx.x * 4 + wave; idx < count; idx += gridDim.x * 4) {
        const int i = __builtin_amdgcn_readfirstlane(rows ? rows[idx] : idx);
        const int d = deg[i];
        const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
        if (newmask) {
            // only the ranks knn_merge_kernel marked as new (a protein centre near the ligand: the ligand atoms that entered its
            // list, ~1 - 10 of 32; the others kept their cached values), compacted to the head of as few 16-edge tiles as they need
            const unsigned m = newmask[i];
            const int cnt = __popc(m);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < KNN && (m >> lane & 1u)) s_rank[wave][__popc(m & ((1u << lane) - 1u))] = lane;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int t0 = 0; t0 < cnt; t0 += 16) {
                const int k = t0 + c;
                const bool valid = k < cnt;
                const int rank = valid ? s_rank[wave][k] : 0;
                const int j = valid ? nbr[(size_t)i * KNN + rank] : i;
                const float gv = gate_tile_value(lds, mu, b2, lane, q, xi, yi, zi, x[3 * j], x[3 * j + 1], x[3 * j + 2]);
                if (q == 0 && valid) e_w[(size_t)i * KNN + rank] = gv;
            }
            continue;
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int e = c + 16 * hf;
            const bool valid = e < d;
            const int j = valid ? nbr[(size_t)i * KNN + e] : i;
            const float gv = gate_tile_value(lds, mu, b2, lane, q, xi, yi, zi, x[3 * j], x[3 * j + 1], x[3 * j + 2]);
            if (q == 0) e_w[(size_t)i * KNN + e] = valid ? gv : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Graph-cached calls, round 5: the listed centres' new neighbour lists AND their gate values in one launch.
// knn_merge_kernel + edge_gate_mfma_kernel re-evaluated the gate MLP on all 32 edges of every listed centre, although a protein
// centre's edges to protein atoms keep their cached values (the gate is a function of the distance between two atoms that never
// move: unitransformer.py:109-112): the pocket entry at position p of the static list lands at rank rs of the merged list and takes
// its value e_w[i][rs] = static e_w[i][p] with it, and the MLP runs only on the ligand atoms that entered the list -- compacted to
// the head of ONE 16-edge tile for most centres (a protein atom near the ligand has ~1 - 10 ligand neighbours) instead of two full
// tiles.  Ligand centres (and graphs beyond the merge's limits) take the scan and the two full tiles as before.  Same arithmetic
// per edge as edge_gate_mfma_kernel (gate_tile_value: a tile's columns are independent), so cached and uncached evaluations of a
// state still agree bit for bit.  Persistent 4-wave workgroups (the 15 KB image is filled once per workgroup).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, GATE_WAVES_PER_EU) void knn_merge_gate_kernel(
    const float* __restrict__ wts, const float* __restrict__ x, const int32_t* __restrict__ graph_ptr, int n_graphs,
    const uint8_t* __restrict__ lig, const int32_t* __restrict__ s_nbr, const int32_t* __restrict__ s_deg,
    const float* __restrict__ s_ew, int32_t* __restrict__ nbr, int32_t* __restrict__ deg, float* __restrict__ e_w,
    const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    __shared__ __attribute__((aligned(16))) float lds[GATE_IMG_SIZE];
    __shared__ int s_j[4][KNN], s_rank[4][KNN], s_sel[4][KNN];
    const int count = *n_rows_ptr;
    if ((int)blockIdx.x * 4 >= count) return;        // no centre for this workgroup: no fill either
    {   // LDS fill, all loads of the thread in flight together
        typedef float fx4 __attribute__((ext_vector_type(4)));
        constexpr int NV = ((int)GATE_IMG_SIZE / 4 + 255) / 256;
        fx4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int t = threadIdx.x + 256 * u;
            v[u] = reinterpret_cast<const fx4*>(wts + GATE_IMG)[t < (int)GATE_IMG_SIZE / 4 ? t : (int)GATE_IMG_SIZE / 4 - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int t = threadIdx.x + 256 * u;
            if (t < (int)GATE_IMG_SIZE / 4) reinterpret_cast<fx4*>(lds)[t] = v[u];
        }
    }
    __syncthreads();
    const float b2 = wts[GATE_B2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    float mu[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) mu[s] = c_mu2[4 * s + q];
    const unsigned NONE = 0xffffffffu;
    for (int idx = blockIdx.x * 4 + wave; idx < count; idx += gridDim.x * 4) {
        const int i = __builtin_amdgcn_readfirstlane(rows[idx]);
        // what depends on the centre alone: requested before the graph search, unconditionally (knn_merge_kernel)
        const unsigned lig_i = lig[i];
        const int sd_raw = s_deg[i];
        const int js_raw = s_nbr[(size_t)i * KNN + (lane & (KNN - 1))];
        const float ews = s_ew[(size_t)i * KNN + (lane & (KNN - 1))];
        const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
        int lo_g = 0, hi_g = n_graphs;
        while (hi_g - lo_g > 1) {
            const int mid = (lo_g + hi_g) >> 1;
            if (graph_ptr[mid] <= i) lo_g = mid; else hi_g = mid;
        }
        const int gs = graph_ptr[lo_g], ge = graph_ptr[lo_g + 1];
        // ligand atoms of the graph: the run of flagged rows at its end, counted over the last 128 rows
        const int r0 = ge - 1 - lane, r1 = ge - 65 - lane;
        const unsigned l0 = lig[r0 >= gs ? r0 : gs], l1 = lig[r1 >= gs ? r1 : gs];
        const unsigned long long b0 = __ballot((r0 >= gs) & (l0 != 0u));
        const unsigned long long b1 = __ballot((r1 >= gs) & (l1 != 0u));
        const int n0 = b0 == ~0ull ? 64 : __builtin_ctzll(~b0);
        const int n1 = n0 < 64 ? 0 : (b1 == ~0ull ? 64 : __builtin_ctzll(~b1));
        const int nl = n0 + n1;
        float* ew_out = e_w + (size_t)i * KNN;
        int jr;      // lane r < 32: the neighbour at rank r that needs a fresh gate value (-1: none)
        if (lig_i != 0u || nl >= 128 || ge - gs > 64 * KNN_SLOTS) {      // wave-uniform: the scan, every valid slot is new
            jr = knn_scan_node(x, graph_ptr, n_graphs, i, lane, nbr, deg);
            if (lane < KNN && jr < 0) ew_out[lane] = 0.f;
        } else {
            // ---- knn_merge_kernel's rank counting (same keys, same comparisons) ----
            const int sd = __builtin_amdgcn_readfirstlane(sd_raw);
            const bool vs = lane < sd && lane < KNN;
            const int js = vs ? js_raw : i;
            const int ls = ge - nl;
            const bool v0 = lane < nl, v1 = lane + 64 < nl;
            const int j0 = v0 ? ls + lane : i, j1 = v1 ? ls + 64 + lane : i;
            const float sx = x[3 * js], sy = x[3 * js + 1], sz = x[3 * js + 2];
            const float ax = x[3 * j0], ay = x[3 * j0 + 1], az = x[3 * j0 + 2];
            const float bx = x[3 * j1], by = x[3 * j1 + 1], bz = x[3 * j1 + 2];
            const unsigned ks_h = vs ? __float_as_uint(dist2_exact2(xi, yi, zi, sx, sy, sz)) : NONE;
            const unsigned ks_l = vs ? (unsigned)js : NONE;
            const unsigned k0_h = v0 ? __float_as_uint(dist2_exact2(xi, yi, zi, ax, ay, az)) : NONE;
            const unsigned k1_h = v1 ? __float_as_uint(dist2_exact2(xi, yi, zi, bx, by, bz)) : NONE;
            const unsigned k0_l = v0 ? (unsigned)j0 : NONE, k1_l = v1 ? (unsigned)j1 : NONE;
            int rs = lane, r0k = 0, r1k = 0;      // ranks in the union
            for (int t = 0; t < min(nl, 64); ++t) {
                const unsigned bh = __builtin_amdgcn_readlane(k0_h, t), bl = __builtin_amdgcn_readlane(k0_l, t);
                rs += key_less(bh, bl, ks_h, ks_l) ? 1 : 0;
                r0k += key_less(bh, bl, k0_h, k0_l) ? 1 : 0;
                r1k += key_less(bh, bl, k1_h, k1_l) ? 1 : 0;
            }
            for (int t = 0; t < nl - 64; ++t) {
                const unsigned bh = __builtin_amdgcn_readlane(k1_h, t), bl = __builtin_amdgcn_readlane(k1_l, t);
                rs += key_less(bh, bl, ks_h, ks_l) ? 1 : 0;
                r0k += key_less(bh, bl, k0_h, k0_l) ? 1 : 0;
                r1k += key_less(bh, bl, k1_h, k1_l) ? 1 : 0;
            }
            for (int p = 0; p < min(sd, KNN); ++p) {
                const unsigned bh = __builtin_amdgcn_readlane(ks_h, p), bl = __builtin_amdgcn_readlane(ks_l, p);
                r0k += key_less(bh, bl, k0_h, k0_l) ? 1 : 0;
                r1k += key_less(bh, bl, k1_h, k1_l) ? 1 : 0;
            }
            const int d = min(KNN, min(sd, KNN) + nl);
            int32_t* out = nbr + (size_t)i * KNN;
            if (lane >= d && lane < KNN) { out[lane] = -1; ew_out[lane] = 0.f; }
            if (vs && rs < KNN) { out[rs] = js; ew_out[rs] = ews; }      // a pocket entry keeps its gate value
            if (v0 && r0k < KNN) out[r0k] = j0;
            if (v1 && r1k < KNN) out[r1k] = j1;
            if (lane == 0) deg[i] = d;
            // the ligand atoms that entered the list, by rank
            if (lane < KNN) s_j[wave][lane] = -1;
            gm_wave_sync();
            if (v0 && r0k < KNN) s_j[wave][r0k] = j0;
            if (v1 && r1k < KNN) s_j[wave][r1k] = j1;
            gm_wave_sync();
            jr = lane < KNN ? s_j[wave][lane] : -1;
        }
        // ---- gate values of the new entries: compacted to the head of as few 16-edge tiles as they need ----
        const bool need = lane < KNN && jr >= 0;
        const unsigned m = (unsigned)(__ballot(need) & 0xffffffffull);
        const int cnt = __popc(m);
        if (need) {
            const int pos = __popc(m & ((1u << lane) - 1u));
            s_rank[wave][pos] = lane;
            s_sel[wave][pos] = jr;
        }
        gm_wave_sync();
        for (int t0 = 0; t0 < cnt; t0 += 16) {
            const int k = t0 + c;
            const bool valid = k < cnt;
            const int j = valid ? s_sel[wave][k] : i;
            const int rank = valid ? s_rank[wave][k] : 0;
            const float gv = gate_tile_value(lds, mu, b2, lane, q, xi, yi, zi, x[3 * j], x[3 * j + 1], x[3 * j + 2]);
            if (q == 0 && valid) ew_out[rank] = gv;
        }
        gm_wave_sync();      // the per-wave arrays are rewritten by the wave's next centre
    }
}

hipError_t launch_knn_merge_gate(const float* packed, const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes,
                                 const uint8_t* lig, const int32_t* s_nbr, const int32_t* s_deg, const float* s_ew, int32_t* nbr,
                                 int32_t* deg, float* e_w, hipStream_t s, const int* rows, const int* n_rows) {
    if (n_nodes == 0) return hipSuccess;
    int grid = (n_nodes + 3) / 4;
    if (grid > 2048) grid = 2048;
    profile_mark_begin(K_KNN, s);
    hipLaunchKernelGGL(knn_merge_gate_kernel, dim3(grid), dim3(256), 0, s, packed, x, graph_ptr, n_graphs, lig, s_nbr, s_deg, s_ew,
                       nbr, deg, e_w, rows, n_rows);
    profile_mark_end(s);
    return hipGetLastError();
}

// gate LDS image: frag [10][5][64] | b1c [160] | gamma [160] | beta [160] | w2 [160]
__global__ void pack_gate_img_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                     const float* __restrict__ g, const float* __restrict__ be,
                                     const float* __restrict__ w2, float* __restrict__ img) {
    __shared__ float colmean[G + 1];
    if (threadIdx.x <= G) {
        float s = 0.f;
        for (int u = 0; u < GH; ++u) s += threadIdx.x < G ? w1[u * G + threadIdx.x] : b1[u];
        colmean[threadIdx.x] = s * (1.f / GH);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < GT * 5 * 64; idx += blockDim.x) {
        const int lane = idx & 63, s = (idx >> 6) % 5, t = idx / 320;
        const int u = 16 * t + (lane & 15), gg = 4 * s + (lane >> 4);
        img[idx] = w1[u * G + gg] - colmean[gg];
    }
    float* p = img + GT * 5 * 64;
    for (int u = threadIdx.x; u < GH; u += blockDim.x) {
        p[u] = b1[u] - colmean[G];
        p[GH + u] = g[u];
        p[2 * GH + u] = be[u];
        p[3 * GH + u] = w2[u];
    }
}

hipError_t launch_pack_gate_img(const float* w1, const float* b1, const float* g, const float* be, const float* w2,
                                float* img, hipStream_t s) {
    hipLaunchKernelGGL(pack_gate_img_kernel, dim3(1), dim3(256), 0, s, w1, b1, g, be, w2, img);
    return hipGetLastError();
}

// (the proximity test itself: dirty[i] = 1 for ligand atoms and for protein atoms that have a ligand atom among their 32 nearest
// neighbours -- the nearest ligand atom of the graph (ligand rows close every graph, common.py:200) is closer than the cached
// distance to the 32nd protein neighbour; ties go to the protein atom (smaller index), hence the strict comparison)
// The proximity flags, their compaction into the D1 list, and the copy of the pocket's own neighbour lists, degrees, gate values and
// cached features of layers 0 / 1 into the call's working arrays, in ONE launch (until round 5: lig_proximity_kernel,
// build_active_kernel, restore_graph_kernel and two device-to-device copies): the head of every
// graph-cached forward call was these three dependent launches (9 + 5 + 5 us of a 600 us one-graph step).  1024-thread workgroups:
// the first ceil(n / 1024) of them also flag their nodes and append the flagged ones to `list` with one returning atomic each
// (*count zero on entry, like build_active_kernel); every workgroup copies its slice.
__global__ __launch_bounds__(1024) void graph_cache_begin_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ graph_ptr, int n_graphs, const uint8_t* __restrict__ lig,
    const float* __restrict__ r32sq, int n, uint8_t* __restrict__ dirty, int* __restrict__ list, int* __restrict__ count,
    const int32_t* __restrict__ s_nbr, const int32_t* __restrict__ s_deg, const float* __restrict__ s_ew,
    int32_t* __restrict__ nbr, int32_t* __restrict__ deg, float* __restrict__ ew, const float* __restrict__ h1,
    const float* __restrict__ h2, float* __restrict__ out1, float* __restrict__ out2) {
    __shared__ int s_cnt[16];
    __shared__ int s_base;
    const long t = (long)blockIdx.x * 1024 + threadIdx.x;
    // ---- copies (independent of everything below) ----
    if (t < (long)n * (KNN / 4)) {      // one thread per 4 neighbour slots
        reinterpret_cast<int4*>(nbr)[t] = reinterpret_cast<const int4*>(s_nbr)[t];
        reinterpret_cast<float4*>(ew)[t] = reinterpret_cast<const float4*>(s_ew)[t];
        if ((t & (KNN / 4 - 1)) == 0) deg[t / (KNN / 4)] = s_deg[t / (KNN / 4)];
    }
    if (t < (long)n * (H / 4)) {        // one thread per 4 features
        reinterpret_cast<float4*>(out1)[t] = reinterpret_cast<const float4*>(h1)[t];
        reinterpret_cast<float4*>(out2)[t] = reinterpret_cast<const float4*>(h2)[t];
    }
    if ((long)blockIdx.x * 1024 >= n) return;       // workgroup-uniform: no node of its own
    // ---- proximity flag of node t (lig_proximity_kernel) ----
    const int i = (int)t;
    bool a = false;
    if (i < n) {
        if (lig[i]) a = true;
        else {
            // the graph of the workgroup's first node by a scalar search, then a few steps forward per thread (1024 consecutive
            // nodes span two or three graphs): a per-thread search was nine dependent vector round trips at 340 graphs
            int g = 0;
            {
                const int first = (int)blockIdx.x * 1024;
                int lo_g = 0, hi_g = n_graphs;
                while (hi_g - lo_g > 1) {
                    const int mid = (lo_g + hi_g) >> 1;
                    if (graph_ptr[mid] <= first) lo_g = mid; else hi_g = mid;
                }
                g = lo_g;
            }
            while (g + 1 < n_graphs && graph_ptr[g + 1] <= i) ++g;
            const int gs = graph_ptr[g], ge = graph_ptr[g + 1];
            const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
            const float lim = r32sq[i];
            // the ligand atoms close the graph's rows: walked from the end four at a time, every load of a step unconditional
            // (clamped) and in flight together -- one atom per step was ~25 dependent round trips per thread
            bool alive = true;
            for (int j = ge - 1; alive && j >= gs; j -= 4) {
                unsigned lf[4];
                float px[4], py[4], pz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j - u >= gs ? j - u : gs;
                    lf[u] = lig[jj];
                    px[u] = x[3 * jj]; py[u] = x[3 * jj + 1]; pz[u] = x[3 * jj + 2];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    alive = alive & (j - u >= gs) & (lf[u] != 0u);
                    a |= alive & (dist2_exact2(xi, yi, zi, px[u], py[u], pz[u]) < lim);
                }
            }
        }
        dirty[i] = a ? 1 : 0;
    }
    // ---- compaction (build_active_kernel) ----
    const unsigned long long m = __ballot(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = s_cnt[w]; s_cnt[w] = tot; tot += c; }   // exclusive prefix
        s_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (a) list[s_base + s_cnt[wave] + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

hipError_t launch_graph_cache_begin(const float* x, const int32_t* graph_ptr, int n_graphs, const uint8_t* lig, const float* r32sq,
                                    int n, uint8_t* dirty, int* list, int* count, const int32_t* s_nbr, const int32_t* s_deg,
                                    const float* s_ew, int32_t* nbr, int32_t* deg, float* ew, const float* h1, const float* h2,
                                    float* out1, float* out2, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const long threads = (long)n * (H / 4);
    hipLaunchKernelGGL(graph_cache_begin_kernel, dim3((unsigned)((threads + 1023) / 1024)), dim3(1024), 0, s, x, graph_ptr, n_graphs,
                       lig, r32sq, n, dirty, list, count, s_nbr, s_deg, s_ew, nbr, deg, ew, h1, h2, out1, out2);
    return hipGetLastError();
}

hipError_t launch_knn_reg(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int32_t* nbr,
                          int32_t* deg, hipStream_t s, const int* rows, const int* n_rows) {
    if (n_nodes == 0) return hipSuccess;
    profile_mark_begin(K_KNN, s);
    hipLaunchKernelGGL(knn_graph_reg_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, s, x, graph_ptr, n_graphs, n_nodes,
                       nbr, deg, rows, n_rows);
    profile_mark_end(s);
    return hipGetLastError();
}

hipError_t launch_knn_merge(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, const uint8_t* lig,
                            const int32_t* s_nbr, const int32_t* s_deg, int32_t* nbr, int32_t* deg, hipStream_t s, const int* rows,
                            const int* n_rows, const float* s_ew, float* e_w, unsigned* newmask) {
    if (n_nodes == 0) return hipSuccess;
    profile_mark_begin(K_KNN, s);
    hipLaunchKernelGGL(knn_merge_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, s, x, graph_ptr, n_graphs, lig, s_nbr, s_deg, nbr,
                       deg, rows, n_rows, s_ew, e_w, newmask);
    profile_mark_end(s);
    return hipGetLastError();
}

hipError_t launch_gate_mfma(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                            float* e_w, hipStream_t s, const int* rows, const int* n_rows, const unsigned* newmask) {
    if (n_nodes == 0) return hipSuccess;
    int grid = (n_nodes + 3) / 4;
    if (grid > 2048) grid = 2048;
    profile_mark_begin(K_GATE, s);
    hipLaunchKernelGGL(edge_gate_mfma_kernel, dim3(grid), dim3(256), 0, s, packed, x, nbr, deg, n_nodes, e_w, rows, n_rows, newmask);
    profile_mark_end(s);
    return hipGetLastError();
}

}  // namespace cbgx

// libcbgx -- third-generation backward of the X2H attention block: ONE WAVEFRONT PER DESTINATION NODE, tiles in registers.
//
// Same contract as edge_backward_mfma_kernel<true> (train_bwd_mfma.hip; autograd of x2h_attention.py:43-97 for one node
// and its <= 32 incoming edges, nothing per-edge ever stored by the forward), different structure.  The second-generation
// kernel spread a node over an 8-wave workgroup with ten barrier-separated phases and 135 KB of LDS tiles (one node in
// flight per CU, 13.4 k vector instructions per node, waves waiting half of their cycles).  Here a wave does what the
// forward kernel (edge_mfma.hip) does -- the [32 x 128] tiles of a path live in MFMA accumulator layout, LayerNorm and
// softmax are DPP / permlane reductions -- and LDS serves one purpose: turning a tile from the edge-major labeling
//   E: lane (c = edge c + 16 hf, q), register (t, r) <-> channel 16 t + 4 q + r        (contractions over channels)
// into the channel-major one
//   C: lane (c, q), (t, hf, r)                       <-> channel 16 t + c, edge 4 q + r + 16 hf   (contractions over edges)
// through a wave-private [32][132] buffer.  A node runs three phases through ONE loop body (docs/x2h_backward.md):
//   0  key path forward -> scores -> alpha; the normalised pre-activation n_k is parked in the wave's 16 KB scratch slot
//   1  value path forward -> G . v_raw -> d alpha, d e_w, w = alpha e_w; backward of the value path
//   2  n_k read back; backward of the key path with d score = alpha (d alpha - sum alpha d alpha)
// and the backward of a path is four passes over the tile:
//   forward (phases 0, 1): pre-activation, split-f16 as the forward kernel   64 MFMAs  E    n in 64 registers
//                          scores / G . v_raw = hid . Qt | Gt                 64        E
//   pass 1  fold T | S[a][m] = sum_e w[e][a] hid[e][m], d hid = w . Qt | Gt   64 + 64   C    LayerNorm affine gradients, the two
//           (labeling: channel 32 u + 2 c + j -- 8-byte tile reads, float2 operand loads)    per-edge sums of LayerNorm's backward
//   pass 2  d hid again (cheaper than 64 live registers of it) -> d pre, in place in the tile        64        C
//   pass 3  d dist[e] = sum_m d pre[e][m] V[e][m], V = rbf'(d_e) . Wr[type_e]: the forward's split-f16 rbf product with
//           rbf' = -(d - mu) rbf (round 6; was d rbf = d pre . Wr^T, 128 fp32 MFMAs + 16 weight loads per lane)   64 (f16)  E
//   pass 4  every atomic of the path in one burst (labeling: channel 16 t + c, so a row's 16 lanes hit one 64-byte run):
//           d PS[j_e] += d pre[e], d PD[i], type columns; d Wr[type] += rbf^T . d pre            64 + 64 (4x4x1)   C
//           -> LDS slab (type 3) / the workgroup's slab in memory, flushed one step late
// What bounds it (round 6, profiles/abl_bwd_r06c.log, profiles/ubench_vmem_r06d.log): not the matrix pipe -- removing 192 of the 528
// fp32 MFMAs per node changed nothing -- but the compute unit's fp32-atomic path: a 64-lane global_atomic_add_f32 costs 57 - 65 ns of it
// whatever its shape (a dword store 5.5 ns), a node issues 136 of them, 64.5 nodes per CU = 0.53 ms of the 0.83 ms launch.
// Per-edge scalars (edge length, neighbour index, d dist, the rbf centres) live in the four pad columns of the tile rows, the LayerNorm
// affine in LDS.  8 waves per workgroup (2 per SIMD, 256 VGPRs), persistent, one workgroup per CU, XCD-aware node partition; LDS =
// 8 x 16.9 KB tiles + 20 KB type-3 d Wr + 8 KB of small slabs = 160 KB.  Outputs, slab layout (train.h PB_*) and launch contract are
// those of launch_edge_backward_mfma, so api_train.hip swaps one call.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "edge_common.h"
#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// libcbgx_ablate.so only: cycle counters per section of the kernel (s_memtime at the section boundaries, summed over all waves),
// printed by the launcher; switched on by bit 512 of CBGX_BWD_ABL
#ifdef CBGX_ABLATE
__device__ unsigned long long g_bx_prof[16];
#define BX_T(k)                                                              \
    do {                                                                     \
        if (abl & 512) {                                                     \
            SCHED_FENCE();                                                   \
            const unsigned long long now_ = __builtin_readcyclecounter();    \
            if (lane0 == 0) atomicAdd(&g_bx_prof[k], now_ - tprev_);         \
            tprev_ = now_;                                                   \
            SCHED_FENCE();                                                   \
        }                                                                    \
    } while (0)
#else
#define BX_T(k) do { } while (0)
#endif

__constant__ float c_mu_x[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

#ifndef CBGX_BX_WAVES
#define CBGX_BX_WAVES 8      // experiments: scripts/build_variant.py <name> -DCBGX_BX_WAVES=4 (one wave per SIMD, 512 registers)
#endif
constexpr int BX_WAVES = CBGX_BX_WAVES;
#ifndef CBGX_BX_DYN_ROUNDS
#define CBGX_BX_DYN_ROUNDS 2     // rounds of nodes handed out dynamically at the end of a launch: the partial round and one full one
                                 // (A/B, scripts/build_variant.py: 1 -> 851, 2 -> 829, 3 -> 845 us per 16.5 k-node launch)
#endif
constexpr int BX_DYN_ROUNDS = CBGX_BX_DYN_ROUNDS;
#ifndef CBGX_BX_SWEEP
#define CBGX_BX_SWEEP 0     // 1: passes 2 + 3 as one E-labeling sweep over v (pass 1 overwrites n with v, n re-read from the scratch slot:
                            //    no second d hidden, but 48 KB more scratch traffic per node); 0: pass 2 in place in C labeling (d hidden
                            //    evaluated again: 64 MFMAs, no memory traffic), pass 3 reads d pre.  Round 6, profiles/abl_bwd_r06c.log:
                            //    the 192 fp32 MFMAs that variant 1 removes per node bought nothing (851 vs 855 us), its scratch traffic costs
                            //    87 us -- this kernel is bound by its memory operations, not by the matrix pipe
#endif
#ifndef CBGX_BX_PARK
#define CBGX_BX_PARK 1      // 1: the key path's normalised pre-activation is parked in the wave's scratch slot in phase 0 and read back in phase 2;
                            // 0: phase 2 gathers, multiplies and normalises it again (A/B knob, scripts/build_variant.py)
#endif
#ifndef CBGX_BX_Y4X4
#define CBGX_BX_Y4X4 1      // rbf columns g = 16..19 of d Wr on v_mfma_f32_4x4x1 (16 blocks of 4 x 4, 8 cycles) instead of a 16x16x4 tile
                            // of which 4 of 16 rows are used (32 cycles): see pass 4
#endif
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
constexpr int BX_PITCH = H + 4;                 // transpose tile row pitch (floats): 16-byte aligned rows
constexpr int BX_TILE = KNN * BX_PITCH;         // 4224 floats per wave
// the four pad columns of a tile row hold per-edge scalars of the node (row = edge) instead of living in registers across the
// phases: edge length, neighbour index, d L / d dist; column 131 of rows 0..19 holds the rbf centres, of rows 20..27 of tiles 0 and 1
// the 16 locks of the d Wr slab
constexpr int BX_DIST = H, BX_NBR = H + 1, BX_DDIST = H + 2, BX_MU = H + 3;
// per-wave slot of the scratch buffer that parks the key path between phase 0 and phase 2: n [2][8][64 lanes][4] + rstd [64][2]
constexpr int BX_NK_SLOT = KNN * H + 128;

struct BwdX2hLds {
    float tile[BX_WAVES][BX_TILE];              // per-wave E <-> C transposes (and the 32 x 16 E1 transposes)
    float dwr3[G][2 * H];                       // d Wr of the dominant edge type 3: read-modify-write under 16 column-block locks
    float wt[NT][2 * H];                        // type-column gradients
    float lng[2 * H], lnb[2 * H];               // LayerNorm affine gradients (k | v)
    float ln[4][H];                             // LayerNorm affine itself: k gamma, k beta, v gamma, v beta
};

// accesses of the wave's tile by different lanes are ordered by the LDS queue (one wave, in issue order); the fence keeps the
// compiler from moving a lane's reads above another lane's writes it cannot see
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// wave-level LDS lock: lane 0 spins on the word, the other lanes wait at the reconvergence point
__device__ __forceinline__ void lds_lock_w(int* lk, int lane) {
    if (lane == 0) {
        int expected = 0;
        while (!__hip_atomic_compare_exchange_strong(lk, &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            expected = 0;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_unlock_w(int* lk, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(lk, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }

// ER (round 6): the neighbour-row gradient of every edge goes to its own row of `dE` [N][32][256] by plain stores (a 64-lane dword
// store costs the CU 5.5 ns, the atomic it replaces 57 - 65) and edge_rows_reduce_kernel (train_scatter.hip) sums the rows of every
// SOURCE node over its incoming edges in a fixed order: no atomics on dP at all, PS columns reproducible bit for bit.  Full launches
// only; a listed launch (pruned last layers) keeps the atomics (ER = false).
template <bool ER>
__global__ __launch_bounds__(BX_WAVES * 64) void edge_backward_x2h_kernel(
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ P,
    const float* __restrict__ Qt, const float* __restrict__ Gt, const float* __restrict__ gb,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg, const uint8_t* __restrict__ lig,
    const float* __restrict__ e_w, const int* __restrict__ rows, const int* __restrict__ n_rows_ptr, int n_nodes,
    float* __restrict__ T, float* __restrict__ S, float* __restrict__ sw, float* __restrict__ dP, float* __restrict__ dx,
    float* __restrict__ de_w, float* __restrict__ partial, float* __restrict__ nk_scratch, int* __restrict__ work_ctr,
    float* __restrict__ dE
#ifdef CBGX_ABLATE
    , int abl
#endif
    ) {
#ifndef CBGX_ABLATE
    constexpr int abl = 0;      // timing ablations (WRONG results) exist only in libcbgx_ablate.so
#endif
    __shared__ BwdX2hLds L;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = lane0 & 15, q0 = lane0 >> 4;
    float* slab = partial + (size_t)blockIdx.x * PB_SIZE;
    // the workgroup's slab is (re)written in full: the LDS-accumulated parts at the end, types 0..2 of d Wr by atomics
    for (int u = tid; u < 3 * G * 2 * H; u += BX_WAVES * 64) slab[PB_WR + u] = 0.f;
    for (int u = tid; u < G * 2 * H; u += BX_WAVES * 64) (&L.dwr3[0][0])[u] = 0.f;
    for (int u = tid; u < NT * 2 * H; u += BX_WAVES * 64) (&L.wt[0][0])[u] = 0.f;
    for (int u = tid; u < 2 * H; u += BX_WAVES * 64) { L.lng[u] = 0.f; L.lnb[u] = 0.f; }
    for (int u = tid; u < 4 * H; u += BX_WAVES * 64) (&L.ln[0][0])[u] = att[A_LNK_G + u];
    if (tid < 16) reinterpret_cast<int*>(&L.tile[tid >> 3][0])[(20 + (tid & 7)) * BX_PITCH + BX_MU] = 0;     // the slab locks
    __threadfence();
    __syncthreads();
    float* tw = L.tile[wave];
#ifdef CBGX_ABLATE
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    const int count = rows ? *n_rows_ptr : n_nodes;
    if (lane0 < G) tw[lane0 * BX_PITCH + BX_MU] = c_mu_x[lane0];
    // A node's header (index, degree, class, neighbour list) is fetched one node ahead: its three dependent round trips
    // (rows -> deg / nbr -> coordinates) would otherwise open every node
    // XCD-aware persistent schedule (as the forward kernel): workgroup b runs on XCD b % 8, so every XCD gets one contiguous
    // eighth of the work list -- a graph's projection rows and their gradient rows then live in one L2
    // Static rounds, dynamic remainder: a training batch is ~8 nodes per wave (16.5 k nodes on 2048 waves), so with a purely
    // static partition the waves that own a ninth node set the kernel time (9 rounds for 8.06 nodes per wave on average).  Every
    // wave runs the rounds that are full for its XCD except the last BX_DYN_ROUNDS - 1 of them; the nodes of those and of the
    // partial round are handed out through one counter per XCD (work_ctr [8], zeroed by the caller) to the waves that finish first.
    int it, it_end, stride, full_rounds, tail_base;
    int* ctr;
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (((count + 7) >> 3) + BX_WAVES - 1) / BX_WAVES * BX_WAVES;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int base = min(count, xcd * per_xcd);
        it_end = min(count, (xcd + 1) * per_xcd);
        stride = (gridDim.x >> 3) * BX_WAVES;
        it = base + slot * BX_WAVES + wave;
        full_rounds = max((it_end - base) / stride - (BX_DYN_ROUNDS - 1), 0);
        tail_base = base + full_rounds * stride;
        ctr = work_ctr + xcd;
    } else {
        it = blockIdx.x * BX_WAVES + wave;
        it_end = count;
        stride = gridDim.x * BX_WAVES;
        full_rounds = max(count / stride - (BX_DYN_ROUNDS - 1), 0);
        tail_base = full_rounds * stride;
        ctr = work_ctr;
    }
    auto grab = [&]() {       // next node of the partial round (wave-uniform); it >= it_end when none is left
        int k = 0;
        if (lane0 == 0) k = atomicAdd(ctr, 1);
        it = tail_base + __builtin_amdgcn_readfirstlane(k);
    };
    int i_n = 0, d_n = 0, lig_n = 0, jr_n[2] = {0, 0};
    auto load_header = [&](int itx) {
        i_n = __builtin_amdgcn_readfirstlane(rows ? rows[itx] : itx);
        d_n = deg[i_n];
        lig_n = lig[i_n];
        jr_n[0] = nbr[(size_t)i_n * KNN + c0];
        jr_n[1] = nbr[(size_t)i_n * KNN + 16 + c0];
    };
    int round = 0;
    if (full_rounds == 0) grab();
    if (it < it_end) load_header(it);
    while (it < it_end) {
        const int i = i_n;
        const int d = __builtin_amdgcn_readfirstlane(d_n), lig_i = __builtin_amdgcn_readfirstlane(lig_n);
        const int jr[2] = {jr_n[0], jr_n[1]};
        const bool next_static = round + 1 < full_rounds;      // the next node is known: its header travels behind this node's work
        if (next_static) load_header(it + stride);
        // the lane coordinates are re-materialised every node: otherwise every per-lane address below is loop-invariant, gets
        // hoisted out of the node loop as a 64-bit pointer pair and spilled
        int lane = lane0, c = c0, q = q0;
        asm volatile("" : "+v"(lane), "+v"(c), "+v"(q));
        const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
        // ---- geometry.  E0: lane (c, q) <-> edges c, c + 16;  E1: lane (c, q), [hf][r] <-> edge 4q + r + 16 hf ---------------------
        int j0[2];
        bool lg0[2];
        float dist0[2];
        {
            float xj[2][3];
            int lj[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {   // unconditional gathers (padded slots read the node itself), all issued together
                j0[hf] = c + 16 * hf < d ? jr[hf] : i;
                lj[hf] = lig[j0[hf]];
#pragma unroll
                for (int k = 0; k < 3; ++k) xj[hf][k] = x[3 * j0[hf] + k];
            }
            SCHED_FENCE();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                lg0[hf] = (c + 16 * hf < d) & (lj[hf] != 0);
                dist0[hf] = edge_len(xi, yi, zi, xj[hf][0], xj[hf][1], xj[hf][2]);
            }
        }
        const unsigned long long b0 = __ballot(lg0[0]), b1 = __ballot(lg0[1]);
        const unsigned mask_lig = (unsigned)(b0 & 0xffffull) | ((unsigned)(b1 & 0xffffull) << 16);
        const unsigned mask_valid = d >= 32 ? 0xffffffffu : ((1u << d) - 1u);
        const bool has_lig = (mask_lig & mask_valid) != 0;
        const bool has_prot = ((~mask_lig) & mask_valid) != 0 || d == 0;
        const bool mixed = has_lig && has_prot;
        const unsigned msh_n = mask_lig >> (4 * q);       // bit r + 16 hf <-> E1 slot (hf, r)
        const unsigned vsh_n = mask_valid >> (4 * q);
        const bool val0[2] = {c < d, c + 16 < d};
        // per-edge scalars into the tile's pad columns (rows = edges); the E1 mapping reads them back where it needs them
        wave_sync();
        if (q < 2) {
            const int e = c + 16 * q;
            tw[e * BX_PITCH + BX_DIST] = q ? dist0[1] : dist0[0];
            reinterpret_cast<int*>(tw)[e * BX_PITCH + BX_NBR] = q ? j0[1] : j0[0];
            tw[e * BX_PITCH + BX_DDIST] = 0.f;
        }
        wave_sync();
        // rbf of the E1 edges (lane (c, q), [hf][r] <-> edge 4q + r + 16 hf) at g = c (tile 0) and g = 16 + c (tile 1, lanes c < 4):
        // A operand of d Wr, factor of d dist.  Recomputed where needed (16 v_exp) instead of living in registers across the phases;
        // u0 / u1 = dist - mu come back as well
        auto rbf_e1 = [&](unsigned vsh, float (&r0)[2][4], float (&r1)[2][4], float (&u0)[2][4], float (&u1)[2][4]) {
            const float mu0 = tw[c * BX_PITCH + BX_MU], mu1 = tw[(16 + (c & 3)) * BX_PITCH + BX_MU];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float vm = ((vsh >> (r + 16 * hf)) & 1u) ? 1.f : 0.f;
                    const float dist = tw[(4 * q + r + 16 * hf) * BX_PITCH + BX_DIST];
                    u0[hf][r] = dist - mu0;
                    u1[hf][r] = dist - mu1;
                    r0[hf][r] = fast_exp(-0.5f * (u0[hf][r] * u0[hf][r])) * vm;
                    // (CBGX_BX_Y4X4: every lane carries g = 16 + (c & 3) -- the A operand of its 4 x 4 block, see pass 4)
                    r1[hf][r] = fast_exp(-0.5f * (u1[hf][r] * u1[hf][r])) * ((CBGX_BX_Y4X4 || c < 4) ? vm : 0.f);
                }
        };
        BX_T(0);
        const int ty_prot = lig_i ? 2 : 3, ty_lig = lig_i ? 0 : 1;
        float alpha[2][4];       // softmax weights; after phase 1: d L / d score

        const int p1 = has_prot ? 0 : 1;       // first (usually only) source class present
        // three phases through one body: 0 = key forward (scores -> alpha), 1 = value forward + backward, 2 = key forward
        // again + backward
#pragma unroll 1
        for (int ph = 0; ph < 3; ++ph) {
            if ((abl & 32) && ph == 2) break;
            const int kv = ph == 1;
            // everything derived from these is phase-invariant and would be hoisted out of the phase loop into ~60 live registers
            unsigned msh = msh_n, vsh = vsh_n;
            asm volatile("" : "+v"(lane), "+v"(c), "+v"(q), "+v"(dist0[0]), "+v"(dist0[1]), "+v"(j0[0]), "+v"(j0[1]), "+v"(msh), "+v"(vsh));
            const float* Brow = (kv ? Gt : Qt) + (size_t)i * HEADS * H;
            const float* lng = L.ln[2 * kv];
            const float* lnb = L.ln[2 * kv + 1];
            // ---- forward of the path, E labeling: n[hf][t][r] <-> edge c + 16 hf, channel 16 t + 4 q + r -------------------------------
            floatx4 n[2][8];
            float rstd[2];
            float4 br[8];
            const float* frag = att + (kv ? A_FRAGV_EM : A_IMG + IMG_FRAG_K);
            const RbfScale rsc = load_rbf_scale(att, kv);      // the tile is carried scaled by S until LayerNorm (edge_common.h)
            // the key path's normalised pre-activation is parked in the wave's slot of a scratch buffer in phase 0 and read back in
            // phase 2 (16 coalesced KB each way) instead of being gathered, multiplied and normalised a second time
            // (two slots per wave, key | value: passes 2 + 3 below read the path's n back in the labeling the forward left it in)
            float* nk = nk_scratch + (((size_t)blockIdx.x * BX_WAVES + wave) * 2 + kv) * BX_NK_SLOT + 4 * lane;
            if (CBGX_BX_PARK && ph == 2) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int t = 0; t < 8; ++t) n[hf][t] = f4(ld4(nk + (hf * 8 + t) * 256));
                const float2 rs = ld2(nk + 16 * 256 - 2 * lane);
                rstd[0] = rs.x; rstd[1] = rs.y;
#pragma unroll
                for (int t = 0; t < 8; ++t) br[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                SCHED_FENCE();
            } else {
                WTuples wt[8];
                // every gather of the stage is issued before its first use (the scheduler otherwise serialises load -> wait -> add
                // to save registers: one memory round trip per row chunk)
                const float* pdp = P + (size_t)i * PROW + kv * H + 4 * q;
                const float* psa = P + (size_t)j0[0] * PROW + (2 + kv) * H + 4 * q;
                const float* psb = P + (size_t)j0[1] * PROW + (2 + kv) * H + 4 * q;
                float4 pd[8], pa[8], pb[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) pd[t] = (abl & 128) ? make_float4(1.f, 2.f, 3.f, 4.f) : ld4(pdp + 16 * t);
#pragma unroll
                for (int t = 0; t < 8; ++t) pa[t] = (abl & 128) ? make_float4(1.f, 2.f, 3.f, 4.f) : ld4(psa + 16 * t);
#pragma unroll
                for (int t = 0; t < 8; ++t) pb[t] = (abl & 128) ? make_float4(1.f, 2.f, 3.f, 4.f) : ld4(psb + 16 * t);
                BX_T(11);
                // the same round trip brings the rbf weight tuples of the first source class
                {
                    const float* fa = frag + (size_t)etype(p1 == 1, lig_i) * (8 * FRAG_BLK);
#pragma unroll
                    for (int t = 0; t < 8; ++t) wt[t] = load_wtuples(fa, t, lane);
                }
                BX_T(10);
                SCHED_FENCE();
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const floatx4 pds = f4(pd[t]) * rsc.S;
                    n[0][t] = f4(pa[t]) * rsc.S + pds; n[1][t] = f4(pb[t]) * rsc.S + pds;
                }
                // the contraction's B rows: in flight behind the rbf MFMAs and the LayerNorm (phase 2 has no contraction; loading
                // them anyway keeps the array from being carried around the phase loop)
#pragma unroll
                for (int t = 0; t < 8; ++t) br[t] = ld4(Brow + (size_t)c * H + 16 * t + 4 * q);
                SCHED_FENCE();
                if (has_lig) {      // type column of ligand sources (P already holds the protein-source one)
                    const float* dw = att + A_IMG + IMG_WT + lig_i * 2 * H + kv * H + 4 * q;
                    const float m0 = lg0[0] ? rsc.S : 0.f, m1 = lg0[1] ? rsc.S : 0.f;
                    float4 w4[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) w4[t] = ld4(dw + 16 * t);
                    SCHED_FENCE();
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        n[0][t] += f4(w4[t]) * m0;
                        n[1][t] += f4(w4[t]) * m1;
                    }
                }
                // rbf columns, split-f16 exactly as the forward kernel (edge_mfma.hip edge_major_half); weight tuples from the
                // packed table in memory, shared by the two halves
                BX_T(1);
                for (int p = p1;;) {
                    half4 B[2][4];
                    float muq[5];
#pragma unroll
                    for (int s = 0; s < 5; ++s) muq[s] = tw[(4 * s + q) * BX_PITCH + BX_MU];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float Rm[5];
#pragma unroll
                        for (int s = 0; s < 5; ++s) {
                            const float u = dist0[hf] - muq[s];
                            Rm[s] = fast_exp(-0.5f * (u * u)) * ((val0[hf] && lg0[hf] == (p == 1)) ? RBF_UP : 0.f);
                        }
                        rbf_tuples(Rm, B[hf]);
                    }
                    if (!(abl & 16)) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            n[0][t] = MFMAH(wt[t].t1, B[0][0], n[0][t]);       n[1][t] = MFMAH(wt[t].t1, B[1][0], n[1][t]);
                            n[0][t] = MFMAH(wt[t].t1, B[0][1], n[0][t]);       n[1][t] = MFMAH(wt[t].t1, B[1][1], n[1][t]);
                            n[0][t] = MFMAH(wt[t].t2, B[0][2], n[0][t]);       n[1][t] = MFMAH(wt[t].t2, B[1][2], n[1][t]);
                            n[0][t] = MFMAH(wt[t].t3, B[0][3], n[0][t]);       n[1][t] = MFMAH(wt[t].t3, B[1][3], n[1][t]);
                        }
                    }
                    if (p == 1 || !has_lig) break;      // mixed neighbourhood: the ligand-source class as well
                    p = 1;
                    const float* fa = frag + (size_t)etype(true, lig_i) * (8 * FRAG_BLK);
#pragma unroll
                    for (int t = 0; t < 8; ++t) wt[t] = load_wtuples(fa, t, lane);
                    SCHED_FENCE();
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {    // LayerNorm: the first Linear is centred, mean(pre) == 0
                    float v = 0.f;
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v = fmaf(n[hf][t][r], n[hf][t][r], v);
                    v = xrow_sum(v);
                    rstd[hf] = fast_rsqrt(__builtin_fmaf(v, rsc.c1, 1e-5f));       // the true 1 / sigma (v c1 = the unscaled variance)
                    const float sc = val0[hf] ? rstd[hf] * rsc.c2 : 0.f;            // padded slots: exact zeros downstream
#pragma unroll
                    for (int t = 0; t < 8; ++t) n[hf][t] = n[hf][t] * sc;
                }
                if (!(abl & 1024) && (CBGX_BX_SWEEP || (CBGX_BX_PARK && ph == 0))) {       // parked: the key path for phase 2 (n and rstd) [sweep: either path, n]
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            *reinterpret_cast<float4*>(nk + (hf * 8 + t) * 256) = make_float4(n[hf][t][0], n[hf][t][1], n[hf][t][2], n[hf][t][3]);
                    if (ph == 0) *reinterpret_cast<float2*>(nk + 16 * 256 - 2 * lane) = make_float2(rstd[0], rstd[1]);
                }
            }
            BX_T(2);
            // ---- scores (k) / G . v_raw (v):  out[hf] lane (c = head, q) reg r <-> edge 4q + r + 16 hf -----------------------------------
            float w[2][4];      // coefficient of hidden[e] per head in the backward (k: d score, v: alpha e_w)
            if ((abl & 64) && ph < 2) {
                if (ph == 0) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) alpha[hf][r] = n[hf][r][0];
                    continue;
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[hf][r] = n[hf][r][1];
            } else if (ph < 2) {
                floatx4 o[2][2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) { o[hf][0] = floatx4{0.f, 0.f, 0.f, 0.f}; o[hf][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float4 g4 = ld4(lng + 16 * t + 4 * q), b4 = ld4(lnb + 16 * t + 4 * q);     // LDS
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {    // padded slots produce ReLU(beta) . B: masked by `valid` / e_w = 0 below
                        const floatx4 nn = n[hf][t];
                        o[hf][0] = MFMA(fmaxf(fmaf(nn[0], g4.x, b4.x), 0.f), br[t].x, o[hf][0]);
                        o[hf][1] = MFMA(fmaxf(fmaf(nn[1], g4.y, b4.y), 0.f), br[t].y, o[hf][1]);
                        o[hf][0] = MFMA(fmaxf(fmaf(nn[2], g4.z, b4.z), 0.f), br[t].z, o[hf][0]);
                        o[hf][1] = MFMA(fmaxf(fmaf(nn[3], g4.w, b4.w), 0.f), br[t].w, o[hf][1]);
                    }
                }
                if (ph == 0) {      // softmax over the incoming edges
                    float mx = -INFINITY;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool valid = (vsh >> (r + 16 * hf)) & 1u;
                            alpha[hf][r] = valid ? o[hf][0][r] + o[hf][1][r] : -INFINITY;
                            mx = fmaxf(mx, alpha[hf][r]);
                        }
                    mx = xrow_max(mx);
                    float den = 0.f;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool valid = (vsh >> (r + 16 * hf)) & 1u;
                            alpha[hf][r] = valid ? expf(alpha[hf][r] - mx) : 0.f;
                            den += alpha[hf][r];
                        }
                    den = xrow_sum(den);
                    const float inv = den > 0.f ? 1.f / den : 0.f;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) alpha[hf][r] *= inv;
                    BX_T(3);
                    continue;
                }
                const float gbc = gb[(size_t)i * HEADS + c];
                float ewv[2][4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ewv[hf][r] = e_w[(size_t)i * KNN + 4 * q + r + 16 * hf];
                SCHED_FENCE();
                float swl = 0.f, dot = 0.f;
                float da[2][4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool valid = (vsh >> (r + 16 * hf)) & 1u;
                        const float ew = valid ? ewv[hf][r] : 0.f;
                        const float g = o[hf][0][r] + o[hf][1][r] + gbc;
                        da[hf][r] = ew * g;                       // d L / d alpha
                        dot = fmaf(alpha[hf][r], da[hf][r], dot);
                        w[hf][r] = alpha[hf][r] * ew;
                        swl += w[hf][r];
                        // d e_w[e] = sum over the heads (the 16 lanes of a row) of alpha (G . v_raw + gb)
                        const float de = row16_sum(alpha[hf][r] * g);
                        if (c == 0 && valid) de_w[(size_t)i * KNN + 4 * q + r + 16 * hf] += de;
                    }
                swl = xrow_sum(swl);
                dot = xrow_sum(dot);
                if (q == 0) sw[(size_t)i * HEADS + c] = swl;
                // softmax backward: from here on `alpha` holds d L / d score, the key path's coefficient in phase 2
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) alpha[hf][r] *= da[hf][r] - dot;
            } else {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[hf][r] = alpha[hf][r];
            }
            BX_T(3);
            // ================================= backward of the path =======================================================================
            if (abl & 8) continue;
            float* fold_dst = (kv ? S : T) + (size_t)i * HEADS * H;
            // w (E1: lane (c = head, q), edge 4q + r + 16 hf) -> wT[hf][s] in lane (c, q) = w[edge c + 16 hf][head 4 s + q]
            float wT[2][4];
            wave_sync();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) tw[(4 * q + r + 16 * hf) * BX_PITCH + c] = w[hf][r];
            wave_sync();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int s = 0; s < 4; ++s) wT[hf][s] = tw[(c + 16 * hf) * BX_PITCH + 4 * s + q];
            wave_sync();
            // n -> tile [edge][channel]
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    *reinterpret_cast<float4*>(tw + (c + 16 * hf) * BX_PITCH + 16 * t + 4 * q) =
                        make_float4(n[hf][t][0], n[hf][t][1], n[hf][t][2], n[hf][t][3]);
            wave_sync();
            BX_T(4);
            // ---- pass 1, C labeling: lane (c, q), step (u, j), [hf][r] <-> channel 32 u + 2 c + j, edge 4q + r + 16 hf ------------
            // folds T / S, LayerNorm affine gradients, the two per-edge sums of the LayerNorm backward.  Padded slots need no mask:
            // their w and wT are zero, so they add nothing to the fold and their d hidden is zero.
            // LayerNorm affine and the A-side rows of d hidden for the four steps: loaded once, used by pass 1 and pass 2
            float2 g2a[4], b2a[4], qaa[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = 32 * u + 2 * c;
                g2a[u] = ld2(lng + ch);
                b2a[u] = ld2(lnb + ch);
#pragma unroll
                for (int s = 0; s < 4; ++s) qaa[u][s] = ld2(Brow + (size_t)(4 * s + q) * H + ch);
            }
            SCHED_FENCE();
            // (LLVM keeps every memory access on its side of an atomic, even a relaxed one to another address space: the tile reads of
            // the next step are therefore issued by hand before a step's atomics, or each would wait out a full LDS round trip)
            float s1[2][4], s2[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[hf][r] = 0.f; s2[hf][r] = 0.f; }
            float2 nvn[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) nvn[hf][r] = *reinterpret_cast<const float2*>(tw + (4 * q + r + 16 * hf) * BX_PITCH + 2 * c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = 32 * u + 2 * c;
                const float2 g2 = g2a[u], b2 = b2a[u];
                const float2 (&qa)[4] = qaa[u];
                float2 nv[2][4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nv[hf][r] = nvn[hf][r];
                if (u < 3) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            nvn[hf][r] = *reinterpret_cast<const float2*>(tw + (4 * q + r + 16 * hf) * BX_PITCH + ch + 32);
                }
                floatx4 fold[2];
                float gsum[2], bsum[2];
                float vv[2][2][4];      // v = d hidden . gamma behind the ReLU mask: written over n below, read by passes 2 + 3
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float gmc = j ? g2.y : g2.x, btc = j ? b2.y : b2.x;
                    floatx4 f0 = {0.f, 0.f, 0.f, 0.f}, f1 = {0.f, 0.f, 0.f, 0.f};
                    float pre[2][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pre[0][r] = fmaf(j ? nv[0][r].y : nv[0][r].x, gmc, btc);
                        pre[1][r] = fmaf(j ? nv[1][r].y : nv[1][r].x, gmc, btc);
                        f0 = MFMA(fmaxf(pre[0][r], 0.f), w[0][r], f0);
                        f1 = MFMA(fmaxf(pre[1][r], 0.f), w[1][r], f1);
                    }
                    fold[j] = f0 + f1;
                    floatx4 de[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // d hidden[e][m] = sum_a w[e][a] Brow[a][m]
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        de[0] = MFMA(wT[0][s], j ? qa[s].y : qa[s].x, de[0]);
                        de[1] = MFMA(wT[1][s], j ? qa[s].y : qa[s].x, de[1]);
                    }
                    float gs = 0.f, bs = 0.f;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float nn = j ? nv[hf][r].y : nv[hf][r].x;
                            const float dy = pre[hf][r] > 0.f ? de[hf][r] : 0.f;
                            gs = fmaf(dy, nn, gs);
                            bs += dy;
                            const float v = dy * gmc;
                            vv[j][hf][r] = v;
                            s1[hf][r] += v;
                            s2[hf][r] = fmaf(v, nn, s2[hf][r]);
                        }
                    gsum[j] = xrow_sum(gs);
                    bsum[j] = xrow_sum(bs);
                }
#if CBGX_BX_SWEEP
                // (every cell of the step was read into nv before the first is overwritten)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<float2*>(tw + (4 * q + r + 16 * hf) * BX_PITCH + ch) = make_float2(vv[0][hf][r], vv[1][hf][r]);
#endif
                // fold D: lane (c = head, q) reg r <-> channel 32 u + 2 (4 q + r) + j
                float* fd = fold_dst + (size_t)c * H + 32 * u + 8 * q;
                if (!(abl & 4096) || fold[0][0] == 12345.f) {
                    *reinterpret_cast<float4*>(fd) = make_float4(fold[0][0], fold[1][0], fold[0][1], fold[1][1]);
                    *reinterpret_cast<float4*>(fd + 4) = make_float4(fold[0][2], fold[1][2], fold[0][3], fold[1][3]);
                }
                if (q == 0) {
                    atomicAdd(&L.lng[kv * H + ch], gsum[0]);
                    atomicAdd(&L.lng[kv * H + ch + 1], gsum[1]);
                    atomicAdd(&L.lnb[kv * H + ch], bsum[0]);
                    atomicAdd(&L.lnb[kv * H + ch + 1], bsum[1]);
                }
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[hf][r] = row16_sum(s1[hf][r]) * (1.f / H);
                    s2[hf][r] = row16_sum(s2[hf][r]) * (1.f / H);
                }
            BX_T(5);
#if CBGX_BX_SWEEP
            wave_sync();
            // ---- passes 2 + 3, E labeling, one sweep (round 6).  Pass 1 left v = d hidden . gamma (ReLU-masked) in the tile in place of n.
            // The LayerNorm backward  d pre = rstd (v - s1 - n s2)  is elementwise, so it runs in the labeling the forward left n in: n
            // comes back from the wave's scratch slot, d pre goes to the tile for pass 4, and on the way
            //     d dist[e] = sum_m d pre[e][m] V[e][m],   V[e][m] = sum_g rbf'_g(d_e) Wr[type_e][g][m],   rbf' = -(d - mu) rbf
            // -- V is the forward's rbf product with rbf' in place of rbf: the same split-f16 weight tuples, 64 f16 MFMAs (17 cycles)
            // where  d rbf = d pre . Wr^T  took 128 fp32 ones (32 cycles), and d hidden is not evaluated a second time (64 more).
            // E1 -> E0: lane (c, q) needs the sums of edge c + 16 hf, which the lanes of row c >> 2 hold in register c & 3
            auto e1_to_e0 = [&](const float (&a)[4]) {
                const int r = c & 3;
                const float sel = r == 0 ? a[0] : (r == 1 ? a[1] : (r == 2 ? a[2] : a[3]));
                return __shfl(sel, 16 * (c >> 2) + r, 64);
            };
            const float s1e[2] = {e1_to_e0(s1[0]), e1_to_e0(s1[1])}, s2e[2] = {e1_to_e0(s2[0]), e1_to_e0(s2[1])};
            float4 nq0[8];          // n of the first half: requested ahead of the rbf' product
#pragma unroll
            for (int t = 0; t < 8; ++t) nq0[t] = (abl & 1024) ? make_float4(0.1f, 0.2f, 0.3f, 0.4f) : ld4(nk + t * 256);
#else
            float rs1[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) rs1[hf][r] = __shfl(rstd[hf], 4 * q + r, 64);
            // ---- pass 2 (C): d hidden again (cheaper than 64 live registers of it) -> d pre, in place in the tile.  All cells of a step
            // are read before the first is written back (the compiler cannot tell the cells apart and would serialise read -> write) ----
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = 32 * u + 2 * c;
                const float2 g2 = g2a[u], b2 = b2a[u];
                const float2 (&qa)[4] = qaa[u];
                float2 nn[2][4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nn[hf][r] = *reinterpret_cast<const float2*>(tw + (4 * q + r + 16 * hf) * BX_PITCH + ch);
                float2 out[2][4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    floatx4 de0 = {0.f, 0.f, 0.f, 0.f}, de1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s) { de0 = MFMA(wT[hf][s], qa[s].x, de0); de1 = MFMA(wT[hf][s], qa[s].y, de1); }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dn0 = fmaf(nn[hf][r].x, g2.x, b2.x) > 0.f ? de0[r] * g2.x : 0.f;
                        const float dn1 = fmaf(nn[hf][r].y, g2.y, b2.y) > 0.f ? de1[r] * g2.y : 0.f;
                        // padded slots: d hidden = 0, n = 0, s1 = s2 = 0 -> exact zeros
                        out[hf][r] = make_float2(rs1[hf][r] * (dn0 - s1[hf][r] - nn[hf][r].x * s2[hf][r]),
                                                 rs1[hf][r] * (dn1 - s1[hf][r] - nn[hf][r].y * s2[hf][r]));
                    }
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(tw + (4 * q + r + 16 * hf) * BX_PITCH + ch) = out[hf][r];
            }
            wave_sync();
            // ---- pass 3 (E):  d dist[e] = sum_m d pre[e][m] V[e][m],   V[e][m] = sum_g rbf'_g(d_e) Wr[type_e][g][m],   rbf' = -(d - mu) rbf
            // -- V is the forward's rbf product with rbf' in place of rbf: the same split-f16 weight tuples, 64 f16 MFMAs (17 cycles)
            // where  d rbf = d pre . Wr^T  took 128 fp32 ones (32 cycles) and sixteen 16-byte weight loads per lane
#endif
            floatx4 V[2][8];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int t = 0; t < 8; ++t) V[hf][t] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (!(abl & 2)) {
                for (int p = p1;;) {        // the source classes present, as in the forward part
                    WTuples wt[8];
                    const float* fa = frag + (size_t)etype(p == 1, lig_i) * (8 * FRAG_BLK);
#pragma unroll
                    for (int t = 0; t < 8; ++t) wt[t] = load_wtuples(fa, t, lane);
                    half4 B[2][4];
                    float muq[5];
#pragma unroll
                    for (int s = 0; s < 5; ++s) muq[s] = tw[(4 * s + q) * BX_PITCH + BX_MU];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float Rm[5];
#pragma unroll
                        for (int s = 0; s < 5; ++s) {
                            const float u = dist0[hf] - muq[s];
                            Rm[s] = (-u * fast_exp(-0.5f * (u * u))) * ((val0[hf] && lg0[hf] == (p == 1)) ? RBF_UP : 0.f);
                        }
                        rbf_tuples(Rm, B[hf]);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        V[0][t] = MFMAH(wt[t].t1, B[0][0], V[0][t]);       V[1][t] = MFMAH(wt[t].t1, B[1][0], V[1][t]);
                        V[0][t] = MFMAH(wt[t].t1, B[0][1], V[0][t]);       V[1][t] = MFMAH(wt[t].t1, B[1][1], V[1][t]);
                        V[0][t] = MFMAH(wt[t].t2, B[0][2], V[0][t]);       V[1][t] = MFMAH(wt[t].t2, B[1][2], V[1][t]);
                        V[0][t] = MFMAH(wt[t].t3, B[0][3], V[0][t]);       V[1][t] = MFMAH(wt[t].t3, B[1][3], V[1][t]);
                    }
                    if (p == 1 || !has_lig) break;
                    p = 1;
                }
            }
            float ddE[2];
#if CBGX_BX_SWEEP
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float4 n4[8], v4[8];
                if (hf == 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) n4[t] = nq0[t];
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) n4[t] = (abl & 1024) ? make_float4(0.1f, 0.2f, 0.3f, 0.4f) : ld4(nk + (8 + t) * 256);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) v4[t] = *reinterpret_cast<const float4*>(tw + (c + 16 * hf) * BX_PITCH + 16 * t + 4 * q);
                SCHED_FENCE();
                const float rs = rstd[hf], a1 = s1e[hf], a2 = s2e[hf];
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    // padded slots: v = 0, n = 0, s1 = s2 = 0 -> exact zeros
                    const floatx4 dp = (f4(v4[t]) - a1 - f4(n4[t]) * a2) * rs;
                    *reinterpret_cast<float4*>(tw + (c + 16 * hf) * BX_PITCH + 16 * t + 4 * q) = make_float4(dp[0], dp[1], dp[2], dp[3]);
                    acc += dp * V[hf][t];
                }
                ddE[hf] = xrow_sum((acc[0] + acc[1]) + (acc[2] + acc[3])) * rsc.c2;      // V carries the forward's scale S; c2 = 1 / S
            }
#else
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    acc += f4(*reinterpret_cast<const float4*>(tw + (c + 16 * hf) * BX_PITCH + 16 * t + 4 * q)) * V[hf][t];
                ddE[hf] = xrow_sum((acc[0] + acc[1]) + (acc[2] + acc[3])) * rsc.c2;      // V carries the forward's scale S; c2 = 1 / S
            }
#endif
            if (q < 2) tw[(c + 16 * q) * BX_PITCH + BX_DDIST] += q ? ddE[1] : ddE[0];     // over the two paths, in the pad column
            wave_sync();
            BX_T(6);
            BX_T(7);
            // ---- pass 4 (C): every atomic of the path in one burst, no global load in between (any vmcnt wait after an atomic is a
            // full drain on gfx9): neighbour rows, own row, type columns, rbf columns of the first Linear -------------------------------------
            // (labeling of this pass: step t, lane (c, q), [hf][r] <-> channel 16 t + c -- a row's 16 lanes add to one 64-byte run)
            float rT0[2][4], rT1[2][4];
            {
                float u0[2][4], u1[2][4];
                rbf_e1(vsh, rT0, rT1, u0, u1);
            }
            gwptr dPb = sbase_w(dP);
            unsigned joff[2][4];    // byte offset of the neighbour's PS columns of this path (ER: of the edge's own row of dE)
            // ER: the node's 32 edge rows [32][k 128 | v 128] are one contiguous 32 KB block behind a wave-uniform base
            gwptr dEb = sbase_w(ER ? dE + (size_t)i * (KNN * 2 * H) : dP);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    joff[hf][r] = ER ? ((unsigned)(4 * q + r + 16 * hf) * (unsigned)(2 * H) + (unsigned)(kv * H + c)) * 4u
                                     : ((unsigned)reinterpret_cast<const int*>(tw)[(4 * q + r + 16 * hf) * BX_PITCH + BX_NBR] * (unsigned)PROW + (unsigned)((2 + kv) * H + c)) * 4u;
            const unsigned ioff = ((unsigned)i * (unsigned)PROW + (unsigned)(kv * H + c)) * 4u;
            const int ty1 = p1 ? ty_lig : ty_prot;
            float a0m[2][4], a1m[2][4];     // rbf of the edges of the first source class (padded slots: rT = 0)
            {
                const unsigned sel = !mixed ? 0xffffffffu : (p1 ? msh : ~msh);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = ((sel >> (r + 16 * hf)) & 1u) ? 1.f : 0.f;
                        a0m[hf][r] = rT0[hf][r] * m;
                        a1m[hf][r] = rT1[hf][r] * m;
                    }
            }
            // D of the rbf-column products: lane (c = channel, q) reg r <-> g = 4 q + r (tile 0), 16 + r (tile 1, q == 0)
            auto flush_rbf_columns = [&](int tyc, int col, floatx4 a0, floatx4 a1) {
                if (abl & 2048) {       // keep the products alive, skip the slab traffic
                    if (a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3] == 12345.f) L.dwr3[0][col] = 1.f;
                    return;
                }
                const int slot = col >> 4;      // 16 locks, one per 16 columns: pad column BX_MU of rows 20..27 of tiles 0, 1
                int* lk = reinterpret_cast<int*>(&L.tile[slot >> 3][(20 + (slot & 7)) * BX_PITCH + BX_MU]);
                if (tyc == 3) {
                    // plain read-modify-write under a per-column-block lock instead of 8 ds_add_f32 (64 LDS cycles each: the LDS
                    // pipe, shared by the 8 waves, was 55 % busy, a third of it these atomics).  16 locks (one per 16 columns):
                    // waves in different steps of the pass never meet.  Lock words: pad column BX_MU of rows 20..27 of tiles 0, 1.
                    lds_lock_w(lk, lane);
                    const int cx = col ^ (16 * (q & 1));
                    float v[4], w4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = L.dwr3[4 * q + r][cx];
                    if (q == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) w4[r] = L.dwr3[16 + r][col];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) L.dwr3[4 * q + r][cx] = v[r] + a0[r];
                    if (q == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) L.dwr3[16 + r][col] = w4[r] + a1[r];
                    }
                    lds_unlock_w(lk, lane);
                } else {
                    float* sl = slab + PB_WR + (size_t)tyc * G * 2 * H + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(sl + (4 * q + r) * 2 * H, a0[r]);
                    if (q == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(sl + (16 + r) * 2 * H, a1[r]);
                    }
                }
            };
            auto ysum = [&](floatx4 y) {      // CBGX_BX_Y4X4: the four blocks (q) of a channel group hold partial sums over their edges
#if CBGX_BX_Y4X4
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = xrow_sum(y[r]);
#endif
                return y;
            };
            floatx4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0, y0 = x0, y1 = x0;
            float dpn[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) dpn[hf][r] = tw[(4 * q + r + 16 * hf) * BX_PITCH + c];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float dp[2][4];
                float all = 0.f, ligs = 0.f;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[hf][r] = dpn[hf][r];
                if (t < 7) {    // the next step's cells, ahead of this step's atomics
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dpn[hf][r] = tw[(4 * q + r + 16 * hf) * BX_PITCH + 16 * (t + 1) + c];
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        all += dp[hf][r];
                        const float lm = ((msh >> (r + 16 * hf)) & 1u) ? 1.f : 0.f;
                        ligs = fmaf(lm, dp[hf][r], ligs);
                        if (!(abl & 1)) {
                            if (ER) *reinterpret_cast<__attribute__((address_space(1))) float*>(dEb + (joff[hf][r] + 64 * t)) = dp[hf][r];   // (padded slots: rows nobody reads)
                            else atomo(dPb, joff[hf][r] + 64 * t, dp[hf][r]);   // padded slots add 0 to the node's own row
                        }
                    }
                all = xrow_sum(all);
                ligs = xrow_sum(ligs);
                if (q == 0) {
                    // the PD columns of a node's own row have one writer, this wave: a plain store (dP is zero-filled; the atomic path is
                    // what bounds the kernel, see the header)
                    *reinterpret_cast<__attribute__((address_space(1))) float*>(dPb + (ioff + 64 * t)) = all;
                    atomicAdd(&L.wt[ty_lig][kv * H + 16 * t + c], ligs);
                    atomicAdd(&L.wt[ty_prot][kv * H + 16 * t + c], all - ligs);
                }
                // rbf columns of the first source class: A[g][edge] = rbf_g(d_e) over the edges of the class, B[edge][channel] = d pre.
                // The products of step t are added to the slab during step t + 1, when the matrix pipe has long delivered them.
                if (!(abl & 4)) {
                    if (t > 0) flush_rbf_columns(ty1, kv * H + 16 * (t - 1) + c, x0 + x1, ysum(y0 + y1));
                    x0 = x1 = y0 = y1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {       // four independent accumulator chains
#if CBGX_BX_Y4X4
                        // block (q, c >> 2) of the 16: A[m] = rbf_{16 + m}(edge 4q + r + 16 hf) from lane c & 3 == m, B[n] = d pre of channel
                        // 16 t + 4 (c >> 2) + n, D[m][n] in register m of lane n: the block's eight edges accumulate over (hf, r), the four
                        // blocks of a channel group (q) are summed by ysum() -- lane (c, *) register m <-> g = 16 + m, channel 16 t + c
                        x0 = MFMA(a0m[0][r], dp[0][r], x0); y0 = MFMA4(a1m[0][r], dp[0][r], y0);
                        x1 = MFMA(a0m[1][r], dp[1][r], x1); y1 = MFMA4(a1m[1][r], dp[1][r], y1);
#else
                        x0 = MFMA(a0m[0][r], dp[0][r], x0); y0 = MFMA(a1m[0][r], dp[0][r], y0);
                        x1 = MFMA(a0m[1][r], dp[1][r], x1); y1 = MFMA(a1m[1][r], dp[1][r], y1);
#endif
                    }
                }
            }
            if (!(abl & 4)) flush_rbf_columns(ty1, kv * H + 16 * 7 + c, x0 + x1, ysum(y0 + y1));
            if (mixed && !(abl & 4)) {      // rbf columns of the ligand-source class (its type is never 3): a second, rare sweep
                float lm[2][4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lm[hf][r] = ((msh >> (r + 16 * hf)) & 1u) ? 1.f : 0.f;
#pragma unroll 1
                for (int t = 0; t < 8; ++t) {
                    floatx4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float b = tw[(4 * q + r + 16 * hf) * BX_PITCH + 16 * t + c];
                            a0 = MFMA(rT0[hf][r] * lm[hf][r], b, a0);
#if CBGX_BX_Y4X4
                            a1 = MFMA4(rT1[hf][r] * lm[hf][r], b, a1);
#else
                            a1 = MFMA(rT1[hf][r] * lm[hf][r], b, a1);
#endif
                        }
                    flush_rbf_columns(ty_lig, kv * H + 16 * t + c, a0, ysum(a1));
                }
            }
            wave_sync();
            BX_T(8);
        }
        // =================================== coordinates ============================================================================
        // one lane per edge: lane (c, q < 2) <-> edge c + 16 q, its d L / d dist from the pad column; the node's own gradient is
        // reduced over the wave first (one atomic per coordinate instead of 32)
        wave_sync();
        {
            const int sel = q & 1;
            const int j = sel ? j0[1] : j0[0];
            const float xj = x[3 * j], yj = x[3 * j + 1], zj = x[3 * j + 2];
            const float dd = tw[(c + 16 * sel) * BX_PITCH + BX_DDIST];
            const float dist = sel ? dist0[1] : dist0[0];
            const bool on = q < 2 && c + 16 * sel < d;
            const float cf = (on && dist > 0.f) ? dd / dist : 0.f;
            const float g3[3] = {cf * (xi - xj), cf * (yi - yj), cf * (zi - zj)};
            if (on) {
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(&dx[3 * j + k], -g3[k]);
            }
            const float sx = wave_sum(g3[0]), sy = wave_sum(g3[1]), sz = wave_sum(g3[2]);
            if (lane == 0) {
                atomicAdd(&dx[3 * i], sx);
                atomicAdd(&dx[3 * i + 1], sy);
                atomicAdd(&dx[3 * i + 2], sz);
            }
        }
        wave_sync();
        BX_T(9);
        if (next_static) {
            it += stride;
            ++round;
        } else {        // the partial round: whoever gets here first takes the next node (its header loads are exposed, once)
            round = full_rounds;
            grab();
            if (it < it_end) load_header(it);
        }
    }
    // per-workgroup partial sums of the edge-indexed weight gradients
    __syncthreads();
    for (int u = tid; u < G * 2 * H; u += BX_WAVES * 64)       // rows 4q + r of q odd are stored with column bit 4 flipped
        slab[PB_WR + 3 * G * 2 * H + u] = (&L.dwr3[0][0])[u ^ (16 * ((u >> 10) & 1) * ((u >> 8) < 16 ? 1 : 0))];
    for (int u = tid; u < NT * 2 * H; u += BX_WAVES * 64) slab[PB_WT + u] = (&L.wt[0][0])[u];
    for (int u = tid; u < 2 * H; u += BX_WAVES * 64) { slab[PB_LNG + u] = L.lng[u]; slab[PB_LNB + u] = L.lnb[u]; }
}

hipError_t launch_edge_backward_x2h(const float* att, const float* x, const float* P, const float* Qt, const float* Gt,
                                    const float* gb, const int32_t* nbr, const int32_t* deg, const uint8_t* lig,
                                    const float* e_w, const int* rows, const int* n_rows, int n_nodes, float* T, float* S,
                                    float* sw, float* dP, float* dx, float* de_w, float* partial, float* nk_scratch,
                                    int* work_ctr, int grid, hipStream_t s, float* dE) {
    if (dE && rows) return hipErrorInvalidValue;      // edge rows are a full-launch mode (a listed launch would leave stale rows behind)
    profile_mark_begin(rows ? K_EDGE_X2H_BWD_LISTED : K_EDGE_X2H_BWD, s);
#ifdef CBGX_ABLATE
    static const int abl = getenv("CBGX_BWD_ABL") ? atoi(getenv("CBGX_BWD_ABL")) : 0;
    static const bool prof = (abl & 512) != 0;      // section timers: CBGX_BWD_ABL bit 512
    if (prof && !rows) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bx_prof), z, sizeof(z));
    }
    if (dE)
        hipLaunchKernelGGL(edge_backward_x2h_kernel<true>, dim3(grid), dim3(BX_WAVES * 64), 0, s, att, x, P, Qt, Gt, gb, nbr, deg, lig,
                           e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial, nk_scratch, work_ctr, dE, abl);
    else
        hipLaunchKernelGGL(edge_backward_x2h_kernel<false>, dim3(grid), dim3(BX_WAVES * 64), 0, s, att, x, P, Qt, Gt, gb, nbr, deg, lig,
                           e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial, nk_scratch, work_ctr, dE, abl);
    if (prof && !rows) {
        unsigned long long z[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_bx_prof), sizeof(z));
        static int shown = 0;
        if (shown++ % 7 == 3) {
            const char* nm[12] = {"prologue", "gather wait + adds", "rbf+LN", "contraction/softmax", "transposes", "pass1", "pass2", "pass3",
                                  "pass4", "coordinates", "tuple issue", "gather issue"};
            unsigned long long tot = 0;
            for (int k = 0; k < 12; ++k) tot += z[k];
            fprintf(stderr, "[bx prof] n=%d grid=%d total wave-cycles %llu:", n_nodes, grid, tot);
            for (int k = 0; k < 12; ++k) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)z[k] / (double)tot);
            fprintf(stderr, "\n");
        }
    }
#else
    if (dE)
        hipLaunchKernelGGL(edge_backward_x2h_kernel<true>, dim3(grid), dim3(BX_WAVES * 64), 0, s, att, x, P, Qt, Gt, gb, nbr, deg, lig,
                           e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial, nk_scratch, work_ctr, dE);
    else
        hipLaunchKernelGGL(edge_backward_x2h_kernel<false>, dim3(grid), dim3(BX_WAVES * 64), 0, s, att, x, P, Qt, Gt, gb, nbr, deg, lig,
                           e_w, rows, n_rows, n_nodes, T, S, sw, dP, dx, de_w, partial, nk_scratch, work_ctr, dE);
#endif
    profile_mark_end(s);
    return hipGetLastError();
}

}  // namespace cbgx

// libcbgx -- fused x2h / h2x edge kernels on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
// One wavefront owns one destination node i and its <= 32 incoming edges; a persistent 8-wave workgroup
// keeps the layer's rbf weight fragments (and, for x2h, the second v Linear) in LDS and loops over nodes.
//
//   pre[e][m] = PD[i][m] + PS[j_e][m] + dWt[e][m] + sum_g Wr[type_e][g][m] rbf_g(|x_i - x_j|)     (k and v)
//               the rbf sum was 160 of the node's 288 fp32 MFMAs; it runs on the f16 matrix pipe in split-f16 arithmetic
//               (weights and rbf values as hi + lo f16 pairs, three products per term, fp32 accumulate: 2^-22 relative, see
//               rbf_tuples) -- 128 v_mfma_f32_16x16x16_f16 of half the issue time each, with the same LDS footprint
//   hid       = ReLU(LayerNorm(pre))
//   score[e][a] = Qt[i][a] . hid_k[e]          (the key's 2nd Linear and 1/sqrt(8) are folded into Qt)
//   alpha     = softmax over the node's incoming edges, per head
//   x2h:  S[a] = sum_e alpha e_w hid_v[e]  ->  h_out = h + Wbv_a S[a] + bbv * sum_e alpha e_w
//   h2x:  wv[e][a] = Wbv[a] . hid_v[e] + bbv[a]  ->  dx = 1/16 sum_a sum_e alpha wv e_w (x_i - x_j)
//
// Pack-time algebra that removes VALU work here (cbgx_pack_weights, node_mfma.hip):
//   * the first Linear is centred over its 128 output channels (W - colmean(W), b - mean(b)), so pre[e][:]
//     has zero mean by construction and LayerNorm only needs sum(pre^2);
//   * PD already contains the bias and the type column of a protein source, Wt[type(src prot, dst i)];
//     only ligand-source edges add dWt = Wt[type(src lig, dst i)] - Wt[type(src prot, dst i)].
//
// All contractions are MFMAs and every accumulator is consumed in the layout it was produced in (no LDS
// transposes, no atomics -> deterministic).  tests/lanesim.py is the lane-by-lane model of this file and is
// checked against the reference on the CPU (tests/test_lanesim.py).  Lane l: c = l & 15, q = l >> 4.
//   edge-major tile t    (k; h2x v): lane column = edge e16, C row rho = 4q+r <-> channel m = 16t + rho
//   channel-major tile t (x2h v):    lane column c <-> channel m = 64(t>>2) + 4c + (t&3); C row 4q+r <-> edge 4q+r+16hf
// Both labelings make every gather instruction touch whole 64..256-byte runs of a row (16 cache lines per
// wave instruction instead of 64): the texture addresser, not the ALUs, was the first bottleneck.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "edge_common.h"
#include "kernels.h"
#include "layout.h"

// Waves of a persistent workgroup that take items.  A launch whose item list is short next to the grid (one graph: 445 nodes, an
// h2x block: the ~25 movable atoms per graph) is ONE node's dependent chain per wave -- 18 - 20 us of a 27 us launch
// (profiles/probe_r04z2.log) -- and two such chains on one SIMD slow each other down.  So a workgroup puts only as many of its
// WAVES waves to work as the list needs, but at least CBGX_EDGE_MIN_WAVES = 2 (each on its own SIMD), and the launchers size the grid
// for that many nodes per workgroup: the other waves help fill the LDS image and leave.  Decided per launch from the DEVICE-side list length,
// so cached / pruned / listed launches of a large batch get it too (the ~250 movable atoms of a 10-graph batch).  Same arithmetic per
// node whatever the schedule.  History: round 4 prepared this as a host-side switch (-DCBGX_EDGE_SMALL_W4, 4-wave workgroups for
// inputs of <= 1016 nodes) next to a dynamic remainder of the persistent loop (-DCBGX_EDGE_DYN); round 5's first GPU call measured
// both (profiles/small_dyn_r05a.log): 4 waves 1 167 -> 1 328 graph-steps/s at one graph (edge launches 26.8 -> 21.0 us), the dynamic
// remainder 9 116 -> 8 278 at ten graphs (x2h launch 52.8 -> 71.4 us) and nothing on the headline or the training line -- deleted.
// Second GPU call (profiles/small_r05b.log, fused node stage in every row): at least 8 / 4 / 2 waves per workgroup = 1 362 / 1 680 /
// 1 796 graph-steps/s at one graph (x2h launch 26.9 / 20.7 / 17.9 us), 9 773 / 10 647 / 10 924 at ten; the headline does not move.
#ifndef CBGX_EDGE_MIN_WAVES
#define CBGX_EDGE_MIN_WAVES 2
#endif
// Which wave of which workgroup takes item o of a round (S = workgroups x active waves items per round).  0: workgroup-major, o =
// slot * waves + wave (rounds 1 - 4) -- a partial last round then fills ALL waves of its first few workgroups while the others idle:
// a 10-graph batch is 2.17 nodes per wave, and the 48 third-round items of an XCD sat on 6 of its 32 workgroups, two to a SIMD.
// 1: wave-major, o = wave * slots + slot -- the same items go to wave 0 of every workgroup first, one to a SIMD, where a node's
// chain runs ~1.5 x faster (the waves-per-workgroup measurements above).  Full rounds are the same load either way.
// 1: the protein-only role's query channels are requested before the wait for the node's rows, and the x2h epilogue's residual row /
// bias before the v path (ahead of the next item's row prefetch) instead of right before the epilogue.  (A/B knob; 0 = rounds 2 - 4.)
#ifndef CBGX_EDGE_EARLY_HEADER
#define CBGX_EDGE_EARLY_HEADER 1     // the first item's header travels with the LDS fill (0: requested after its barrier, as until round 5)
#endif
#ifndef CBGX_EDGE_EARLY_LOADS
#define CBGX_EDGE_EARLY_LOADS 1
#endif
#ifndef CBGX_EDGE_WAVE_MAJOR
#define CBGX_EDGE_WAVE_MAJOR 1
#endif

namespace cbgx {


__constant__ float c_mu[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                              3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

// waves of a WAVES-wave workgroup that take items when `n_items` items are shared by `n_wg` workgroups (wave-uniform; the same
// formula sizes the roles of edge_x2h_dual_kernel and is restated in tests/test_bx_partition.py)
constexpr int EDGE_MIN_WAVES = CBGX_EDGE_MIN_WAVES;
__host__ __device__ __forceinline__ int edge_active_waves(int n_items, int n_wg, int waves) {
    const int per_wg = (n_items + n_wg - 1) / (n_wg > 0 ? n_wg : 1);
    const int lo = EDGE_MIN_WAVES < waves ? EDGE_MIN_WAVES : waves;
    return per_wg < lo ? lo : (per_wg > waves ? waves : per_wg);
}

// ---- edge-major path for one half (16 edges): pre-activation -> LayerNorm -> ReLU -> contraction with a
// per-lane row of B (Qt[i][a] for scores: registers `pre`; Wbv[a] for the h2x values: LDS `lds_brow`, already lane-offset).  Returns the 16x16 result tile:
// lane (c = a, q), reg r <-> edge 4q + r + 16hf.  `kv` selects the k (0) or v (1) quarter everywhere.
// `acc` enters as PD[i] + PS[j] (this node's and the neighbour's projection rows, channels 16t + 4q .. +3), summed by
// the caller as soon as the gathered rows arrive, so that the registers of the gather can be reused for the next one.
// FOLD (protein-only x2h kernel, first k half): `pre` is an OUTPUT here -- the folded query row Qt[i][a = c][16 t + 4 q ..] is
// computed from the node's eight query channels of head c (`q8`) and the fold table in LDS (layout.h PP_WFOLD) between the rbf MFMA
// block and the LayerNorm tail, i.e. after the q row has had a whole MFMA block to arrive: Qt never exists in memory.
template <bool PRE, bool FOLD = false>
__device__ __forceinline__ floatx4 edge_major_half(floatx4 (&acc)[8], bool lg, int kv,
                                                   const float* lds_frag, const float* lds_dwt, const float* lds_ln,
                                                   const float (&R)[5], bool has_prot, bool has_lig, int lig_i,
                                                   int lane, int q, const float* lds_brow,
                                                   float4 (&pre)[8], const RbfScale sc,
                                                   const float4 q8a = float4{0.f, 0.f, 0.f, 0.f},
                                                   const float4 q8b = float4{0.f, 0.f, 0.f, 0.f}) {
    if (has_lig) {  // wave-uniform: only nodes with a ligand neighbour pay for the type correction
        const float* dw = lds_dwt + lig_i * 2 * H + kv * H + 4 * q;
        const float m = lg ? sc.S : 0.f;     // the tile is carried scaled by S (edge_common.h RbfScale)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] += f4(ld4(dw + 16 * t)) * m;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p == 0 ? !has_prot : !has_lig) continue;  // wave-uniform
        float Rm[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) Rm[s] = (lg == (p == 1)) ? R[s] : 0.f;
        // split-f16 on the f16 matrix pipe, weight tuples as the A operand; two tiles at a time
        half4 B[4];
        rbf_tuples(Rm, B);
        const float* fa = lds_frag + (size_t)etype(p == 1, lig_i) * (8 * FRAG_BLK);
#pragma unroll
        for (int tg = 0; tg < 8; tg += 2) {
            const WTuples w0 = load_wtuples(fa, tg, lane), w1 = load_wtuples(fa, tg + 1, lane);
            acc[tg] = MFMAH(w0.t1, B[0], acc[tg]);         acc[tg + 1] = MFMAH(w1.t1, B[0], acc[tg + 1]);
            acc[tg] = MFMAH(w0.t1, B[1], acc[tg]);         acc[tg + 1] = MFMAH(w1.t1, B[1], acc[tg + 1]);
            acc[tg] = MFMAH(w0.t2, B[2], acc[tg]);         acc[tg + 1] = MFMAH(w1.t2, B[2], acc[tg + 1]);
            acc[tg] = MFMAH(w0.t3, B[3], acc[tg]);         acc[tg + 1] = MFMAH(w1.t3, B[3], acc[tg + 1]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the tail's operand loads (g, b, B row) below the MFMA block
    if (FOLD) {
        // Qt[c][16 t + 4 q + r] = sum_d q[8 c + d] Wbk[8 c + d][16 t + 4 q + r] / sqrt(8): 64 conflict-free ds_read_b128 and 128
        // packed FMAs per node instead of 8 KB of Qt written by node_qfold_kernel and read back here
        const float qd[8] = {q8a.x, q8a.y, q8a.z, q8a.w, q8b.x, q8b.y, q8b.z, q8b.w};
        const float* wf = lds_brow;     // fold table, already lane-offset: (t, d) at 2048 t + 256 d floats
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float2v lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const floatx4 w4 = f4(ld4(wf + 2048 * t + 256 * d));
                const float2v qq = splat2(qd[d]);
                lo = lo2(w4) * qq + lo;
                hi = hi2(w4) * qq + hi;
            }
            pre[t] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    // LayerNorm: the first Linear is centred, so mean(pre) == 0 and var = mean(pre^2)
    float2v v2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        v2 = lo2(acc[t]) * lo2(acc[t]) + v2;
        v2 = hi2(acc[t]) * hi2(acc[t]) + v2;
    }
    const float v = xrow_sum(v2.x + v2.y);
    const float rstd = fast_rsqrt(__builtin_fmaf(v, sc.c1, 1e-5f)) * sc.c2;
    const float2v r2 = splat2(rstd);
    const float* lg_ = lds_ln + (2 * kv) * H + 4 * q;
    const float* lb_ = lds_ln + (2 * kv + 1) * H + 4 * q;
    // two interleaved accumulators: a dependent 16x16x4 MFMA needs 40 cycles, an independent one 32
    floatx4 out0 = {0.f, 0.f, 0.f, 0.f}, out1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const floatx4 g = f4(ld4(lg_ + 16 * t)), b = f4(ld4(lb_ + 16 * t));
        const float2v ya = (lo2(acc[t]) * r2) * lo2(g) + lo2(b);
        const float2v yb = (hi2(acc[t]) * r2) * hi2(g) + hi2(b);
        const float4 bb = PRE ? pre[t] : ld4(lds_brow + 256 * t);    // h2x: Wbv[head c][16 t + 4 q ..] (layout.h, image "wbv")
        out0 = MFMA(fmaxf(ya.x, 0.f), bb.x, out0);
        out1 = MFMA(fmaxf(ya.y, 0.f), bb.y, out1);
        out0 = MFMA(fmaxf(yb.x, 0.f), bb.z, out0);
        out1 = MFMA(fmaxf(yb.y, 0.f), bb.w, out1);
    }
    return out0 + out1;
}

// Geometry of one work item in the two lane mappings the kernel uses:
//   E0: lane (c, q) <-> edges c and c + 16 (distances, rbf, edge-major tiles);  E1: lane (c, q) <-> edges 4q + r (+16)
// Invalid slots (e >= deg) point at the node itself, so every gather below is unconditional: a predicated load compiles
// to branch + load + wait per element, which is what used to serialise the memory latencies of a node.
struct ItemGeom {
    int node, d, lig_i;
    float xi, yi, zi;
    int j0[2];          // E0 neighbour ids
};
// E1 neighbour id of slot r of a half: the raw row entry (-1 padded) or the node itself
__device__ __forceinline__ int e1_id(int raw, int e, int d, int self) { return e < d ? raw : self; }

// LISTED: the launch iterates over a device-side node list (h2x always; x2h in the pruned last layers) -- a separate
// instantiation so that profilers report full-graph and listed launches under different kernel names.
//
// Memory pipeline of one iteration (every gather has at least one MFMA block between its issue and its first use):
//   previous iteration   ids / flags / coordinates of this node and its neighbours; PD[i], PS_k[j] rows (issued before
//                        that node's epilogue)
//   top                  both halves' PD + PS_k summed (frees the 24 gather registers); then Qt[i] rows, PS_v gathers of
//                        half 0 (x2h), ids of the next node (a)
//   after k half 0       PS_v gathers of half 1 (x2h), e_w; coordinates / flags of the next node's neighbours (b)
//   after k half 1       distances / flags of the next node's edges
//   before the epilogue  PD / PS_k rows of the next node (c)
// PP (x2h, listed): the protein-only specialisation -- every listed destination and all of its neighbours are protein atoms (the
// launcher's list guarantees it), so only the type-3 rbf tables are needed and the LDS image (layout.h A_IMG_PP) carries the second
// k Linear instead of the other three types: the query fold happens in registers, `Qt` is then the UNFOLDED query q[N,128].
// The body is a device function over (`wg`, `n_wg`) = this workgroup's index and the number of workgroups that share the item list:
// the plain kernels pass blockIdx.x / gridDim.x, the two-role x2h kernel (edge_x2h_dual_kernel) gives each role its own range.
// `lds` [IMG floats], `lds_mu` [G floats]: the calling kernel's shared arrays.
template <bool X2H, int WAVES, bool LISTED, bool PP>
__device__ __forceinline__ void edge_body(
    float* __restrict__ lds, float* __restrict__ lds_mu, const int wg, const int n_wg,
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ h,
    const float* __restrict__ P, const float* __restrict__ Qt, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ deg, const uint8_t* __restrict__ lig, const uint8_t* __restrict__ gen,
    const float* __restrict__ e_w, int n_nodes, float* __restrict__ out, float* __restrict__ dx_out,
    const int* __restrict__ act_arg, const int* __restrict__ act_count) {
    const int* __restrict__ act = LISTED ? act_arg : nullptr;
    static_assert(!PP || (X2H && LISTED), "the protein-only kernel is an x2h work-list kernel");
    static_assert(PP_IMG_SIZE == IMG_SIZE_X2H, "both x2h images fill the same LDS array");
    constexpr int IMG = X2H ? (int)IMG_SIZE_X2H : (int)IMG_SIZE_H2X;
    // work list mode of h2x: x_out = x for every node that cannot move -- by all workgroups, also those without an item.  The
    // flag and the position are requested together (one round trip, not two), and a workgroup with items does this behind its
    // LDS fill's loads instead of in front of them.
    auto copy_fixed_nodes = [&]() {
        if (!X2H && act) {
            for (int n = wg * (WAVES * 64) + threadIdx.x; n < n_nodes; n += n_wg * WAVES * 64) {
                unsigned gn = gen[n];
                float px = x[3 * n], py = x[3 * n + 1], pz = x[3 * n + 2];
                asm volatile("" : "+v"(gn), "+v"(px), "+v"(py), "+v"(pz));      // (otherwise the position load sinks into the `if`)
                if (!gn) {
                    out[3 * n] = px; out[3 * n + 1] = py; out[3 * n + 2] = pz;
                    if (dx_out) { dx_out[3 * n] = 0.f; dx_out[3 * n + 1] = 0.f; dx_out[3 * n + 2] = 0.f; }
                }
            }
        }
    };
    const int n_items = act ? *act_count : n_nodes;
    const int wv = edge_active_waves(n_items, n_wg, WAVES);     // waves of this workgroup that take items (see CBGX_EDGE_MIN_WAVES)
    {   // a workgroup with no item skips the LDS fill altogether (its wave 0 holds its first item in either mapping)
        int first;
        if ((n_wg & 7) == 0) {
            const int per_xcd = (((n_items + 7) >> 3) + wv - 1) / wv * wv;
            first = (wg & 7) * per_xcd + (wg >> 3) * (CBGX_EDGE_WAVE_MAJOR ? 1 : wv);
            if (first >= min(n_items, ((int)(wg & 7) + 1) * per_xcd)) { copy_fixed_nodes(); return; }
        } else {
            first = wg * (CBGX_EDGE_WAVE_MAJOR ? 1 : wv);
            if (first >= n_items) { copy_fixed_nodes(); return; }
        }
    }
    // the wave index -- and with it every node index of the persistent loop -- lives in scalar registers: node-level values
    // (degree, flag, position, row bases) are then scalar loads and SGPR operands instead of 64 identical lanes
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;

    // XCD-aware persistent schedule: workgroup b runs on XCD b % 8 (observed dispatch order), so give every
    // XCD one contiguous eighth of the item range: a graph's PS / Qt rows are then pulled into one L2 only.  (A role of the
    // two-role kernel starts at a multiple of 8, so wg % 8 is still the XCD.)
    int i_begin, i_end, i_step;
    if ((n_wg & 7) == 0) {
        const int per_xcd = (((n_items + 7) >> 3) + wv - 1) / wv * wv;
        const int xcd = wg & 7, slot = wg >> 3, slots = n_wg >> 3;
        i_begin = xcd * per_xcd + (CBGX_EDGE_WAVE_MAJOR ? wave * slots + slot : slot * wv + wave);
        i_end = min(n_items, (xcd + 1) * per_xcd);
        i_step = slots * wv;
    } else {
        i_begin = CBGX_EDGE_WAVE_MAJOR ? wave * n_wg + wg : wg * wv + wave;
        i_end = n_items;
        i_step = n_wg * wv;
    }
    const bool has_item = wave < wv && i_begin < i_end;      // wave-uniform
    // The first item's header -- list entry -> node -> degree, class, position, neighbour row: two dependent round trips -- is
    // requested BEFORE the LDS fill and travels with it (CBGX_EDGE_EARLY_HEADER): behind the fill's barrier it opened every launch,
    // ~1.4 us of the ~8 us a small input's launch works (27 launches per denoising step).
    ItemGeom g;
    int r0 = 0, r1 = 0;
    int4 nb0 = {0, 0, 0, 0}, nb1 = {0, 0, 0, 0};      // the node's neighbour row in the E1 mapping (raw, -1 padded)
    int lig_first = 0;      // (raw: made wave-uniform after the barrier, so that nothing here waits for it)
    auto first_header = [&]() {
        const int i = __builtin_amdgcn_readfirstlane(act ? act[i_begin] : i_begin);
        g.node = i;
        g.d = deg[i]; lig_first = PP ? 0 : lig[i];
        g.xi = x[3 * i]; g.yi = x[3 * i + 1]; g.zi = x[3 * i + 2];
        const gptr nrow = sbase(nbr + (size_t)i * KNN);
        const unsigned oc = vop(4 * c), oq = vop(16 * q);
        r0 = ldoi(nrow, oc); r1 = ldoi(nrow, oc + 64);
        nb0 = ldoi4(nrow, oq); nb1 = ldoi4(nrow, oq + 64);
    };
    {
        // LDS fill: every load of the thread is requested before the first is written (19 float4 per thread for the x2h image).
        // Left as a plain loop the compiler emits load -> wait -> ds_write per iteration: 19 dependent L2 round trips, ~25 us at
        // the head of EVERY launch -- the whole duration of a small one (a 1-graph step is 18 such launches).
        const floatx4* src = reinterpret_cast<const floatx4*>(att + (PP ? A_IMG_PP : A_IMG));
        floatx4* dst = reinterpret_cast<floatx4*>(lds);
        constexpr int NV = (IMG / 4 + WAVES * 64 - 1) / (WAVES * 64);
        if constexpr (WAVES >= 8) {
            floatx4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int t = threadIdx.x + u * (WAVES * 64);
                v[u] = src[t < IMG / 4 ? t : IMG / 4 - 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (CBGX_EDGE_EARLY_HEADER && has_item) first_header();      // behind the fill's loads, ahead of its writes
            copy_fixed_nodes();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int t = threadIdx.x + u * (WAVES * 64);
                if (t < IMG / 4) dst[t] = v[u];
            }
        } else {
            if (CBGX_EDGE_EARLY_HEADER && has_item) first_header();
            copy_fixed_nodes();        // fewer threads: the same in passes of at most 20 float4 per thread (38 at once would not fit the registers)
            constexpr int NVC = 20;
#pragma unroll 1
            for (int u0 = 0; u0 < NV; u0 += NVC) {
                floatx4 v[NVC];
#pragma unroll
                for (int u = 0; u < NVC; ++u) {
                    const int t = threadIdx.x + (u0 + u) * (WAVES * 64);
                    v[u] = src[t < IMG / 4 ? t : IMG / 4 - 1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NVC; ++u) {
                    const int t = threadIdx.x + (u0 + u) * (WAVES * 64);
                    if (t < IMG / 4) dst[t] = v[u];
                }
            }
        }
        if (threadIdx.x < G) lds_mu[threadIdx.x] = c_mu[threadIdx.x];
    }
    __syncthreads();
    // PP: the tables of edge type 3 sit at the head of the image; the table bases are biased so that etype() = 3 indexes them
    const float* lds_fk = PP ? lds + PP_FRAG_K - 3 * 8 * (int)FRAG_BLK : lds + IMG_FRAG_K;
    const float* lds_fv = PP ? lds + PP_FRAG_V - 3 * 8 * (int)FRAG_BLK : lds + IMG_FRAG_V;
    const float* lds_dwt = lds + IMG_WT;     // (never read by PP: no ligand source)
    const float* lds_ln = lds + (PP ? PP_LN : IMG_LN);

    if (!has_item) return;     // (after the barrier: this wave has done its share of the LDS fill)
    if (!CBGX_EDGE_EARLY_HEADER) first_header();
    g.lig_i = __builtin_amdgcn_readfirstlane(lig_first);
    // power-of-two scales of the split-f16 rbf tables (wave-uniform: scalar registers)
    const RbfScale sck = load_rbf_scale(att, 0), scv = load_rbf_scale(att, 1);
    // h2x: bias of this lane's head, once per launch -- loaded inside the loop it sat behind the next node's 24-row prefetch in the
    // in-order vmcnt queue, so the wait for this one word drained the whole prefetch before the node's store
    const float bbv = X2H ? 0.f : att[A_BBV + c];

    // ---- first item: geometry and the k-path rows, everything unconditional -------------------------------------
    bool lg0[2];
    float dist0[2];
    float4 pd[8], ps0[8], ps1[8];
    {
        const int i = g.node;
        const unsigned oq = vop(16 * q);
        g.j0[0] = c < g.d ? r0 : i;
        g.j0[1] = c + 16 < g.d ? r1 : i;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int j = g.j0[hf];
            const int lj = PP ? 0 : ldob(sbase(lig), (unsigned)j);
            lg0[hf] = (c + 16 * hf < g.d) && lj;
            dist0[hf] = edge_len(g.xi, g.yi, g.zi, ldo1(sbase(x), 12u * j), ldo1(sbase(x), 12u * j + 4), ldo1(sbase(x), 12u * j + 8));
        }
        const gptr pdp = sbase(P + (size_t)i * PROW);
        const unsigned o0 = (unsigned)g.j0[0] * (PROW * 4) + (2 * H + 4 * q) * 4, o1 = (unsigned)g.j0[1] * (PROW * 4) + (2 * H + 4 * q) * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) { pd[t] = ldo4(pdp, oq + 64 * t); ps0[t] = ldo4(sbase(P), o0 + 64 * t); }
#pragma unroll
        for (int t = 0; t < 8; ++t) ps1[t] = ldo4(sbase(P), o1 + 64 * t);
    }

    for (int k = i_begin; k < i_end; k += i_step) {
        const int i = __builtin_amdgcn_readfirstlane(g.node), d = g.d, lig_i = g.lig_i;
        const bool more = k + i_step < i_end;   // wave-uniform
        const int k_next = k + i_step;
        // PP: the node's eight query channels of head c (32 bytes per lane), folded in k half 0.  Requested BEFORE the wait for the
        // rows below: in the latency regime (one node per wave: the rows were requested a moment ago) the two round trips then
        // overlap instead of following each other -- the fold sits one short MFMA block behind the top of the iteration
        float4 q8a = {0.f, 0.f, 0.f, 0.f}, q8b = {0.f, 0.f, 0.f, 0.f};
        if (PP && CBGX_EDGE_EARLY_LOADS) {
            const gptr qp = sbase(Qt + (size_t)i * H);
            const unsigned oq8 = vop(32 * c);
            q8a = ldo4(qp, oq8); q8b = ldo4(qp, oq8 + 16);
        }
        // both halves' PD[i] + PS_k[j] as soon as the rows (requested one epilogue ago) are here: the 24 gather registers
        // are then free for this iteration's other gathers
        floatx4 acc0[8], acc1[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {   // (PD + PS) S: the tile is carried scaled by S (edge_common.h RbfScale)
            const floatx4 pds = f4(pd[t]) * sck.S;
            acc0[t] = f4(ps0[t]) * sck.S + pds; acc1[t] = f4(ps1[t]) * sck.S + pds;
        }
        __builtin_amdgcn_sched_barrier(0);
        // (a) next item: id, degree, flag, position, neighbour ids in both mappings.  The last iteration re-requests its own
        // node (every prefetch below is unconditional: no divergent joins for the register allocator, no predicated loads).
        // Issued BEFORE this node's gathers: vmcnt retires in order, so waiting for these few words later (b) leaves the
        // gathers in flight, while the other order would drain them.  The values stay in vector registers until (b).
        ItemGeom ng;
        const int inext = __builtin_amdgcn_readfirstlane(more ? (act ? act[k_next] : k_next) : i);
        ng.node = inext;
        const int nd_raw = deg[inext];
        const int nlig_raw = PP ? 0 : ldob(sbase(lig), vop((unsigned)inext));
        const gptr xrow = sbase(x + 3 * (size_t)inext);
        const unsigned ozero = vop(0u);
        const float nx_raw = ldo1(xrow, ozero), ny_raw = ldo1(xrow, ozero + 4), nz_raw = ldo1(xrow, ozero + 8);
        const gptr nrow = sbase(nbr + (size_t)inext * KNN);
        const unsigned oc = vop(4 * c);
        const int nr0 = ldoi(nrow, oc), nr1 = ldoi(nrow, oc + 64);
        __builtin_amdgcn_sched_barrier(0);
        // this node's folded query row (B operand of the score MFMAs): consumed after the first pre-activation block
        float4 qrow[8];
        if (PP && !CBGX_EDGE_EARLY_LOADS) {
            const gptr qp = sbase(Qt + (size_t)i * H);
            const unsigned oq8 = vop(32 * c);
            q8a = ldo4(qp, oq8); q8b = ldo4(qp, oq8 + 16);
        }
        if (!PP) {
            const gptr qp = sbase(Qt + (size_t)i * HEADS * H);
            const unsigned oqr = vop((c * H + 4 * q) * 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) qrow[t] = ldo4(qp, oqr + 64 * t);
        }
        // x2h: PS_v rows (channel-major gather, E1 mapping) of half 0 and this node's own PD_v, needed two MFMA blocks from here
        float4 sva[2][4], svb[2][4];
        float4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
        if (X2H) {
            const int nbv[4] = {nb0.x, nb0.y, nb0.z, nb0.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned o = (unsigned)e1_id(nbv[r], 4 * q + r, d, i) * (PROW * 4) + (3 * H + 4 * c) * 4;
                sva[0][r] = ldo4(sbase(P), o);
                svb[0][r] = ldo4(sbase(P), o + 256);
            }
            const gptr pvp = sbase(P + (size_t)i * PROW);
            const unsigned ocv = vop((H + 4 * c) * 4);
            pa = ldo4(pvp, ocv);
            pb = ldo4(pvp, ocv + 256);
        }
        __builtin_amdgcn_sched_barrier(0);
        float R[2][5];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            // exp() of every slot, then a 0 / 1 factor: `valid ? exp(..) : 0` compiles to a divergent branch per value
            const float vm = c + 16 * hf < d ? RBF_UP : 0.f;   // 0 / 2^RBF_EXP: the rbf values enter the f16 pipe scaled
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float u = dist0[hf] - lds_mu[4 * s + q];   // re-read per node: five registers less across the loop
                R[hf][s] = fast_exp(-0.5f * (u * u)) * vm;
            }
        }
        const unsigned long long b0 = __ballot(lg0[0]), b1 = __ballot(lg0[1]);
        const unsigned mask_lig = (unsigned)(b0 & 0xffffull) | ((unsigned)(b1 & 0xffffull) << 16);
        const unsigned mask_valid = d >= 32 ? 0xffffffffu : ((1u << d) - 1u);
        const bool has_lig = PP ? false : (mask_lig & mask_valid) != 0;
        const bool has_prot = PP ? true : ((~mask_lig) & mask_valid) != 0 || d == 0;

        // ---- k path: hidden (edge-major) -> scores -> softmax ------------------------------------------
        floatx4 sc[2];
        sc[0] = edge_major_half<true, PP>(acc0, lg0[0], 0, lds_fk, lds_dwt, lds_ln, R[0], has_prot, has_lig, lig_i, lane, q,
                                          PP ? lds + PP_WFOLD + 4 * lane : nullptr, qrow, sck, q8a, q8b);
        __builtin_amdgcn_sched_barrier(0);
        // (b) next item: resolve its neighbour ids (they arrived during the first half), request their flags and
        // coordinates; then the second half of the PS_v gather and the gate values
        int nlj[2];
        float nxj[2][3];
        {
            ng.d = __builtin_amdgcn_readfirstlane(nd_raw);
            ng.lig_i = __builtin_amdgcn_readfirstlane(nlig_raw);
            ng.xi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(nx_raw)));
            ng.yi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ny_raw)));
            ng.zi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(nz_raw)));
            ng.j0[0] = c < ng.d ? nr0 : inext;
            ng.j0[1] = c + 16 < ng.d ? nr1 : inext;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int j = ng.j0[hf];
                nlj[hf] = PP ? 0 : ldob(sbase(lig), (unsigned)j);
                nxj[hf][0] = ldo1(sbase(x), 12u * j); nxj[hf][1] = ldo1(sbase(x), 12u * j + 4); nxj[hf][2] = ldo1(sbase(x), 12u * j + 8);
            }
        }
        float4 ew0 = {0.f, 0.f, 0.f, 0.f}, ew1 = {0.f, 0.f, 0.f, 0.f};
        if (X2H) {
            const int nbv[4] = {nb1.x, nb1.y, nb1.z, nb1.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned o = (unsigned)e1_id(nbv[r], 4 * q + r + 16, d, i) * (PROW * 4) + (3 * H + 4 * c) * 4;
                sva[1][r] = ldo4(sbase(P), o);
                svb[1][r] = ldo4(sbase(P), o + 256);
            }
        }
        const gptr ewp = sbase(e_w + (size_t)i * KNN);
        const unsigned oqe = vop(16 * q);
        ew0 = ldo4(ewp, oqe);
        ew1 = ldo4(ewp, oqe + 64);
        // h2x: this node's PD_v row and the PS_v rows of half 0 (edge-major), needed after the second k half: the registers of the
        // first half's accumulators carry them (until round 3 all 24 v rows were requested and waited for in one place, a full
        // exposed round trip per listed node)
        float4 vd[8], vs0[8], vs1[8];
        if (!X2H) {
            const gptr pdp = sbase(P + (size_t)i * PROW);
            const unsigned o0 = (unsigned)g.j0[0] * (PROW * 4) + (3 * H + 4 * q) * 4;
            const unsigned ovd = vop((H + 4 * q) * 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) { vd[t] = ldo4(pdp, ovd + 64 * t); vs0[t] = ldo4(sbase(P), o0 + 64 * t); }
        }
        __builtin_amdgcn_sched_barrier(0);
        sc[1] = edge_major_half<true>(acc1, lg0[1], 0, lds_fk, lds_dwt, lds_ln, R[1], has_prot, has_lig, lig_i, lane, q,
                                      nullptr, qrow, sck);
        __builtin_amdgcn_sched_barrier(0);
        if (!X2H) {   // PS_v rows of half 1: in flight during the softmax and the first v half
            const unsigned o1 = (unsigned)g.j0[1] * (PROW * 4) + (3 * H + 4 * q) * 4;
#pragma unroll
            for (int t = 0; t < 8; ++t) vs1[t] = ldo4(sbase(P), o1 + 64 * t);
        }
        // h2x: the neighbours' coordinates in the E1 mapping (equivariant update of the epilogue), requested here (the query
        // row and the k accumulators are dead) with the ids that came with this node's rows: the epilogue then has no load of its own (it used to run 24 dependent-looking dword loads, each
        // behind a vmcnt wait that also retired the next node's row prefetch: ~15 L2 round trips per listed node)
        float xj[2][4][3] = {};
        if (!X2H) {
            const int nbv[2][4] = {{nb0.x, nb0.y, nb0.z, nb0.w}, {nb1.x, nb1.y, nb1.z, nb1.w}};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // invalid slots point at the node itself: rel = 0 and alpha = 0, so they add exactly nothing
                    const unsigned o = 12u * (unsigned)e1_id(nbv[hf][r], 4 * q + r + 16 * hf, d, i);
                    xj[hf][r][0] = ldo1(sbase(x), o); xj[hf][r][1] = ldo1(sbase(x), o + 4); xj[hf][r][2] = ldo1(sbase(x), o + 8);
                }
        }
        // next item: distances and ligand flags of its edges from the (b) loads
        bool nlg[2];
        float ndist[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            nlg[hf] = (c + 16 * hf < ng.d) && nlj[hf];
            ndist[hf] = edge_len(ng.xi, ng.yi, ng.zi, nxj[hf][0], nxj[hf][1], nxj[hf][2]);
        }
        // E1 mapping: lane (c = head a, q), reg (hf, r) <-> edge e = 4q + r + 16hf
        float al[2][4];
        float mx = -INFINITY;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = 4 * q + r + 16 * hf < d;
                al[hf][r] = valid ? sc[hf][r] : -INFINITY;
                mx = fmaxf(mx, al[hf][r]);
            }
        mx = xrow_max(mx);
        float den = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = 4 * q + r + 16 * hf < d;
                al[hf][r] = valid ? fast_exp(al[hf][r] - mx) : 0.f;
                den += al[hf][r];
            }
        den = xrow_sum(den);
        const float inv_den = den > 0.f ? fast_rcp(den) : 0.f;
        const float ew[2][4] = {{ew0.x, ew0.y, ew0.z, ew0.w}, {ew1.x, ew1.y, ew1.z, ew1.w}};
        __builtin_amdgcn_sched_barrier(0);

        int4 nnb0, nnb1;   // next node's neighbour row in the E1 mapping, requested before the epilogue
        if (X2H) {
            float w[2][4];
            float sw = 0.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // softmax first (alpha = ex / den, like scatter_softmax), then the gate
                    w[hf][r] = (al[hf][r] * inv_den) * ew[hf][r];
                    sw += w[hf][r];
                }
            sw = xrow_sum(sw);   // sum_e alpha e_w for head a = c
            // residual row and bias of this lane's two outputs, used after the Wbv products of the epilogue: requested here, ahead
            // of the next item's 24-row prefetch (vmcnt retires in order: behind it, the wait for these two words drained the
            // prefetch -- and in the latency regime was a round trip of its own)
            const int n0 = 8 * c + 2 * q;
            const unsigned on0 = vop(n0 * 4);
            float2 hres, bias2;
            if (CBGX_EDGE_EARLY_LOADS) { hres = ldo2(sbase(h + (size_t)i * H), on0); bias2 = ldo2(sbase(att + A_BBV), on0); }
            // ---- v path, channel-major, one half at a time: lane (c, q) reg r <-> edge 4q + r + 16hf, m = 8c + t
            // aggregated straight into s2[t] = hid_v^T . w : lane (c = head a, q) reg r' <-> channel 64(t>>2) + 16q + 4r' + (t&3)
            floatx4 s2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) s2[t] = floatx4{0.f, 0.f, 0.f, 0.f};
            const float pdv[8] = {pa.x * scv.S, pa.y * scv.S, pa.z * scv.S, pa.w * scv.S,
                                  pb.x * scv.S, pb.y * scv.S, pb.z * scv.S, pb.w * scv.S};
            const float4 ga = ld4(lds_ln + 2 * H + 4 * c), gb = ld4(lds_ln + 2 * H + 64 + 4 * c);
            const float4 ba = ld4(lds_ln + 3 * H + 4 * c), bb = ld4(lds_ln + 3 * H + 64 + 4 * c);
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
            const float bv[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                floatx4 hv[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // a gathered float4 is four channels of one edge, a tile register quad is four edges of one channel:
                    // plain v_add_f32 writes each sum where the MFMA wants it.  The empty asm keeps the vectoriser from
                    // pairing the adds (it would build every pair with two v_mov first: 3 instructions per 2 sums).
                    const float4 sa = sva[hf][r], sb = svb[hf][r];
                    const float in[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        float sum = __builtin_fmaf(in[t], scv.S, pdv[t]);    // (PD_v + PS_v) S
                        asm("" : "+v"(sum));
                        hv[t][r] = sum;
                    }
                }
                if (has_lig) {
                    const float* dw = lds_dwt + lig_i * 2 * H + H + 4 * c;
                    const float4 da = ld4(dw), db = ld4(dw + 64);
                    const float dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
                    const unsigned msh = mask_lig >> (4 * q);   // bit r (+16) <-> edge 4q + r (+16): immediate bit-field extracts
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = ((msh >> (r + 16 * hf)) & 1u) ? scv.S : 0.f;
#pragma unroll
                        for (int t = 0; t < 8; ++t) hv[t][r] = fmaf(m, dv[t], hv[t][r]);
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (p == 0 ? !has_prot : !has_lig) continue;
                    float Rm[5];
#pragma unroll
                    for (int s = 0; s < 5; ++s) Rm[s] = (lg0[hf] == (p == 1)) ? R[hf][s] : 0.f;
                    // split-f16, rbf tuples as the A operand, weight tuples as B
                    half4 A[4];
                    rbf_tuples(Rm, A);
                    const float* fb = lds_fv + (size_t)etype(p == 1, lig_i) * (8 * FRAG_BLK);
#pragma unroll
                    for (int tg = 0; tg < 8; tg += 2) {
                        const WTuples w0 = load_wtuples(fb, tg, lane), w1 = load_wtuples(fb, tg + 1, lane);
                        hv[tg] = MFMAH(A[0], w0.t1, hv[tg]);         hv[tg + 1] = MFMAH(A[0], w1.t1, hv[tg + 1]);
                        hv[tg] = MFMAH(A[1], w0.t1, hv[tg]);         hv[tg + 1] = MFMAH(A[1], w1.t1, hv[tg + 1]);
                        hv[tg] = MFMAH(A[2], w0.t2, hv[tg]);         hv[tg + 1] = MFMAH(A[2], w1.t2, hv[tg + 1]);
                        hv[tg] = MFMAH(A[3], w0.t3, hv[tg]);         hv[tg + 1] = MFMAH(A[3], w1.t3, hv[tg + 1]);
                    }
                }
                // LayerNorm per edge r (zero mean by construction): in-lane over t, across the 16 lanes of the row
                {   // edges r and r + 1 share the packed instructions
                    float2v va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        va = lo2(hv[t]) * lo2(hv[t]) + va;
                        vb = hi2(hv[t]) * hi2(hv[t]) + vb;
                    }
                    float2v ra, rb;
                    ra.x = fast_rsqrt(__builtin_fmaf(row16_sum(va.x), scv.c1, 1e-5f)) * scv.c2;
                    ra.y = fast_rsqrt(__builtin_fmaf(row16_sum(va.y), scv.c1, 1e-5f)) * scv.c2;
                    rb.x = fast_rsqrt(__builtin_fmaf(row16_sum(vb.x), scv.c1, 1e-5f)) * scv.c2;
                    rb.y = fast_rsqrt(__builtin_fmaf(row16_sum(vb.y), scv.c1, 1e-5f)) * scv.c2;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float2v g2 = splat2(gv[t]), b2 = splat2(bv[t]);
                        const float2v ya = (lo2(hv[t]) * ra) * g2 + b2;
                        const float2v yb = (hi2(hv[t]) * rb) * g2 + b2;
                        hv[t] = floatx4{fmaxf(ya.x, 0.f), fmaxf(ya.y, 0.f), fmaxf(yb.x, 0.f), fmaxf(yb.y, 0.f)};
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 8; ++t) s2[t] = MFMA(hv[t][r], w[hf][r], s2[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (c) next item: its PD / PS_k rows fly during the epilogue (the v path's gather registers are free again)
            {
                const gptr pdp = sbase(P + (size_t)inext * PROW);
                const unsigned o0 = (unsigned)ng.j0[0] * (PROW * 4) + (2 * H + 4 * q) * 4;
                const unsigned o1 = (unsigned)ng.j0[1] * (PROW * 4) + (2 * H + 4 * q) * 4;
                const unsigned oq = vop(16 * q);
#pragma unroll
                for (int t = 0; t < 8; ++t) { pd[t] = ldo4(pdp, oq + 64 * t); ps0[t] = ldo4(sbase(P), o0 + 64 * t); }
#pragma unroll
                for (int t = 0; t < 8; ++t) ps1[t] = ldo4(sbase(P), o1 + 64 * t);
            }
            // (the residual row and bias of this lane's two outputs were requested before the v path); the next node's neighbour
            // row in the E1 mapping is last in the queue: it is only needed at the top of the next iteration
            if (!CBGX_EDGE_EARLY_LOADS) { hres = ldo2(sbase(h + (size_t)i * H), on0); bias2 = ldo2(sbase(att + A_BBV), on0); }
            {
                const gptr nrow2 = sbase(nbr + (size_t)inext * KNN);
                const unsigned oq2 = vop(16 * q);
                nnb0 = ldoi4(nrow2, oq2); nnb1 = ldoi4(nrow2, oq2 + 64);
            }
            __builtin_amdgcn_sched_barrier(0);
            // epilogue: out[8a + cc] = sum_m Wbv[8a + cc][m] S[a][m] + bbv[8a + cc] sum_e alpha e_w ; this lane holds
            // S[a = c][m] for m = 64 hh + 16 q + 4 r + j in s2[4 hh + j][r].  Wbv rows live in LDS as 16-byte chunks
            // K = 16 hh + 4 q + j holding the four channels r = 0..3 of that (hh, q, j) -- the register quad of s2[4 hh + j], so
            // the 8 x 32 products are packed FMAs on the accumulators as they are -- with the chunk index XOR-swizzled by
            // wbv_swizzle(head).  A ds_read_b128 is served in four groups of 16 lanes that mix two values of q
            // ({0-3, 12-15 | q} with {4-11 | q + 1}, MI355X_MICROARCH.md LDS): the permutation makes the 16 lanes of every
            // group land on 16 different 16-byte slots (XOR with the head index itself was 2-way conflicted on every read:
            // 256 of the node's 880 LDS cycles).
            float o8[8];
            const float* lds_wbv = lds + (PP ? PP_WBV : IMG_WBV);
            const int swz = wbv_swizzle(c);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const float* wrow = lds_wbv + (size_t)(8 * c + cc) * H;
                float2v a2 = {0.f, 0.f};
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const floatx4 w4 = f4(ld4(wrow + (((16 * hh + 4 * q + j) ^ swz) << 2)));
                        a2 = lo2(w4) * lo2(s2[4 * hh + j]) + a2;
                        a2 = hi2(w4) * hi2(s2[4 * hh + j]) + a2;
                    }
                o8[cc] = xrow_sum(a2.x + a2.y);
            }
            // lane (c, q) writes outputs n = 8c + 2q, 8c + 2q + 1
            const float oa = q == 0 ? o8[0] : (q == 1 ? o8[2] : (q == 2 ? o8[4] : o8[6]));
            const float ob = q == 0 ? o8[1] : (q == 1 ? o8[3] : (q == 2 ? o8[5] : o8[7]));
            float2 o;
            o.x = hres.x + (oa + bias2.x * sw);
            o.y = hres.y + (ob + bias2.y * sw);
            sto2(sbase_w(out + (size_t)i * H), vop(n0 * 4), o);
        } else {
            // ---- h2x: v hidden edge-major, wv[e][a] = Wbv[a] . hid_v[e] + bbv[a] -------------------------------
            floatx4 wv[2];
            {
                const float* lds_brow = lds + IMG_WBV + 4 * lane;       // Wbv[head c][16 t + 4 q ..] at 256 t (layout.h)
                floatx4 acc[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { vd[t] = make_float4(vd[t].x * scv.S, vd[t].y * scv.S, vd[t].z * scv.S, vd[t].w * scv.S); acc[t] = f4(vs0[t]) * scv.S + f4(vd[t]); }
                wv[0] = edge_major_half<false>(acc, lg0[0], 1, lds_fv, lds_dwt, lds_ln, R[0], has_prot, has_lig, lig_i, lane, q,
                                               lds_brow, qrow, scv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = f4(vs1[t]) * scv.S + f4(vd[t]);
                wv[1] = edge_major_half<false>(acc, lg0[1], 1, lds_fv, lds_dwt, lds_ln, R[1], has_prot, has_lig, lig_i, lane, q,
                                               lds_brow, qrow, scv);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (c) next item: its PD / PS_k rows
            {
                const gptr pdp = sbase(P + (size_t)inext * PROW);
                const unsigned o0 = (unsigned)ng.j0[0] * (PROW * 4) + (2 * H + 4 * q) * 4;
                const unsigned o1 = (unsigned)ng.j0[1] * (PROW * 4) + (2 * H + 4 * q) * 4;
                const unsigned oq = vop(16 * q);
#pragma unroll
                for (int t = 0; t < 8; ++t) { pd[t] = ldo4(pdp, oq + 64 * t); ps0[t] = ldo4(sbase(P), o0 + 64 * t); }
#pragma unroll
                for (int t = 0; t < 8; ++t) ps1[t] = ldo4(sbase(P), o1 + 64 * t);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const gptr nrow2 = sbase(nbr + (size_t)inext * KNN);
                const unsigned oq2 = vop(16 * q);
                nnb0 = ldoi4(nrow2, oq2); nnb1 = ldoi4(nrow2, oq2 + 64);
            }
            float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float coef = (al[hf][r] * inv_den) * ((wv[hf][r] + bbv) * ew[hf][r]);
                    const float cf = 4 * q + r + 16 * hf < d ? coef : 0.f;
                    dx = fmaf(cf, g.xi - xj[hf][r][0], dx);
                    dy = fmaf(cf, g.yi - xj[hf][r][1], dy);
                    dz = fmaf(cf, g.zi - xj[hf][r][2], dz);
                }
            dx = wave_sum(dx); dy = wave_sum(dy); dz = wave_sum(dz);
            if (lane < 3) {
                const float v = (lane == 0 ? dx : (lane == 1 ? dy : dz)) * (1.f / HEADS);
                const float xin = lane == 0 ? g.xi : (lane == 1 ? g.yi : g.zi);
                if (dx_out) dx_out[3 * i + lane] = v;
                out[3 * i + lane] = xin + (gen[i] ? v : 0.f);
            }
        }
        // rotate the pipelined geometry
        nb0 = nnb0; nb1 = nnb1;
        lg0[0] = nlg[0]; lg0[1] = nlg[1]; dist0[0] = ndist[0]; dist0[1] = ndist[1];
        g = ng;
    }
}

template <bool X2H, int WAVES, bool LISTED>
__global__ __launch_bounds__(WAVES * 64) void edge_mfma_kernel(
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ h,
    const float* __restrict__ P, const float* __restrict__ Qt, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ deg, const uint8_t* __restrict__ lig, const uint8_t* __restrict__ gen,
    const float* __restrict__ e_w, int n_nodes, float* __restrict__ out, float* __restrict__ dx_out,
    const int* __restrict__ act, const int* __restrict__ act_count) {
    __shared__ __attribute__((aligned(16))) float lds[X2H ? (int)IMG_SIZE_X2H : (int)IMG_SIZE_H2X];
    __shared__ float lds_mu[G];
    edge_body<X2H, WAVES, LISTED, false>(lds, lds_mu, blockIdx.x, gridDim.x, att, x, h, P, Qt, nbr, deg, lig, gen, e_w, n_nodes, out,
                                         dx_out, act, act_count);
}

// x2h over TWO work lists in one launch: the protein-only destinations (`list_pp`: the node and all its neighbours are protein
// atoms; query folded in registers from q[N,128]) and the rest (`list_gen`: general kernel, folded rows from Qt[N,16,128], which
// node_qfold_kernel has produced for THESE nodes only).  The first `n_pp_wg` workgroups play the protein-only role, the others
// the general one; the split follows the list lengths (device-side counts; a general node is priced at 1.15 protein-only ones:
// mixed neighbourhoods run both source-class passes and read 8 KB of Qt) in multiples of 8 workgroups, so that a role's
// workgroup index modulo 8 is still its XCD.  One launch per layer, one LDS fill per workgroup, both lists' tails overlap.
#ifndef CBGX_DUAL_GEN_COST
#define CBGX_DUAL_GEN_COST 1.15f      // (an A/B knob of scripts/build_variant.py; the product build uses this value:
                                      //  1.0 / 1.15 / 1.3 / 1.45 / 1.7 -> 768 / 1185 / 1222 / 1220 / 748 us per launch in their sessions, profiles/ab_fwd_r04[df].log)
#endif
constexpr float DUAL_GEN_COST = CBGX_DUAL_GEN_COST;
// FULL_LAYER changes no code: the launches whose two lists together are all N nodes of a batch (the dominant kernel of the roofline)
// get their own kernel name, so that rocprofv3 --stats reports them apart from the cached / pruned layers' launches.
template <int WAVES, bool FULL_LAYER>
__global__ __launch_bounds__(WAVES * 64) void edge_x2h_dual_kernel(
    const float* __restrict__ att, const float* __restrict__ x, const float* __restrict__ h,
    const float* __restrict__ P, const float* __restrict__ Qt, const float* __restrict__ qbuf,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg, const uint8_t* __restrict__ lig,
    const uint8_t* __restrict__ gen, const float* __restrict__ e_w, int n_nodes, float* __restrict__ out,
    const int* __restrict__ list_pp, const int* __restrict__ count_pp, const int* __restrict__ list_gen,
    const int* __restrict__ count_gen) {
    __shared__ __attribute__((aligned(16))) float lds[IMG_SIZE_X2H];
    __shared__ float lds_mu[G];
    const int c_pp = *count_pp, c_gen = *count_gen;
    const int n_wg = gridDim.x;
    int n_pp_wg;
    // small input: both roles at `wv` nodes per workgroup, the spacing a single list of c_pp + c_gen items would get (the launcher adds
    // a workgroup for the second role's remainder): a proportional split left one role two nodes per wave, 36 instead of 27 us at one graph
    const int wv = edge_active_waves(c_pp + c_gen, n_wg > 1 ? n_wg - 1 : 1, WAVES);
    const int need_pp = (c_pp + wv - 1) / wv, need_gen = (c_gen + wv - 1) / wv;
    if (c_gen == 0) n_pp_wg = n_wg;
    else if (c_pp == 0) n_pp_wg = 0;
    else if (need_pp + need_gen <= n_wg) n_pp_wg = need_pp;
    else {
        const float share = (float)c_pp / ((float)c_pp + DUAL_GEN_COST * (float)c_gen);
        const int unit = n_wg >= 64 ? 8 : 1;
        n_pp_wg = (int)(share * (float)(n_wg / unit) + 0.5f) * unit;
        n_pp_wg = max(unit, min(n_wg - unit, n_pp_wg));
        if (n_wg < 2) n_pp_wg = 0;            // a single workgroup cannot play both roles: see the launcher (grid >= 2)
    }
    if ((int)blockIdx.x < n_pp_wg)
        edge_body<true, WAVES, true, true>(lds, lds_mu, blockIdx.x, n_pp_wg, att, x, h, P, qbuf, nbr, deg, lig, gen, e_w, n_nodes,
                                           out, nullptr, list_pp, count_pp);
    else
        edge_body<true, WAVES, true, false>(lds, lds_mu, (int)blockIdx.x - n_pp_wg, n_wg - n_pp_wg, att, x, h, P, Qt, nbr, deg, lig,
                                            gen, e_w, n_nodes, out, nullptr, list_gen, count_gen);
}

// ------------------------------------------------------------------------------------------------
// packing helpers for the LDS image
// ------------------------------------------------------------------------------------------------
// split-f16 pieces of the rbf columns of a (centred) first Linear W_a [128][340], block order [type][t] (layout.h): a weight
// w is carried as h = f16(w) (round to nearest) and l = f16(w - h)
// ---- pack kernels, one launch for all attention blocks (blockIdx.y = block; kernels.h PackBlocks) -------------------------------
// rbf table scales of a block (layout.h A_RBF_SC): per path the exponent kw that puts the largest |Wr| of all four edge types
// into [2^14, 2^15), clamped to RBF_KW_MAX; one 1024-thread workgroup per (path, block), from the CENTRED first Linears
__global__ __launch_bounds__(1024) void pack_rbf_scale_kernel(PackBlocks pb) {
    __shared__ float red[16];
    const int kv = blockIdx.x;
    float* att = pb.att[blockIdx.y];
    const float* w = att + (kv ? A_WAVC : A_WAKC);
    float mx = 0.f;
    // row m of the first Linear holds its 80 rbf columns contiguously at [m][NT .. NT + 80)
    for (int u = threadIdx.x; u < H * NT * G; u += 1024) mx = fmaxf(mx, fabsf(w[(size_t)(u / (NT * G)) * KV_IN + NT + u % (NT * G)]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
        const int E = (int)((__float_as_uint(mx) >> 23) & 0xffu);      // max in [2^(E-127), 2^(E-126))
        const int kw = max(RBF_KW_MIN, min(RBF_KW_MAX, 141 - E));
        const float S = ldexpf(1.f, kw + RBF_EXP);
        float* sc = att + A_RBF_SC + 4 * kv;
        sc[0] = S;
        sc[1] = 1.f / ((float)H * S * S);
        sc[2] = 1.f / S;
        sc[3] = (float)kw;
    }
}

// split-f16 pieces of the rbf columns of a (centred) first Linear W_a [128][340], block order [type][t] (layout.h): a weight
// w 2^kw is carried as h = f16(.) (round to nearest) and l = f16(. - h).  blockIdx.z: 0 = k table (edge-major), 1 = v table
// (x2h: channel-major B operand; h2x: edge-major), 2 = the edge-major v table of x2h blocks (training backward, A_FRAGV_EM)
__global__ void pack_frag_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (type*8 + t)*64 + lane
    if (idx >= NT * 8 * 64) return;
    float* att = pb.att[blockIdx.y];
    const bool x2h = pb.x2h[blockIdx.y] != 0;
    const int which = blockIdx.z;
    if (which == 2 && !x2h) return;
    const float* w_a = att + (which == 0 ? A_WAKC : A_WAVC);
    const int mode = which == 1 && x2h ? 1 : 0;
    float* dst = which == 0 ? att + A_IMG + IMG_FRAG_K : (which == 1 ? att + A_IMG + IMG_FRAG_V : att + A_FRAGV_EM);
    const int lane = idx & 63, t = (idx >> 6) & 7, type = idx >> 9;
    const int c = lane & 15, q = lane >> 4;
    const int m = mode == 0 ? 16 * t + c : 64 * (t >> 2) + 4 * c + (t & 3);
    const int kw = (int)att[A_RBF_SC + (which == 0 ? 0 : 4) + 3];
    _Float16 h[5], l[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const float w = ldexpf(w_a[(size_t)m * KV_IN + NT + G * type + 4 * s + q], kw);
        h[s] = (_Float16)w;
        l[s] = (_Float16)(w - (float)h[s]);
    }
    float* tb = dst + (size_t)type * 8 * FRAG_BLK;     // the type's 8 tiles: layout of load_wtuples (edge_common.h)
    _Float16* a0 = reinterpret_cast<_Float16*>(tb + frag_d0_index(t, lane));   // (d0, d1) = [h0 h1 h2 h3], (d2, d3) = [h4 l0 l1 l2]
    _Float16* a2 = reinterpret_cast<_Float16*>(tb + frag_d4_index(t, lane));   //  d4      = [l3 l4]
    a0[0] = h[0]; a0[1] = h[1]; a0[2] = h[2]; a0[3] = h[3];
    a0[4] = h[4]; a0[5] = l[0]; a0[6] = l[1]; a0[7] = l[2];
    a2[0] = l[3]; a2[1] = l[4];
}

// dWt[dst class lig_i][k|v][m] = Wt[type(src lig, lig_i)][m] - Wt[type(src prot, lig_i)][m]   (centred first Linears)
__global__ void pack_dwt_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // [lig_i][kv][m]
    if (idx >= 2 * 2 * H) return;
    float* att = pb.att[blockIdx.y];
    const int m = idx & 127, kv = (idx >> 7) & 1, li = idx >> 8;
    const float* w = att + (kv ? A_WAVC : A_WAKC);
    const int tl = li ? 0 : 1, tp = li ? 2 : 3;
    att[A_IMG + IMG_WT + idx] = w[(size_t)m * KV_IN + tl] - w[(size_t)m * KV_IN + tp];
}

// x2h blocks: second v Linear [128 n][128 m] as the edge kernel's epilogue reads it: column m = 64 hh + 16 q + 4 r + j goes to
// 16-byte chunk K = 16 hh + 4 q + j, position r; the chunk index is XOR-swizzled by wbv_swizzle(head n >> 3)
__global__ void pack_wbv_swz_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // n * 128 + m
    if (idx >= H * H) return;
    if (!pb.x2h[blockIdx.y]) {
        // h2x blocks: the 16 head rows [head][m] in the B-operand order of the value contraction (layout.h, image "wbv"):
        // lane (c = head, q) reads Wbv[c][16 t + 4 q + r] at (64 t + lane) * 4 + r
        if (idx >= HEADS * H) return;
        const int head = idx >> 7, m = idx & 127, t = m >> 4, q = (m >> 2) & 3, r = m & 3;
        pb.att[blockIdx.y][A_IMG + IMG_WBV + ((64 * t + 16 * q + head) << 2) + r] = pb.wv1[blockIdx.y][idx];
        return;
    }
    const int n = idx >> 7, m = idx & 127;
    const int chunk = 16 * (m >> 6) + 4 * ((m >> 4) & 3) + (m & 3), r = (m >> 2) & 3;
    pb.att[blockIdx.y][A_IMG + IMG_WBV + n * H + (((chunk ^ wbv_swizzle((n >> 3) & 15)) << 2) | r)] = pb.wv1[blockIdx.y][idx];
}

// x2h blocks: the LDS image of the protein-only kernel (layout.h A_IMG_PP), assembled from the general image this stream has just
// written (type-3 table slices, LayerNorm affine, swizzled second v Linear) plus the fold table of the second k Linear
__global__ void pack_pp_image_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int)PP_IMG_SIZE || !pb.x2h[blockIdx.y]) return;
    float* att = pb.att[blockIdx.y];
    const float* img = att + A_IMG;
    float v;
    if (idx < (int)PP_FRAG_V) v = img[IMG_FRAG_K + 3 * 8 * FRAG_BLK + idx];
    else if (idx < (int)PP_LN) v = img[IMG_FRAG_V + 3 * 8 * FRAG_BLK + (idx - PP_FRAG_V)];
    else if (idx < (int)PP_WBV) v = img[IMG_LN + (idx - PP_LN)];
    else if (idx < (int)PP_WFOLD) v = img[IMG_WBV + (idx - PP_WBV)];
    else {
        const int u = idx - (int)PP_WFOLD;                      // ((t * 8 + d) * 64 + lane) * 4 + r
        const int r = u & 3, lane = (u >> 2) & 63, d = (u >> 8) & 7, t = u >> 11;
        const int c = lane & 15, q = lane >> 4;
        v = pb.wk1[blockIdx.y][(size_t)(8 * c + d) * H + 16 * t + 4 * q + r] * 0.35355339059327376220f;   // same factor as A_WBK_FRAG
    }
    att[A_IMG_PP + idx] = v;
}

// centre the first Linears of k and v over their 128 output channels: wc = w - colmean(w), bc = b - mean(b)   (w [128][340]);
// blockIdx.x = column (KV_IN -> the bias), blockIdx.y = 2 block + (k | v)
__global__ void center_linear_kernel(PackBlocks pb) {
    const int col = blockIdx.x, blk = blockIdx.y >> 1, kv = blockIdx.y & 1;
    const float* w = kv ? pb.wv0[blk] : pb.wk0[blk];
    const float* b = kv ? pb.bv0[blk] : pb.bk0[blk];
    float* wc = pb.att[blk] + (kv ? A_WAVC : A_WAKC);
    float* bc = pb.att[blk] + (kv ? A_BAVC : A_BAKC);
    __shared__ float red[128];
    const int r = threadIdx.x;
    const float v = col < KV_IN ? w[(size_t)r * KV_IN + col] : b[r];
    red[r] = v;
    __syncthreads();
    for (int o = 64; o >= 1; o >>= 1) {
        if (r < o) red[r] += red[r + o];
        __syncthreads();
    }
    const float mean = red[0] * (1.f / 128.f);
    if (col < KV_IN) wc[(size_t)r * KV_IN + col] = v - mean; else bc[r] = v - mean;
}

hipError_t launch_pack_stage1(const PackBlocks& pb, hipStream_t s) {
    if (pb.n == 0) return hipSuccess;
    hipLaunchKernelGGL(center_linear_kernel, dim3(KV_IN + 1, 2 * pb.n), dim3(128), 0, s, pb);
    return hipGetLastError();
}

hipError_t launch_pack_stage2(const PackBlocks& pb, hipStream_t s) {
    if (pb.n == 0) return hipSuccess;
    hipError_t e = launch_pack_node_tables(pb, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_rbf_scale_kernel, dim3(2, pb.n), dim3(1024), 0, s, pb);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(NT * 8 * 64 / 256, pb.n, 3), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_dwt_kernel, dim3(2, pb.n), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_wbv_swz_kernel, dim3(H * H / 256, pb.n), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_pp_image_kernel, dim3((PP_IMG_SIZE + 255) / 256, pb.n), dim3(256), 0, s, pb);
    return hipGetLastError();
}

// cbgx_set_edge_workgroups (include/cbgx.h): upper bound of the persistent x2h grid, 0 = one workgroup per CU
// process-wide (the one piece of library state shared between host threads): relaxed atomic, read once per launch
static std::atomic<int> g_edge_wg_limit{0};
int set_edge_workgroup_limit(int n) { return g_edge_wg_limit.exchange(n < 0 ? 0 : n, std::memory_order_relaxed); }

hipError_t launch_edge_mfma(bool x2h, const float* att, const float* x, const float* h, const float* P,
                            const float* Qt, const int32_t* nbr, const int32_t* deg, const uint8_t* lig,
                            const uint8_t* gen, const float* e_w, int n_nodes, float* out, float* dx_out,
                            const int* act, const int* act_count, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    if ((size_t)n_nodes * PROW * sizeof(float) >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit byte offsets into P
    constexpr int W = 8;                        // waves per persistent workgroup: 2 per SIMD at <= 256 VGPRs per lane
    // sized for EDGE_MIN_WAVES nodes per workgroup: a short list spreads over more CUs with one wave per SIMD (edge_active_waves)
    int grid = (n_nodes + EDGE_MIN_WAVES - 1) / EDGE_MIN_WAVES;
    if (grid > 256) grid = 256;                 // persistent: one workgroup per CU (LDS-limited)
    const int wg_limit = g_edge_wg_limit.load(std::memory_order_relaxed);
    if (x2h && wg_limit >= 8 && grid > wg_limit) grid = wg_limit;   // the caller keeps CUs free for another stream
    if (grid >= 64) grid &= ~7;                 // multiple of 8 -> XCD-aware node partition
    profile_mark_begin(x2h ? (act ? K_EDGE_X2H_LISTED : K_EDGE_X2H) : K_EDGE_H2X, s);
#define CBGX_LAUNCH_EDGE(X2H_, L_)                                                                                  \
    hipLaunchKernelGGL((edge_mfma_kernel<X2H_, W, L_>), dim3(grid), dim3(W * 64), 0, s, att, x, h, P, Qt, nbr, deg, \
                       lig, gen, e_w, n_nodes, out, dx_out, act, act_count)
    if (x2h) {
        if (act) CBGX_LAUNCH_EDGE(true, true); else CBGX_LAUNCH_EDGE(true, false);
    } else {
        if (act) CBGX_LAUNCH_EDGE(false, true); else CBGX_LAUNCH_EDGE(false, false);
    }
#undef CBGX_LAUNCH_EDGE
    profile_mark_end(s);
    return hipGetLastError();
}

// x2h edge stage over the (protein-only, general) list pair of a layer: one launch of edge_x2h_dual_kernel.  `full_layer`: the two
// lists together are all N nodes (profile class of the dominant kernel) -- otherwise a cached / pruned layer.
hipError_t launch_edge_x2h_dual(const float* att, const float* x, const float* h, const float* P, const float* Qt,
                                const float* qbuf, const int32_t* nbr, const int32_t* deg, const uint8_t* lig,
                                const uint8_t* gen, const float* e_w, int n_nodes, float* out, const int* list_pp,
                                const int* count_pp, const int* list_gen, const int* count_gen, bool full_layer, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    if ((size_t)n_nodes * PROW * sizeof(float) >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit byte offsets into P
    constexpr int W = 8;
    int grid = (n_nodes + EDGE_MIN_WAVES - 1) / EDGE_MIN_WAVES + 1;     // + 1: each role rounds its list up to whole workgroups
    if (grid > 256) grid = 256;                 // persistent: one workgroup per CU (a multiple of 8 -> XCD-aware partition per role)
    const int wg_limit = g_edge_wg_limit.load(std::memory_order_relaxed);
    if (wg_limit >= 8 && grid > wg_limit) grid = wg_limit & ~7;
    if (grid < 2) grid = 2;                     // one workgroup per role at least
    profile_mark_begin(full_layer ? K_EDGE_X2H : K_EDGE_X2H_LISTED, s);
    if (full_layer)
        hipLaunchKernelGGL((edge_x2h_dual_kernel<W, true>), dim3(grid), dim3(W * 64), 0, s, att, x, h, P, Qt, qbuf, nbr, deg, lig, gen,
                           e_w, n_nodes, out, list_pp, count_pp, list_gen, count_gen);
    else
        hipLaunchKernelGGL((edge_x2h_dual_kernel<W, false>), dim3(grid), dim3(W * 64), 0, s, att, x, h, P, Qt, qbuf, nbr, deg, lig, gen,
                           e_w, n_nodes, out, list_pp, count_pp, list_gen, count_gen);
    profile_mark_end(s);
    return hipGetLastError();
}

}  // namespace cbgx

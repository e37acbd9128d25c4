// libcbgx -- node-level GEMMs of one attention block on the gfx950 matrix cores (split-f16 projection, exact-fp32 MFMA else).
//
//   node_proj :  P[N,640]   = h[N,128] @ Wn + bn          (PDk | PDv | PSk | PSv | q-hidden), LDS-staged Wn chunks
//   node_qmlp :  q[N,128]   = ReLU(LN(P[:,512:640])) @ Wq1^T + bq1
//   node_qfold:  Qt[N,16,128] = (1/sqrt 8) * q[N, 8a:8a+8] @ Wbk[8a:8a+8, :]      (key's 2nd Linear folded into q)
//
// Common shape: one wavefront owns 16 rows (nodes); its A operand is read straight from global memory in
// MFMA A layout using a K permutation (step (s4, j) uses k = 16 s4 + 4q + j, so the four q-lanes of a row read
// one contiguous 64-byte run per load instruction), the B operand comes from LDS in "fragment order" (written by cbgx_pack_weights so the
// global->LDS copy is linear and every ds_read_b128 is conflict-free), and output tiles use a column
// permutation (tile j of a group of four <-> column 4c + j) so each lane stores float4s.
// MFMA 16x16x4 maps: A[i=c][k=q], B[k=q][j=c], C reg r: [row 4q+r][col c];  lane l: c = l & 15, q = l >> 4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "layout.h"

namespace cbgx {

typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// split-f16 (edge_mfma.hip): an fp32 value v = hi + lo with hi = f16(v) (round to nearest), lo = f16(v - hi);
// a product is hi*hi + hi*lo + lo*hi on the f16 matrix pipe with fp32 accumulation (2^-22 relative)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define MFMAH32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float4 nld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float nxrow_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float nxrow_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Range-safe split-f16 (layout.h): a row of activations is multiplied by the power of two that puts its largest
// magnitude `mx` into [2^14, 2^15) before it is split into hi = f16(v), lo = f16(v - hi) -- every |v| >= mx 2^-17 then has a
// normal `lo` (2^-21 relative), smaller ones an absolute error of mx 2^-39 -- and the fp32 accumulator is multiplied by the inverse.
// Any finite input works (|h| > 65504 included); the exponent is clamped so that both factors stay normal fp32 numbers.
// Returns the up-scale, `inv` gets its reciprocal.  All lanes of a row must pass the same mx.
__device__ __forceinline__ float row_pow2(float mx, float& inv) {
    const int E = (int)((__float_as_uint(mx) >> 23) & 0xffu);      // mx in [2^(E-127), 2^(E-126))
    const int ka = max(-100, min(100, 141 - E));
    inv = __uint_as_float((unsigned)(127 - ka) << 23);
    return __uint_as_float((unsigned)(127 + ka) << 23);
}
// the inverse row scales in the MFMA C layout (register r <-> row 4q + r) from the A layout (lane c <-> row c)
__device__ __forceinline__ void rows_to_c_layout(float inv, int q, float (&rinv)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        rinv[r] = __uint_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * q + r), __float_as_uint(inv)));
}
// split 8 consecutive operand slots of a row (already loaded) after scaling by `up`
__device__ __forceinline__ void split8(const float (&v)[8], float up, half8& hi8, half8& lo8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float sv = v[j] * up;
        const _Float16 hi = (_Float16)sv;
        hi8[j] = hi;
        lo8[j] = (_Float16)(sv - (float)hi);
    }
}

// ------------------------------------------------------------------------------------------------
// node_proj: P[:, chunk] for one 64-column chunk per workgroup, the chunk's split-f16 table RESIDENT in LDS (32 KB, filled once);
// a 4-wave workgroup then streams 64-row tiles past it, one wave per 16 rows, with no barrier and no LDS write in the loop.
// (Round 4.  Until round 3 a workgroup kept its rows and streamed the ten chunk tables through a double-buffered LDS slot: every
// chunk was load -> wait -> ds_write -> barrier -> 48 MFMAs -> stores with nothing overlapped and two workgroups per CU, 106 us
// per 99.5 k-node launch against a store floor of 38 us.  Here four workgroups per CU = four waves per SIMD run independent
// load / split / MFMA / store chains, so one wave's L2 round trip is another wave's MFMA block.)
// The product runs in split-f16 on v_mfma_f32_16x16x32_f16 (K = 32 per instruction, 8 f16 per lane and operand): the 128
// features of a row are split per row tile and chunk (hi / lo, 4 + 4 operands per lane), the weights at pack time; 3 x 4 MFMAs per
// 16 x 16 output tile instead of 32 exact-fp32 ones.
// K slot j of instruction u in lane group q <-> k = 16 (2u + (j >> 2)) + 4q + (j & 3), so the four q-lanes of a row still read
// one contiguous 64-byte run of h per load.  Chunk table order: [part hi|lo][ct 4][u 4][lane 64][8 f16]
//   = split(Wn[k(u, q, j)][col = 64 ch + 4c + ct])
// Same arithmetic per output element as the streaming kernel it replaces (same operands, same MFMA order): P is bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int NP_CHUNK = 2 * 4 * 4 * 64 * 4;  // 8192 floats = 32 KB (two f16 per float slot)
constexpr int NP_CHUNKS = PROW / 64;          // 10
#ifndef CBGX_NPROJ_CPW
#define CBGX_NPROJ_CPW 2              // column chunks per workgroup (A/B knob of scripts/build_variant.py: 2 = two resident tables,
#endif                                // the rows of a tile loaded and split once per TWO chunks, two workgroups per CU)
constexpr int NP_CPW = CBGX_NPROJ_CPW;
constexpr int NP_WGS_PER_CU = NP_CPW == 1 ? 3 : 2;   // node_proj_kernel: __launch_bounds__(256, NP_WGS_PER_CU), NP_CPW x 32 KB of LDS each
constexpr int NP_RESIDENT_WGS = 256 * NP_WGS_PER_CU;

typedef float lds_fx4 __attribute__((ext_vector_type(4)));
// LDS fill of the float4 range [begin, end) by 256 threads with every load of a thread requested before its first store (at most
// MAXV per thread).  Written as a plain loop the compiler emits load -> wait -> ds_write per iteration: up to 16 dependent L2 round
// trips (~11 us) at the head of every launch.
template <int MAXV>
__device__ __forceinline__ void lds_fill_f4(float* lds, const float* src_base, int begin, int end, int tid) {
    const lds_fx4* src = reinterpret_cast<const lds_fx4*>(src_base);
    lds_fx4* dst = reinterpret_cast<lds_fx4*>(lds);
    lds_fx4 v[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int t = begin + tid + 256 * u;
        v[u] = src[t < end ? t : end - 1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int t = begin + tid + 256 * u;
        if (t < end) dst[t] = v[u];
    }
}

// One wave's view of a 16-row tile, in two stages so that no load of the loop waits for another one of the same iteration:
//   ProjIdx  the row numbers (A-operand row of lane c; the four output rows 4q + r of the C layout), requested TWO tiles ahead --
//            with a work list they are a gather through `rows`;
//   ProjTile the raw feature rows in A-operand order and the destination classes, requested ONE tile ahead from the indices
//            that arrived during the previous tile.
// Rows past the end are clamped to the last row (orow = -1 masks their stores): every load is unconditional -- a predicated load
// compiles to branch + load + wait per element.
struct ProjIdx { int arow; int orow[4]; };
struct ProjTile {
    float4 hv[8];       // h[arow][32u + 4q ..], h[arow][32u + 16 + 4q ..]  (u = 0..3)
    int orow[4];        // output row, -1 past the end
    int lgr;            // bit r: output row r is a ligand atom
};
template <bool LISTED>
__device__ __forceinline__ ProjIdx proj_idx_load(const int* __restrict__ rows, int n_rows, int row0, int c, int q) {
    ProjIdx x;
    const int ak = min(row0 + c, n_rows - 1);
    x.arow = LISTED ? rows[ak] : ak;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = row0 + 4 * q + r;
        const int kk = min(k, n_rows - 1);
        const int row = LISTED ? rows[kk] : kk;
        x.orow[r] = k < n_rows ? row : -1 - row;       // past the end: the clamped row, encoded negative
    }
    return x;
}
__device__ __forceinline__ ProjTile proj_tile_load(const float* __restrict__ h, const uint8_t* __restrict__ lig, const ProjIdx& x,
                                                   int q) {
    ProjTile t;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        t.hv[2 * u] = nld4(h + (size_t)x.arow * H + 32 * u + 4 * q);
        t.hv[2 * u + 1] = nld4(h + (size_t)x.arow * H + 32 * u + 16 + 4 * q);
    }
    int flags[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t.orow[r] = x.orow[r] >= 0 ? x.orow[r] : -1;
        flags[r] = lig[x.orow[r] >= 0 ? x.orow[r] : -1 - x.orow[r]];
    }
    t.lgr = (flags[0] != 0) | ((flags[1] != 0) << 1) | ((flags[2] != 0) << 2) | ((flags[3] != 0) << 3);
    return t;
}

// one 16-row tile of one wave against the workgroup's resident chunk(s): row scale + split ONCE, then per chunk 48 MFMAs (B from LDS)
// and the scaled stores
struct ProjChunk { float4 bP, bL, ci; int ch; };
__device__ __forceinline__ void proj_tile_compute(const ProjTile& cur, const half8* __restrict__ Bh0, const ProjChunk (&pc)[NP_CPW],
                                                  int n_ch, float* __restrict__ P, int c, int q) {
    half8 ah[4], al[4];
    float rinv[4];
    {
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {     // chains of max(max(m, |a|), |b|): one v_max3_f32 with |.| source modifiers each
            mx = fmaxf(fmaxf(mx, fabsf(cur.hv[u].x)), fabsf(cur.hv[u].y));
            mx = fmaxf(fmaxf(mx, fabsf(cur.hv[u].z)), fabsf(cur.hv[u].w));
        }
        float inv;
        const float up = row_pow2(nxrow_max(mx), inv);
        rows_to_c_layout(inv, q, rinv);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v0 = cur.hv[2 * u], v1 = cur.hv[2 * u + 1];
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            split8(v, up, ah[u], al[u]);
        }
    }
#pragma unroll
    for (int k = 0; k < NP_CPW; ++k) {
        if (k >= n_ch) break;          // workgroup-uniform: the last group of a launch may hold fewer chunks
        const half8* Bh = Bh0 + (size_t)k * (NP_CHUNK / 4);      // chunk k's table: NP_CHUNK floats = NP_CHUNK / 4 half8
        const half8* Bl = Bh + 4 * 4 * 64;
        floatx4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            half8 bh[4], bl[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { bh[ct] = Bh[(ct * 4 + u) * 64]; bl[ct] = Bl[(ct * 4 + u) * 64]; }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMAH32(ah[u], bl[ct], acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMAH32(al[u], bh[ct], acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMAH32(ah[u], bh[ct], acc[ct]);
        }
        const float4 bP = pc[k].bP, bL = pc[k].bL, ci = pc[k].ci;
        const int ch = pc[k].ch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (cur.orow[r] >= 0) {
                const float4 b = ((cur.lgr >> r) & 1) ? bL : bP;
                float4 o = {fmaf(acc[0][r] * rinv[r], ci.x, b.x), fmaf(acc[1][r] * rinv[r], ci.y, b.y),
                            fmaf(acc[2][r] * rinv[r], ci.z, b.z), fmaf(acc[3][r] * rinv[r], ci.w, b.w)};
                *reinterpret_cast<float4*>(P + (size_t)cur.orow[r] * PROW + 64 * ch + 4 * c) = o;
            }
        }
    }
}

// `rows` / `n_rows_ptr` (LISTED): compute only the listed rows (device-side count, no host sync); results are
// written to their natural positions P[rows[k]].  `chunk_mask`: which of the 10 column chunks to produce; workgroup (x, y) owns
// the y-th selected chunk and the row tiles x, x + gridDim.x, ...
template <bool LISTED>
__global__ __launch_bounds__(256, NP_WGS_PER_CU) void node_proj_kernel(const float* __restrict__ att, const float* __restrict__ h,
                                                           const uint8_t* __restrict__ lig, float* __restrict__ P,
                                                           int n_nodes, const int* __restrict__ rows,
                                                           const int* __restrict__ n_rows_ptr, unsigned chunk_mask) {
    __shared__ __attribute__((aligned(16))) float lds[NP_CPW * NP_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    const int n_rows = LISTED ? *n_rows_ptr : n_nodes;
    const int n_tiles = (n_rows + 63) / 64;
    int chs[NP_CPW], n_ch = 0;       // workgroup (x, y) owns the selected chunks NP_CPW y .. NP_CPW y + NP_CPW - 1
    {
        unsigned m = chunk_mask;
        for (unsigned k = 0; k < NP_CPW * blockIdx.y; ++k) m &= m - 1;
#pragma unroll
        for (int k = 0; k < NP_CPW; ++k) {
            chs[k] = m ? __ffs(m) - 1 : 0;
            n_ch += m != 0;
            m &= m - 1;
        }
    }
    if ((int)blockIdx.x >= n_tiles || n_ch == 0) return;   // whole workgroup: nothing to do
    const int step = 64 * (int)gridDim.x;
    int row0 = blockIdx.x * 64 + wave * 16;
    // first tile's rows requested before the table fill: the round trips overlap
    ProjTile cur;
    ProjIdx idx1;
    {
        const ProjIdx idx0 = proj_idx_load<LISTED>(rows, n_rows, row0, c, q);
        idx1 = proj_idx_load<LISTED>(rows, n_rows, row0 + step, c, q);
        cur = proj_tile_load(h, lig, idx0, q);
    }
    const float* bias = att + A_BN2;  // [dst class][640]: bias + type column of a protein source
    ProjChunk pc[NP_CPW];
#pragma unroll
    for (int k = 0; k < NP_CPW; ++k) {
        const int ch = chs[k];
        if (k < n_ch) lds_fill_f4<NP_CHUNK / 4 / 256>(lds + (size_t)k * NP_CHUNK, att + A_NPROJ_FRAG + (size_t)ch * NP_CHUNK, 0, NP_CHUNK / 4, tid);
        pc[k].ch = ch;
        pc[k].bP = nld4(bias + 64 * ch + 4 * c); pc[k].bL = nld4(bias + PROW + 64 * ch + 4 * c);
        pc[k].ci = nld4(att + A_NPROJ_CINV + 64 * ch + 4 * c);     // 2^-kc of this lane's four columns
    }
    __syncthreads();
    const half8* Bh = reinterpret_cast<const half8*>(lds) + lane;   // chunk k at + k NP_CHUNK / 4: [ct][u][lane]
    // two tiles per trip, (current, next) tile registers swapping roles instead of being copied (40 registers per tile)
    ProjTile other;
    ProjIdx idx2;
    for (int tile = blockIdx.x; tile < n_tiles; tile += 2 * gridDim.x, row0 += 2 * step) {
        // the next tile's rows (indices arrived during the previous tile) and the indices of the tile after it: in flight during
        // the split and the MFMAs below
        other = proj_tile_load(h, lig, idx1, q);
        idx2 = proj_idx_load<LISTED>(rows, n_rows, row0 + 2 * step, c, q);
        __builtin_amdgcn_sched_barrier(0);
        proj_tile_compute(cur, Bh, pc, n_ch, P, c, q);
        if (tile + (int)gridDim.x >= n_tiles) break;
        cur = proj_tile_load(h, lig, idx2, q);
        idx1 = proj_idx_load<LISTED>(rows, n_rows, row0 + 3 * step, c, q);
        __builtin_amdgcn_sched_barrier(0);
        proj_tile_compute(other, Bh, pc, n_ch, P, c, q);
    }
}

// ------------------------------------------------------------------------------------------------
// node_qmlp: LayerNorm + ReLU on the q-hidden quarter of P (A layout: LN is in-lane + across q), then
// q = z @ Wq1^T + bq1 in split-f16 (K = 32 per MFMA, same slot map as node_proj).
// Wq1 tables [part hi|lo][nt 8][u 4][lane][8 f16] = split(Wq1[n = 64(nt>>2) + 4c + (nt&3)][k = 16 (2u + (j>>2)) + 4q + (j&3)]).
// ------------------------------------------------------------------------------------------------
constexpr int NQ_FRAG = 8 * 8 * 64 * 4;  // 16384 floats = 64 KB

__global__ __launch_bounds__(256) void node_qmlp_kernel(const float* __restrict__ att, const float* __restrict__ P,
                                                        float* __restrict__ qout, int n_nodes,
                                                        const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    __shared__ __attribute__((aligned(16))) float lds[NQ_FRAG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    if ((int)blockIdx.x >= ((rows ? *n_rows_ptr : n_nodes) + 63) / 64) return;   // no tile for this workgroup
    const int grp_begin = gridDim.y > 1 ? blockIdx.y : 0, grp_end = gridDim.y > 1 ? blockIdx.y + 1 : 2;
    // group g owns the tiles nt = 4g .. 4g + 3 of both parts (hi | lo): two float4 ranges of at most NQ_FRAG / 8 float4 (8 per thread)
#pragma unroll
    for (int part = 0; part < 2; ++part)
        lds_fill_f4<8>(lds, att + A_WQ1_FRAG, part * (NQ_FRAG / 8) + grp_begin * (NQ_FRAG / 16),
                       part * (NQ_FRAG / 8) + grp_end * (NQ_FRAG / 16), tid);
    __syncthreads();
    const int n_rows = rows ? *n_rows_ptr : n_nodes;
    const int n_tiles = (n_rows + 63) / 64;
    // bias and column scale of the (at most two) column groups: once per workgroup.  Loaded inside the group loop their wait sat
    // INSIDE each `if (row is valid)` block of the epilogue -- four `s_waitcnt vmcnt(0); global_store` in a row, every store waiting
    // for the one before it to be acknowledged (loads and stores share the counter)
    float4 b4g[2], cig[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int grp = min(grp_begin + g, grp_end - 1);
        b4g[g] = nld4(att + A_BQ1 + 64 * grp + 4 * c);
        cig[g] = nld4(att + A_WQ1_CINV + 64 * grp + 4 * c);
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 64 + wave * 16;
        if (row0 >= n_rows) continue;      // (wave-uniform: the last tile's trailing waves)
        const int ak = min(row0 + c, n_rows - 1);
        const int arow = rows ? rows[ak] : ak;
        int orow[4];      // the rows this lane writes: other lanes' list entries, by ds_bpermute (were four more dependent loads)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = row0 + 4 * q + r;
            const int o = __builtin_amdgcn_ds_bpermute((4 * q + r) << 2, arow);      // lane 4q + r holds row 4q + r of the tile
            orow[r] = k < n_rows ? o : -1;
        }
        float z[32];
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = nld4(P + (size_t)arow * PROW + 4 * H + 16 * u + 4 * q);
            z[4 * u] = v.x; z[4 * u + 1] = v.y; z[4 * u + 2] = v.z; z[4 * u + 3] = v.w;
            sm += (v.x + v.y) + (v.z + v.w);
        }
        const float mean = nxrow_sum(sm) * (1.f / H);
        float var = 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) { z[u] -= mean; var += z[u] * z[u]; }
        const float rstd = 1.f / sqrtf(nxrow_sum(var) * (1.f / H) + 1e-5f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 g = nld4(att + A_LNQ_G + 16 * u + 4 * q), b = nld4(att + A_LNQ_B + 16 * u + 4 * q);
            z[4 * u + 0] = fmaxf(z[4 * u + 0] * rstd * g.x + b.x, 0.f);
            z[4 * u + 1] = fmaxf(z[4 * u + 1] * rstd * g.y + b.y, 0.f);
            z[4 * u + 2] = fmaxf(z[4 * u + 2] * rstd * g.z + b.z, 0.f);
            z[4 * u + 3] = fmaxf(z[4 * u + 3] * rstd * g.w + b.w, 0.f);
        }
        half8 zh[4], zl[4];   // z[8u + j] <-> k = 16 (2u + (j >> 2)) + 4q + (j & 3): exactly the order z was loaded in
        float rinv[4];
        {
            float mx = 0.f;   // z >= 0 after the ReLU
#pragma unroll
            for (int u = 0; u < 32; ++u) mx = fmaxf(mx, z[u]);
            float inv;
            const float up = row_pow2(nxrow_max(mx), inv);
            rows_to_c_layout(inv, q, rinv);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v[8] = {z[8 * u], z[8 * u + 1], z[8 * u + 2], z[8 * u + 3], z[8 * u + 4], z[8 * u + 5], z[8 * u + 6], z[8 * u + 7]};
                split8(v, up, zh[u], zl[u]);
            }
        }
        const half8* Bh = reinterpret_cast<const half8*>(lds) + lane;   // [nt][u][lane]
        const half8* Bl = Bh + 8 * 4 * 64;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = grp_begin + gi;
            if (grp >= grp_end) break;
            const float4 b4 = b4g[gi], ci = cig[gi];
            floatx4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                half8 bh[4], bl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { bh[j] = Bh[((grp * 4 + j) * 4 + u) * 64]; bl[j] = Bl[((grp * 4 + j) * 4 + u) * 64]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = MFMAH32(zh[u], bl[j], acc[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = MFMAH32(zl[u], bh[j], acc[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = MFMAH32(zh[u], bh[j], acc[j]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (orow[r] >= 0) {
                    float4 o = {fmaf(acc[0][r] * rinv[r], ci.x, b4.x), fmaf(acc[1][r] * rinv[r], ci.y, b4.y),
                                fmaf(acc[2][r] * rinv[r], ci.z, b4.z), fmaf(acc[3][r] * rinv[r], ci.w, b4.w)};
                    *reinterpret_cast<float4*>(qout + (size_t)orow[r] * H + 64 * grp + 4 * c) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// node_qfold: Qt[row][a][m] = sum_cc q[row][8a+cc] * Wbk[8a+cc][m] / sqrt(8).  K = 8 per head = 2 MFMA steps with
// cc = 2 kk + step.  Wbk fragments [a 16][g 2][lane][8: j*2+step] = Wbk[8a + 2q + step][64g + 4c + j] / sqrt(8).
// ------------------------------------------------------------------------------------------------
constexpr int NF_FRAG = 16 * 2 * 64 * 8;  // 16384 floats = 64 KB

// HP = heads per workgroup (HEADS / gridDim.y: 16, or 4 when the heads of a row tile are spread over four workgroups).  Round 5: a
// row tile's HP query pairs are requested together, before the first head's MFMAs -- one global round trip per head, each behind a
// full wait, was the kernel's inner loop (`load; s_waitcnt vmcnt(0); 16 MFMAs; stores`, scripts/isa_report.py) -- and the rows a
// lane writes come from the other lanes' list entries by ds_bpermute instead of four more dependent loads.
template <int HP>
__global__ __launch_bounds__(256) void node_qfold_kernel(const float* __restrict__ att, const float* __restrict__ qin,
                                                         float* __restrict__ Qt, int n_nodes,
                                                         const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    __shared__ __attribute__((aligned(16))) float lds[NF_FRAG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    const int n_rows = rows ? *n_rows_ptr : n_nodes;
    const int n_tiles = (n_rows + 63) / 64;
    if ((int)blockIdx.x >= n_tiles) return;   // no tile for this workgroup
    const int a_begin = blockIdx.y * HP, a_end = a_begin + HP;
    auto list_row = [&](int tile) {      // the list entry of this lane's A-operand row (clamped past the end)
        const int ak = min(tile * 64 + wave * 16 + c, n_rows - 1);
        return rows ? rows[ak] : ak;
    };
    auto load_queries = [&](int arow, float2 (&qv)[HP]) {
#pragma unroll
        for (int k = 0; k < HP; ++k) qv[k] = *reinterpret_cast<const float2*>(qin + (size_t)arow * H + 8 * (a_begin + k) + 2 * q);
    };
    // the first tile's list entry travels with the LDS fill, its queries with the barrier behind it (a workgroup of a listed launch
    // has ONE tile: fill -> barrier -> entry -> queries -> MFMAs was four dependent round trips for 16 x HP x 512 bytes of output)
    int arow = list_row(blockIdx.x);
    lds_fill_f4<HP>(lds, att + A_WBK_FRAG, a_begin * (NF_FRAG / 64), a_end * (NF_FRAG / 64), tid);   // HP float4 per thread
    float2 qv[HP];
    load_queries(arow, qv);
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 64 + wave * 16;
        if (tile != (int)blockIdx.x) {
            arow = list_row(tile);
            load_queries(arow, qv);
        }
        if (row0 >= n_rows) continue;      // (wave-uniform: the last tile's trailing waves)
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = row0 + 4 * q + r;
            const int o = __builtin_amdgcn_ds_bpermute((4 * q + r) << 2, arow);      // lane 4q + r holds row 4q + r of the tile
            orow[r] = k < n_rows ? o : -1;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ONE wait for all of them here: a `s_waitcnt vmcnt(n)` inside the head loop also counts the stores of the heads before it
        // (loads and stores share the counter on this part), i.e. every head would wait for the previous head's 8 KB to be written
#pragma unroll
        for (int k = 0; k < HP; ++k) asm volatile("" : "+v"(qv[k].x), "+v"(qv[k].y));
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            const int a = a_begin + k;
            const float2 qa = qv[k];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float* fb = lds + ((a * 2 + g) * 64 + lane) * 8;
                const float4 b0 = nld4(fb), b1 = nld4(fb + 4);
                floatx4 acc[4];
                const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[0] = MFMA(qa.x, b0.x, zero); acc[1] = MFMA(qa.x, b0.z, zero);
                acc[2] = MFMA(qa.x, b1.x, zero); acc[3] = MFMA(qa.x, b1.z, zero);
                acc[0] = MFMA(qa.y, b0.y, acc[0]); acc[1] = MFMA(qa.y, b0.w, acc[1]);
                acc[2] = MFMA(qa.y, b1.y, acc[2]); acc[3] = MFMA(qa.y, b1.w, acc[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (orow[r] >= 0) {
                        float4 o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                        *reinterpret_cast<float4*>(Qt + ((size_t)orow[r] * HEADS + a) * H + 64 * g + 4 * c) = o;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// node_stage: the whole node stage of one attention block in ONE launch -- projection, query MLP, query fold -- built for
// LATENCY: small batches (1 - 10 graphs) are bound by ~60 dependent launches of ~10 us per step, and three of every four
// or five launches of a block were the node kernels above.  Here a 16-wave workgroup owns ONE 16-row tile and spreads the
// columns over its waves, so that each of the three dependent phases costs one round of operand loads:
//   phase 1  20 units of 32 projection columns over the 16 waves (split-f16, 24 MFMAs per unit), B operands straight from
//            the packed tables in global memory (L2-resident; no LDS staging, nothing to wait for but the wave's own loads);
//            the two q-hidden chunks go first and also land in a [16][128] LDS tile
//   phase 2  waves 0..7: LayerNorm + ReLU of the tile (each wave for itself), one 16-column tile of the query MLP's second
//            Linear each (split-f16, 12 MFMAs) -> q into a second LDS tile
//   phase 3  wave a: the fold of head a (exact fp32, 16 MFMAs) -> Qt[:, a]
// Every workgroup re-reads the block's 448 KB of tables (28 KB per row), which is why this kernel is only used up to
// CBGX_NODE_STAGE_MAX_ROWS rows; beyond that the three throughput kernels above take over.
// ------------------------------------------------------------------------------------------------
constexpr int NS_TPITCH = H + 4;   // LDS tile row pitch (floats)
#ifndef CBGX_NODE_STAGE_WAVES_DEFAULT
#define CBGX_NODE_STAGE_WAVES_DEFAULT 8
#endif
constexpr int NODE_STAGE_WAVES = CBGX_NODE_STAGE_WAVES_DEFAULT;   // waves per node_stage workgroup (launch_node_stage_grid)
#ifndef CBGX_NS_P2_EARLY
#define CBGX_NS_P2_EARLY 1
#endif
#ifndef CBGX_NS_LOADS_FIRST
#define CBGX_NS_LOADS_FIRST 1
#endif

// Several jobs in one launch (blockIdx.y = job, kernels.h NodeStageJobs): every job is the stage -- or, `proj_only`, just its phase 1
// -- of one attention block on one row list with one column-chunk mask, all on the same input features `h`:
//   an h2x block        the full stage on the movable rows (own columns) + the PS columns of the rows that can be their neighbours
//   an x2h block        the full stage on all rows, or (cached / pruned layers) on the destination list + PS on the source list
//   a small input's layer   BOTH: the h2x block of layer l and the x2h block of layer l + 1 read the same h_{l+1}, so their node
//                       stages are one launch (round 5) -- until then the second ran on an auxiliary stream next to the first, and
//                       the fork / join events cost the caller's queue ~7 us each, twice per layer (profiles/step_timeline_r05a_*)
// A launch is bound by its longest job; small batches are bound by launch boundaries, which is what this removes.
// NW = 16 or 8 waves per workgroup.  A 16-wave workgroup has a CU to itself (128 registers per wave: four waves per SIMD), so a
// launch with more than 256 busy (tile, job) pairs runs in two rounds of a ~14 us dependent chain: 30 us per layer at the 10-graph
// batch of sample.py (276 row tiles x {x2h stage, h2x stage, h2x source columns}; profiles/step_timeline_r05z_p1s10.json).  With 8
// waves a CU holds two workgroups and every wave takes two or three units / two heads in sequence: a longer chain, one round.
// Which wave computes a column does not enter its arithmetic: both variants give the same bits.
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void node_stage_kernel(NodeStageJobs jobs, const float* __restrict__ h,
                                                                const uint8_t* __restrict__ lig, int n_nodes) {
    __shared__ __attribute__((aligned(16))) float qh[16][NS_TPITCH];
    __shared__ __attribute__((aligned(16))) float qt[16][NS_TPITCH];
    // phase 2's per-column constants (query LayerNorm gamma / beta, second Linear's bias and column scale), staged once per
    // workgroup: read from global memory inside phase 2 they were one exposed L2 round trip per 16 columns -- the emitted code was
    // `load x2, s_waitcnt vmcnt(0)` eight times in a row, a third of the kernel's 17 - 19 us at one graph (scripts/isa_report.py)
    __shared__ __attribute__((aligned(16))) float cst[4][H];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NodeStageJob& jb = jobs.j[blockIdx.y];       // kernel arguments: scalar loads
    const float* __restrict__ att = jb.att;
    float* __restrict__ P = jb.P;
    float* __restrict__ qout = jb.q;
    float* __restrict__ Qt = jb.Qt;
    const int* __restrict__ rows = jb.rows;
    const int* __restrict__ n_rows_ptr = jb.n_rows;
    const unsigned chunk_mask = jb.chunk_mask;
    const bool proj_only = jb.proj_only != 0;        // workgroup-uniform
    const uint8_t* __restrict__ fold_flag = jb.fold_flag;
    const int n_rows = rows ? *n_rows_ptr : n_nodes;
    const int n_tiles = (n_rows + 15) / 16;
    if ((int)blockIdx.x >= n_tiles) return;
    if (!proj_only) {       // visible after the barrier that ends phase 1
#pragma unroll
        for (int u = tid; u < 4 * H; u += NW * 64) {
            const int k = u >> 7, m = u & (H - 1);
            cst[k][m] = att[(k == 0 ? A_LNQ_G : (k == 1 ? A_LNQ_B : (k == 2 ? A_BQ1 : A_WQ1_CINV))) + m];
        }
    }
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int row0 = tl * 16;
        // the lane coordinates are re-materialised per tile: every per-lane address below is otherwise loop-invariant, gets hoisted
        // out of the tile loop as a 64-bit pointer pair and spilled (30 scratch reloads, each a memory round trip of this
        // latency-bound kernel, once phase 1 keeps a unit's sixteen operand loads in flight together)
        int lane;      // (asm: a plain mbcnt is hoisted and spilled like everything else)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int c = lane & 15, q = lane >> 4;
        // One dependent index load per tile: lane (c, q) reads the list entry of row c (its A-operand row); the rows it WRITES
        // (4q + r) are other lanes' entries, fetched by ds_bpermute, and the per-row predicates are ballots over lanes 0..15 --
        // until round 5 these were eight more index / flag loads per lane behind the first, serialised by the register allocator
        // (load, full wait, spill) in front of phase 1.
        const int ak = min(row0 + c, n_rows - 1);
        const int arow = rows ? rows[ak] : ak;
        const uint8_t* __restrict__ ff = fold_flag ? fold_flag : lig;     // (no flag array: read `lig` again, result ignored)
        unsigned lig_b = lig[arow], fold_b = ff[arow];
        // ---- phase 1: projection, 20 units of 32 columns (half a chunk); the q-hidden chunks first: wave w takes unit w, waves
        // 0..3 a second one.  One unit's 16 operand loads are all in flight at once (64 VGPRs of the 128 a 16-wave group gets).
        half8 ah[4], al[4];
        float rinv[4];
        float hv[4][8];
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v0 = nld4(h + (size_t)arow * H + 32 * u + 4 * q), v1 = nld4(h + (size_t)arow * H + 32 * u + 16 + 4 * q);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) { hv[u][j] = v[j]; mx = fmaxf(mx, fabsf(v[j])); }
        }
        asm volatile("" : "+v"(lig_b), "+v"(fold_b));      // both loads issued here, unconditionally, ahead of the row's features
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = row0 + 4 * q + r;
            const int o = __builtin_amdgcn_ds_bpermute((4 * q + r) << 2, arow);       // lane 4q + r (q' = 0) holds row 4q + r of the tile
            orow[r] = k < n_rows ? o : -1;
        }
        // lgm bit r: row 4q + r is a ligand row; fm bit r: its folded query is wanted (phase 3) -- rows past the list's end: no
        const unsigned valid16 = n_rows - row0 >= 16 ? 0xffffu : (1u << (n_rows - row0)) - 1u;
        const unsigned lig16 = (unsigned)__ballot(lig_b != 0u) & 0xffffu;
        const unsigned fold16 = (fold_flag ? (unsigned)__ballot(fold_b != 0u) & 0xffffu : 0xffffu) & valid16;
        const unsigned lgm = (lig16 >> (4 * q)) & 15u, fm = (fold16 >> (4 * q)) & 15u;
        {
            float inv;
            const float up = row_pow2(nxrow_max(mx), inv);
            rows_to_c_layout(inv, q, rinv);
#pragma unroll
            for (int u = 0; u < 4; ++u) split8(hv[u], up, ah[u], al[u]);
        }
        for (int unit = wave; unit < 2 * NP_CHUNKS; unit += NW) {
            const int ch = unit < 4 ? 8 + (unit >> 1) : (unit - 4) >> 1, half = unit & 1;   // units 0..3: chunks 8, 9
            if (!(((chunk_mask | (proj_only ? 0u : 0x300u)) >> ch) & 1u)) continue;
            const half8* Bh = reinterpret_cast<const half8*>(att + A_NPROJ_FRAG + (size_t)ch * NP_CHUNK) + (2 * half) * 4 * 64 + lane;
            const half8* Bl = Bh + 4 * 4 * 64;
            half8 bh[4][2], bl[4][2];     // [u][ct]
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { bh[u][ct] = Bh[(ct * 4 + u) * 64]; bl[u][ct] = Bl[(ct * 4 + u) * 64]; }
#if CBGX_NS_LOADS_FIRST
            __builtin_amdgcn_sched_barrier(0);      // all sixteen in flight before the first MFMA waits for one
#endif
            const float* bias = att + A_BN2 + 64 * ch + 4 * c + 2 * half;   // [dst class][640]: bias + type column of a protein source
            const float2 bP = *reinterpret_cast<const float2*>(bias), bL = *reinterpret_cast<const float2*>(bias + PROW);
            const float2 ci = *reinterpret_cast<const float2*>(att + A_NPROJ_CINV + 64 * ch + 4 * c + 2 * half);
            floatx4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = MFMAH32(ah[u], bl[u][ct], acc[ct]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = MFMAH32(al[u], bh[u][ct], acc[ct]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = MFMAH32(ah[u], bh[u][ct], acc[ct]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool lg = (lgm >> r) & 1u;
                const float2 o = {fmaf(acc[0][r] * rinv[r], ci.x, lg ? bL.x : bP.x),
                                  fmaf(acc[1][r] * rinv[r], ci.y, lg ? bL.y : bP.y)};
                int orw = orow[r];      // (opaque: the 64-bit row offsets are otherwise kept across the unit loop, and spilled)
                asm volatile("" : "+v"(orw));
                if (orw >= 0) *reinterpret_cast<float2*>(P + (size_t)orw * PROW + 64 * ch + 4 * c + 2 * half) = o;
                if (ch >= 8) *reinterpret_cast<float2*>(&qh[4 * q + r][64 * (ch - 8) + 4 * c + 2 * half]) = o;
            }
        }
        if (proj_only) continue;     // the source job ends with the projection (no barrier was entered: uniform per workgroup)
        __syncthreads();
        // ---- phase 2: query MLP, output tile nt = wave (16 columns 64 (nt >> 2) + 4c + (nt & 3)) ---------------------------
        if (wave < 8) {
#if CBGX_NS_P2_EARLY
            // the first output tile's B operands are requested before the LayerNorm of the row, which they do not depend on (not
            // before the barrier: from there the compiler moves them up into phase 1 and spills)
            half8 bh0[4], bl0[4];
            {
                const half8* Bh = reinterpret_cast<const half8*>(att + A_WQ1_FRAG) + (size_t)wave * 4 * 64 + lane;
                const half8* Bl = Bh + 8 * 4 * 64;
#pragma unroll
                for (int u = 0; u < 4; ++u) { bh0[u] = Bh[u * 64]; bl0[u] = Bl[u * 64]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            float z[32];
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 v = nld4(&qh[c][16 * u + 4 * q]);
                z[4 * u] = v.x; z[4 * u + 1] = v.y; z[4 * u + 2] = v.z; z[4 * u + 3] = v.w;
                sm += (v.x + v.y) + (v.z + v.w);
            }
            const float mean = nxrow_sum(sm) * (1.f / H);
            float var = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) { z[u] -= mean; var += z[u] * z[u]; }
            const float rstd = 1.f / sqrtf(nxrow_sum(var) * (1.f / H) + 1e-5f);
            half8 zh[4], zl[4];
            float zmx = 0.f;   // z >= 0 after the ReLU
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 g = nld4(&cst[0][16 * u + 4 * q]), b = nld4(&cst[1][16 * u + 4 * q]);
                z[4 * u + 0] = fmaxf(z[4 * u + 0] * rstd * g.x + b.x, 0.f); z[4 * u + 1] = fmaxf(z[4 * u + 1] * rstd * g.y + b.y, 0.f);
                z[4 * u + 2] = fmaxf(z[4 * u + 2] * rstd * g.z + b.z, 0.f); z[4 * u + 3] = fmaxf(z[4 * u + 3] * rstd * g.w + b.w, 0.f);
                zmx = fmaxf(fmaxf(zmx, fmaxf(z[4 * u], z[4 * u + 1])), fmaxf(z[4 * u + 2], z[4 * u + 3]));
            }
            float zinv, zrinv[4];
            const float zup = row_pow2(nxrow_max(zmx), zinv);
            rows_to_c_layout(zinv, q, zrinv);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v[8] = {z[8 * u], z[8 * u + 1], z[8 * u + 2], z[8 * u + 3], z[8 * u + 4], z[8 * u + 5], z[8 * u + 6], z[8 * u + 7]};
                split8(v, zup, zh[u], zl[u]);
            }
#pragma unroll
            for (int nt = wave; nt < 8; nt += NW) {      // NW = 4: two output tiles per wave
                // the tile's B operands: eight loads in flight together, requested once the 32 row values are split (requested first,
                // as until round 5, the compiler parked them in scratch across the LayerNorm: four reloads behind full waits)
                __builtin_amdgcn_sched_barrier(0);
                const half8* Bh = reinterpret_cast<const half8*>(att + A_WQ1_FRAG) + (size_t)nt * 4 * 64 + lane;   // [nt][u][lane]
                const half8* Bl = Bh + 8 * 4 * 64;
                half8 bh[4], bl[4];
#if CBGX_NS_P2_EARLY
                if (nt == wave) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { bh[u] = bh0[u]; bl[u] = bl0[u]; }
                } else
#endif
                {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { bh[u] = Bh[u * 64]; bl[u] = Bl[u * 64]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                const int col = 64 * (nt >> 2) + 4 * c + (nt & 3);
                const float b1 = cst[2][col], ci = cst[3][col];
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = MFMAH32(zh[u], bl[u], acc);
                    acc = MFMAH32(zl[u], bh[u], acc);
                    acc = MFMAH32(zh[u], bh[u], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = fmaf(acc[r] * zrinv[r], ci, b1);
                    qt[4 * q + r][col] = o;
                    int orw = orow[r];
                    asm volatile("" : "+v"(orw));
                    if (orw >= 0) qout[(size_t)orw * H + col] = o;
                }
            }
        }
        // ---- phase 3: the fold of head a = wave: Qt[row][a][m] = sum_cc q[row][8a+cc] Wbk[8a+cc][m] / sqrt 8 (exact fp32) ---
        // (skipped when no row of the tile wants it: the same answer in every wave, they all look at the same 16 rows); its B
        // operands do not depend on phase 2, so they are requested BEFORE the barrier that ends it and fly while the wave waits
        const bool do_fold = __ballot(fm != 0u) != 0ull;
        float4 b00 = {0.f, 0.f, 0.f, 0.f}, b01 = b00, b10 = b00, b11 = b00;
        if (do_fold) {
            const float* fb = att + A_WBK_FRAG + ((size_t)wave * 2 * 64 + lane) * 8;
            b00 = nld4(fb); b01 = nld4(fb + 4); b10 = nld4(fb + 64 * 8); b11 = nld4(fb + 64 * 8 + 4);
        }
        __syncthreads();
        if (do_fold) {
#pragma unroll
            for (int k = 0; k < HEADS / NW; ++k) {
                const int a = wave + k * NW;
                if (k) {      // NW = 8: the wave's second head
                    const float* fb = att + A_WBK_FRAG + ((size_t)a * 2 * 64 + lane) * 8;
                    b00 = nld4(fb); b01 = nld4(fb + 4); b10 = nld4(fb + 64 * 8); b11 = nld4(fb + 64 * 8 + 4);
                }
                const float2 qv = *reinterpret_cast<const float2*>(&qt[c][8 * a + 2 * q]);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 b0 = g ? b10 : b00, b1 = g ? b11 : b01;
                    floatx4 acc[4];
                    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
                    acc[0] = MFMA(qv.x, b0.x, zero); acc[1] = MFMA(qv.x, b0.z, zero);
                    acc[2] = MFMA(qv.x, b1.x, zero); acc[3] = MFMA(qv.x, b1.z, zero);
                    acc[0] = MFMA(qv.y, b0.y, acc[0]); acc[1] = MFMA(qv.y, b0.w, acc[1]);
                    acc[2] = MFMA(qv.y, b1.y, acc[2]); acc[3] = MFMA(qv.y, b1.w, acc[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if ((fm >> r) & 1u) {
                            const float4 o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                            int orw = orow[r];
                            asm volatile("" : "+v"(orw));
                            *reinterpret_cast<float4*>(Qt + ((size_t)orw * HEADS + a) * H + 64 * g + 4 * c) = o;
                        }
                    }
                }
            }
        }
        __syncthreads();   // the tiles are rewritten by the next row tile
    }
}

// ------------------------------------------------------------------------------------------------
// node_linear: C[M, nout] = act(A[M,128] @ Wt[128, nout] + bias), nout <= 128 (the classifier's two Linears,
// unitransformer.py:119-122).  Persistent 4-wave workgroups, Wt staged once in LDS ([128][nout padded to 16]), one
// wavefront per 16 rows, A in MFMA layout straight from global memory (k = 16u + 4q + j), scalar stores of 64-byte runs.
// ACT: 0 none, 1 softplus(x) - ln 2 (ShiftedSoftplus, repo/modules/common.py:174-180).
// ------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256) void node_linear_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ Wt, const float* __restrict__ bias,
                                                          float* __restrict__ C, int ldc, int M_all, int nout,
                                                          const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    __shared__ __attribute__((aligned(16))) float lds[H * H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    const int M = rows ? *n_rows_ptr : M_all;
    const int n_tiles = (M + 63) / 64;
    if ((int)blockIdx.x >= n_tiles) return;
    const int npad = (nout + 15) & ~15;
    // gridDim.y > 1 (few rows: the samplers' classifier on the ligand rows of a small batch): the 16-column tiles are dealt over
    // blockIdx.y, so a workgroup stages and multiplies only its own columns -- at one graph the first Linear was ONE wave per 16 rows
    // walking all eight column tiles behind a 64 KB fill, 27 - 46 us (profiles/step_timeline_r05[ab]_p1s1*.json)
    const int n_ct = npad / 16, ct_per = (n_ct + (int)gridDim.y - 1) / (int)gridDim.y;
    const int ct0 = (int)blockIdx.y * ct_per, ct1 = min(n_ct, ct0 + ct_per);
    if (ct0 >= ct1) return;
    const int wcols = 16 * (ct1 - ct0), col0 = 16 * ct0;
    // LDS fill [128][wcols] with every load of a pass in flight before the first store: the plain loop compiled to load -> wait ->
    // ds_write per element, 64 dependent L2 round trips for the [128][128] matrix
    if ((nout & 3) == 0 && col0 + wcols <= nout) {
        floatx4* dst = reinterpret_cast<floatx4*>(lds);
        const int w4 = wcols / 4, n4 = H * w4;         // <= 4096: 16 float4 per thread
        floatx4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (256 * u < n4) {          // (workgroup-uniform: a 16-column slice is two float4 per thread, the whole matrix sixteen)
                const int t = min(tid + 256 * u, n4 - 1), k = t / w4, c4 = t - k * w4;
                v[u] = *reinterpret_cast<const floatx4*>(Wt + (size_t)k * nout + col0 + 4 * c4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int t = tid + 256 * u; if (t < n4) dst[t] = v[u]; }
    } else {
        for (int t0 = 0; t0 < H * wcols; t0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + tid + 256 * u, k = t / wcols, col = col0 + (t - k * wcols);
                const float w = Wt[min(k, H - 1) * nout + min(col, nout - 1)];     // unconditional: a predicated load is a branch + wait
                v[u] = (t < H * wcols && col < nout) ? w : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + tid + 256 * u; if (t < H * wcols) lds[t] = v[u]; }
        }
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 64 + wave * 16;
        const int ak = min(row0 + c, M - 1);
        const int arow = rows ? rows[ak] : ak;
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = row0 + 4 * q + r;
            orow[r] = k < M ? (rows ? rows[k] : k) : -1;
        }
        float a[32];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = nld4(A + (size_t)arow * lda + 16 * u + 4 * q);
            a[4 * u] = v.x; a[4 * u + 1] = v.y; a[4 * u + 2] = v.z; a[4 * u + 3] = v.w;
        }
        for (int ct = ct0; ct < ct1; ++ct) {
            const int col = 16 * ct + c;
            const float b = (bias && col < nout) ? bias[min(col, nout - 1)] : 0.f;
            floatx4 acc0 = {b, b, b, b}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* wp = lds + (16 * u + 4 * q) * wcols + (col - col0);
                acc0 = MFMA(a[4 * u + 0], wp[0], acc0);
                acc1 = MFMA(a[4 * u + 1], wp[wcols], acc1);
                acc0 = MFMA(a[4 * u + 2], wp[2 * wcols], acc0);
                acc1 = MFMA(a[4 * u + 3], wp[3 * wcols], acc1);
            }
            if (col < nout) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (orow[r] >= 0) {
                        float v = acc0[r] + acc1[r];
                        if (ACT == 1) v = (v > 20.f ? v : log1pf(expf(v))) - 0.69314718055994530942f;
                        C[(size_t)orow[r] * ldc + col] = v;
                    }
                }
            }
        }
    }
}

hipError_t launch_node_gemm(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M,
                            int nout, int act, hipStream_t s, const int* rows, const int* n_rows) {
    if (M == 0) return hipSuccess;
    if (nout < 1 || nout > H) return hipErrorInvalidValue;
    const int tiles = (M + 63) / 64;
    // few row tiles: one workgroup per (row tile, 16-column tile) -- see the kernel
    const dim3 grid(tiles < 512 ? tiles : 512, tiles <= 128 ? (nout + 15) / 16 : 1), block(256);
    profile_mark_begin(K_NODE_GEMM, s);
    if (act == 0)
        hipLaunchKernelGGL(node_linear_kernel<0>, grid, block, 0, s, A, lda, Wt, bias, C, ldc, M, nout, rows, n_rows);
    else
        hipLaunchKernelGGL(node_linear_kernel<1>, grid, block, 0, s, A, lda, Wt, bias, C, ldc, M, nout, rows, n_rows);
    profile_mark_end(s);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fragment packing (from reference-layout tensors)
// ------------------------------------------------------------------------------------------------
// node projection: split-f16 chunk tables, dst[ch][part][ct][u][lane][j] (f16) = hi / lo of
// Wcat[col = 64ch + 4c + ct][k = 16 (2u + (j >> 2)) + 4q + (j & 3)]; Wcat rows are assembled from W_a_k / W_a_v (dst and src
// thirds) and W_q0, exactly like the K-major A_WN table.
// weight of the assembled node projection Wcat[col][k] (col = PDk | PDv | PSk | PSv | q hidden)
__device__ __forceinline__ float nproj_weight(const PackBlocks& pb, int block, int col, int k) {
    const int grp = col >> 7, n = col & 127;      // PDk, PDv: destination third of W_a; PSk, PSv: source third; q hidden: wq0
    const int ld = grp < 4 ? KV_IN : H;
    const int off = grp < 2 ? NT + NT * G : (grp < 4 ? NT + NT * G + H : 0);
    return pb.nsrc[block][grp][(size_t)n * ld + off + k];
}
// 2^-kc for a column whose largest |w| is mx: w 2^kc lands in [2^14, 2^15) (kc clamped so that both factors are normal)
__device__ __forceinline__ float col_pow2_inv(float mx) {
    const int E = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    const int kc = max(-100, min(60, 141 - E));
    return __uint_as_float((unsigned)(127 - kc) << 23);
}
// per-column inverse scales of the two split-f16 node tables: cinv[0..640) node projection, cinv[640..768) query MLP second Linear
// ---- one launch for all attention blocks (blockIdx.y = block; kernels.h PackBlocks); wk0 / wv0 below are the CENTRED first Linears
// one wavefront per column (the 128 weights of a column are contiguous in every source tensor: two coalesced loads per lane)
__global__ __launch_bounds__(256) void pack_colscale_kernel(PackBlocks pb) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= PROW + H) return;
    float* att = pb.att[blockIdx.y];
    const float* wq1 = pb.wq1[blockIdx.y];
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = lane + 64 * u;
        mx = fmaxf(mx, fabsf(col < PROW ? nproj_weight(pb, blockIdx.y, col, k) : wq1[(size_t)(col - PROW) * H + k]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) { if (col < PROW) att[A_NPROJ_CINV + col] = col_pow2_inv(mx); else att[A_WQ1_CINV + col - PROW] = col_pow2_inv(mx); }
}

// node projection: split-f16 chunk tables, dst[ch][part][ct][u][lane][j] (f16) = hi / lo of
// Wcat[col = 64ch + 4c + ct][k = 16 (2u + (j >> 2)) + 4q + (j & 3)] 2^kc; Wcat rows are assembled from W_a_k / W_a_v (dst and src
// thirds) and W_q0, exactly like the K-major A_WN table.
__global__ void pack_nproj_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per weight: ((ch*4 + ct)*4 + u)*64*8 + lane*8 + j
    if (idx >= NP_CHUNKS * 4 * 4 * 64 * 8) return;
    float* att = pb.att[blockIdx.y];
    const int j = idx & 7, lane = (idx >> 3) & 63, u = (idx >> 9) & 3, ct = (idx >> 11) & 3, ch = idx >> 13;
    const int c = lane & 15, q = lane >> 4;
    const int col = 64 * ch + 4 * c + ct, k = 16 * (2 * u + (j >> 2)) + 4 * q + (j & 3);
    const float v = nproj_weight(pb, blockIdx.y, col, k) * (1.f / att[A_NPROJ_CINV + col]);   // exact: a power of two
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    _Float16* chunk = reinterpret_cast<_Float16*>(att + A_NPROJ_FRAG) + (size_t)ch * (NP_CHUNK * 2);   // f16 elements per chunk
    const size_t off = ((size_t)(ct * 4 + u) * 64 + lane) * 8 + j;
    chunk[off] = hi;
    chunk[(size_t)4 * 4 * 64 * 8 + off] = lo;
}

__global__ void pack_wq1_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per weight: [nt 8][u 4][lane 64][j 8]
    if (idx >= H * H) return;
    float* att = pb.att[blockIdx.y];
    const int j = idx & 7, lane = (idx >> 3) & 63, u = (idx >> 9) & 3, nt = idx >> 11;
    const int c = lane & 15, q = lane >> 4;
    const int n = 64 * (nt >> 2) + 4 * c + (nt & 3);
    const float v = pb.wq1[blockIdx.y][(size_t)n * H + 16 * (2 * u + (j >> 2)) + 4 * q + (j & 3)] * (1.f / att[A_WQ1_CINV + n]);
    const _Float16 hi = (_Float16)v;
    _Float16* dst = reinterpret_cast<_Float16*>(att + A_WQ1_FRAG);
    dst[idx] = hi;
    dst[(size_t)H * H + idx] = (_Float16)(v - (float)hi);
}

__global__ void pack_wbk_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // [a][g][lane][j*2+step]
    if (idx >= NF_FRAG) return;
    const int step = idx & 1, j = (idx >> 1) & 3, lane = (idx >> 3) & 63, g = (idx >> 9) & 1, a = idx >> 10;
    const int c = lane & 15, q = lane >> 4;
    pb.att[blockIdx.y][A_WBK_FRAG + idx] = pb.wk1[blockIdx.y][(size_t)(8 * a + 2 * q + step) * H + 64 * g + 4 * c + j] * 0.35355339059327376220f;
}

// bn2[li][col]: PDk | PDv get the centred bias plus the type column of a protein source for destination class li
__global__ void pack_bn2_kernel(PackBlocks pb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * PROW) return;
    float* att = pb.att[blockIdx.y];
    const int li = idx / PROW, col = idx % PROW, blk = col >> 7, n = col & 127;
    const int tp = li ? 2 : 3;  // type(src prot, dst lig) = 2, type(src prot, dst prot) = 3
    float v = 0.f;
    if (blk == 0) v = att[A_BAKC + n] + att[A_WAKC + (size_t)n * KV_IN + tp];
    else if (blk == 1) v = att[A_BAVC + n] + att[A_WAVC + (size_t)n * KV_IN + tp];
    else if (blk == 4) v = pb.bq0[blockIdx.y][n];
    att[A_BN2 + idx] = v;
}

hipError_t launch_pack_node_tables(const PackBlocks& pb, hipStream_t s) {
    hipLaunchKernelGGL(pack_colscale_kernel, dim3((PROW + H + 3) / 4, pb.n), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_nproj_kernel, dim3(NP_CHUNKS * 4 * 4 * 64 * 8 / 256, pb.n), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_wq1_kernel, dim3(H * H / 256, pb.n), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_wbk_kernel, dim3(NF_FRAG / 256, pb.n), dim3(256), 0, s, pb);
    hipLaunchKernelGGL(pack_bn2_kernel, dim3((2 * PROW + 255) / 256, pb.n), dim3(256), 0, s, pb);
    return hipGetLastError();
}

// compact the indices of flagged nodes (gen_flag) -> list[0 .. *count); order is irrelevant (each node is computed
// independently, so results are identical for any order).  *count must be zero on entry.
// One returning atomic per 1024-thread workgroup, not per wave: a single counter word serves ~88 returning atomics per microsecond
// (MI355X_MICROARCH.md, dequeue row), so 1 556 wave-level atomics for a 99.5 k-node batch were the kernel's whole 18 us.
constexpr int BA_THREADS = 1024;
__global__ __launch_bounds__(BA_THREADS) void build_active_kernel(const uint8_t* __restrict__ flag, int n, int* __restrict__ list,
                                                                  int* __restrict__ count) {
    __shared__ int s_cnt[BA_THREADS / 64];
    __shared__ int s_base;
    const int idx = blockIdx.x * BA_THREADS + threadIdx.x;
    const bool a = idx < n && flag[idx] != 0;
    const unsigned long long m = __ballot(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < BA_THREADS / 64; ++w) { const int c = s_cnt[w]; s_cnt[w] = tot; tot += c; }   // exclusive prefix
        s_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (a) list[s_base + s_cnt[wave] + __popcll(m & ((1ull << lane) - 1ull))] = idx;
}

// split a work list (or, without one, the nodes 0 .. n-1) by a per-node flag: flagged nodes -> out1, the others -> out0, each with its
// own device-side count (both zero on entry).  Same aggregation as build_active_kernel: one returning atomic per list and workgroup.
// Used for the (general, protein-only) list pair of every x2h layer (edge_mfma.hip, edge_x2h_dual_kernel).
__global__ __launch_bounds__(BA_THREADS) void split_list_kernel(const int* __restrict__ list, const int* __restrict__ count, int n,
                                                                const uint8_t* __restrict__ flag, int* __restrict__ out1,
                                                                int* __restrict__ cnt1, int* __restrict__ out0,
                                                                int* __restrict__ cnt0) {
    __shared__ int s_cnt[2][BA_THREADS / 64];
    __shared__ int s_base[2];
    const int idx = blockIdx.x * BA_THREADS + threadIdx.x;
    const int n_items = list ? *count : n;
    const bool valid = idx < n_items;
    const int node = valid ? (list ? list[idx] : idx) : 0;
    const bool f = valid && flag[node] != 0;
    const unsigned long long m1 = __ballot(f), m0 = __ballot(valid && !f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_cnt[1][wave] = __popcll(m1); s_cnt[0][wave] = __popcll(m0); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        int tot = 0;
#pragma unroll
        for (int w = 0; w < BA_THREADS / 64; ++w) { const int c = s_cnt[k][w]; s_cnt[k][w] = tot; tot += c; }   // exclusive prefix
        s_base[k] = tot ? atomicAdd(k ? cnt1 : cnt0, tot) : 0;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (f) out1[s_base[1] + s_cnt[1][wave] + __popcll(m1 & below)] = node;
    else if (valid) out0[s_base[0] + s_cnt[0][wave] + __popcll(m0 & below)] = node;
}

hipError_t launch_split_list(const int* list, const int* count, int n, const uint8_t* flag, int* out1, int* cnt1, int* out0,
                             int* cnt0, hipStream_t s, bool counters_zeroed) {
    if (!counters_zeroed) {   // the two counters live in one 128-byte region (cnt1 first): one memset
        hipError_t e = hipMemsetAsync(cnt1, 0, (size_t)((char*)cnt0 - (char*)cnt1) + sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(split_list_kernel, dim3((n + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, s, list, count, n, flag,
                       out1, cnt1, out0, cnt0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Every node list of one cbgx_unitransformer_forward call in FOUR launches (round 4; they were ~21: a mark kernel and a
// compaction with its own counter fill per list -- 100 us of a 1-graph step that is ~1 ms of dependent ~5 us launches).
// All flag arrays and counters are zero on entry (ONE fill of the workspace region that holds them).  Three level kernels over
// (node, neighbour slot) pairs -- a level only reads flags the previous level has completed --
//   level 0:  a1 = gen | lig | nbr(gen)              d1 = lig | {i : nbr(i) has a ligand atom}
//   level 1:  a2 = a1 | nbr(a1)     D2 = D1 | {i : nbr(i) meets D1}     S1 = D1 | nbr(D1)
//   level 2:  a3 = a2 | nbr(a2)     S2 = D2 | nbr(D2)
// (nbr(set) = the in-neighbours of its members = the sources their edge kernels read; D1 = the caller's "differs from the
// ligand-free pocket" flags: the proximity flags of the graph cache, or d1 itself) and one compaction launch, blockIdx.y = job.
// List order is irrelevant: every node is computed independently of its position in a list.
// ------------------------------------------------------------------------------------------------
__global__ void list_level_kernel(GraphFlags f, const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg, int n, int level,
                                  int cached, int prune) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t >> 5), e = (int)(t & 31);
    if (i >= n) return;
    const bool valid = e < deg[i];
    const int j = valid ? nbr[(size_t)i * KNN + e] : i;
    if (level == 0) {
        const bool g = f.gen[i] != 0, l = f.lig[i] != 0;
        if (e == 0 && (g || l)) f.a1[i] = 1;
        if (g && valid) f.a1[j] = 1;
        if ((e == 0 && l) || (valid && f.lig[j])) f.d1[i] = 1;
    } else if (level == 1) {
        if (prune) {
            const bool a = f.a1[i] != 0;
            if (e == 0 && a) f.a2[i] = 1;
            if (a && valid) f.a2[j] = 1;
        }
        if (cached) {
            const bool d = f.D1[i] != 0;
            if ((e == 0 && d) || (valid && f.D1[j])) f.D2[i] = 1;
            if (e == 0 && d) f.S1[i] = 1;
            if (d && valid) f.S1[j] = 1;
        }
    } else {
        if (prune) {
            const bool a = f.a2[i] != 0;
            if (e == 0 && a) f.a3[i] = 1;
            if (a && valid) f.a3[j] = 1;
        }
        if (cached) {
            const bool d = f.D2[i] != 0;
            if (e == 0 && d) f.S2[i] = 1;
            if (d && valid) f.S2[j] = 1;
        }
    }
}

__global__ __launch_bounds__(BA_THREADS) void build_lists_kernel(ListJobs jobs, int n) {
    __shared__ int s_cnt[BA_THREADS / 64];
    __shared__ int s_base;
    const int job = blockIdx.y;
    const uint8_t* __restrict__ fa = jobs.flag[job];
    const uint8_t* __restrict__ fb = jobs.flag2[job];
    const int idx = blockIdx.x * BA_THREADS + threadIdx.x;
    bool a = idx < n && (fa == nullptr || fa[idx] != 0);
    if (a && fb) a = (fb[idx] != 0) == (jobs.want2[job] != 0);
    const unsigned long long m = __ballot(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < BA_THREADS / 64; ++w) { const int c = s_cnt[w]; s_cnt[w] = tot; tot += c; }   // exclusive prefix
        s_base = tot ? atomicAdd(jobs.count[job], tot) : 0;
    }
    __syncthreads();
    if (a) jobs.list[job][s_base + s_cnt[wave] + __popcll(m & ((1ull << lane) - 1ull))] = idx;
}

// ---- the same four steps for a small input in one launch: one 1024-thread workgroup per graph, flags in LDS (kernels.h) ----------
// One 32-bit word per node holds all its flags (bit = GraphFlagId).  A level visits every (node i, slot e) pair once: it reads the
// words of i and of the neighbour j, ORs what j contributes to i over the node's 32 lanes with a ballot (one atomic OR per node, by
// the lane of slot 0) and ORs what i contributes to j into j's word (lanes with nothing to contribute are masked, waves without any
// skip the instruction).  The first two versions kept one byte per flag and stored per mark -- ~8 LDS instructions per pair and
// level on the ONE compute unit that holds a one-graph batch: 24 - 27 us, more than the four launches this kernel replaced
// (profiles/step_timeline_r05[c-e]_p1s1.json); the words need two reads and at most two sparse atomics.
__global__ __launch_bounds__(1024) void graph_lists_kernel(const uint8_t* __restrict__ gen, const uint8_t* __restrict__ lig,
                                                           const uint8_t* __restrict__ d1_in, const int32_t* __restrict__ nbr,
                                                           const int32_t* __restrict__ deg, const int32_t* __restrict__ graph_ptr,
                                                           GraphListJobs jobs, int cached, int prune,
                                                           uint8_t* __restrict__ d1_out) {
    __shared__ unsigned F[GRAPH_LISTS_MAX_NODES];
    __shared__ int s_cnt[LIST_JOBS_MAX][16];
    __shared__ int s_base[LIST_JOBS_MAX];
    const int gs = graph_ptr[blockIdx.x], n = graph_ptr[blockIdx.x + 1] - gs;
    if (n <= 0 || n > GRAPH_LISTS_MAX_NODES) return;     // (the launcher only takes inputs whose total is within the bound)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (the three flag bytes of a node are requested together -- without caller flags the third load reads `lig` again and is
    // ignored -- and BEHIND the neighbour entries below: one round trip for everything the levels need)
    const uint8_t* __restrict__ d1p = d1_in ? d1_in : lig;
    // The (node, slot) pairs of a thread -- pair t = tid + 1024 u, so a wave holds the 2 x 32 slots of two nodes -- are loaded ONCE,
    // all in flight, and kept in registers over the three levels when the graph has at most GL_KEEP * 32 nodes (768: every pocket
    // of the shipped data).  Rows are -1 padded past the node's degree (knn_graph writes them so; cbgx.h states it for a caller's
    // static_nbr): a slot is valid iff its entry is a node of this graph -- the range check also keeps a foreign table from indexing
    // LDS out of bounds (`deg` itself is not read: the check costs two VALU per pair, a load would cost a round trip).
    constexpr int GL_KEEP = 24;
    const int n_pairs = n * KNN;
    const bool keep = n_pairs <= GL_KEEP * 1024;
    int jl[GL_KEEP];
    if (keep) {
#pragma unroll
        for (int u = 0; u < GL_KEEP; ++u) {
            const int t = tid + 1024 * u;
            jl[u] = nbr[(size_t)gs * KNN + min(t, n_pairs - 1)];
        }
    }
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = min(k0 + tid, n - 1);
        const unsigned fg = gen[gs + k], fl = lig[gs + k], fd = d1p[gs + k];
        if (k0 + tid < n)
            F[k] = (fg ? 1u << GF_GEN : 0u) | (fl ? 1u << GF_LIG : 0u) | ((d1_in != nullptr && fd) ? 1u << GF_D1IN : 0u);
    }
    __syncthreads();
    const unsigned Dx = d1_in ? GF_D1IN : GF_d1;     // "differs from the ligand-free pocket": the caller's proximity flags, or d1 itself
    const int levels = (cached || prune) ? 3 : 1;
    const bool hi = lane >= 32;                       // which of the wave's two nodes this lane belongs to
    const bool e0 = (lane & 31) == 0;
    for (int level = 0; level < levels; ++level) {
        for (int u0 = 0; u0 * 1024 < n_pairs; u0 += GL_KEEP) {
            if (!keep) {
#pragma unroll
                for (int u = 0; u < GL_KEEP; ++u) {
                    const int t = tid + 1024 * (u0 + u);
                    jl[u] = nbr[(size_t)gs * KNN + min(t, n_pairs - 1)];
                }
            }
            // every flag word a level READS was completed by the level before it (a level only sets bits that the next one reads),
            // so a thread's reads go out twelve pairs at a time, ahead of those pairs' atomics: two LDS round trips per level
            // instead of one per pair (all 24 at once do not fit the 128 registers of a 1024-thread workgroup)
            constexpr int GL_BATCH = 12;
#pragma unroll
            for (int ub = 0; ub < GL_KEEP; ub += GL_BATCH) {
            if ((u0 + ub) * 1024 + (tid & ~63) >= n_pairs) break;      // wave-uniform
            unsigned fi_[GL_BATCH], fj_[GL_BATCH];
#pragma unroll
            for (int v = 0; v < GL_BATCH; ++v) {
                const int u = ub + v;
                const int t = tid + 1024 * (u0 + u);
                const int i = min(t, n_pairs - 1) >> 5;
                const bool valid = t < n_pairs && (unsigned)(jl[u] - gs) < (unsigned)n;     // -1 padding, and ids outside the graph
                fi_[v] = F[i];
                fj_[v] = F[valid ? jl[u] - gs : i];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < GL_BATCH; ++v) {
                const int u = ub + v;
                if ((u0 + u) * 1024 + (tid & ~63) >= n_pairs) break;      // wave-uniform: the wave's 64 pairs are past the end
                const int t = tid + 1024 * (u0 + u);                      // (n_pairs is a multiple of 32: a half-wave is all in or all out)
                const bool in = t < n_pairs;
                const int i = min(t, n_pairs - 1) >> 5;
                const bool valid = in && (unsigned)(jl[u] - gs) < (unsigned)n;
                const int j = valid ? jl[u] - gs : i;
                const unsigned fi = fi_[v], fj = fj_[v];
                // what the neighbour tells the node (OR over the node's slots) and what the node tells the neighbour
                unsigned from_j, to_j, self;
                if (level == 0) {
                    from_j = (valid && (fj >> GF_LIG & 1u)) ? 1u << GF_d1 : 0u;
                    self = ((fi >> GF_GEN | fi >> GF_LIG) & 1u) << GF_a1 | (fi >> GF_LIG & 1u) << GF_d1;
                    to_j = (valid && (fi >> GF_GEN & 1u)) ? 1u << GF_a1 : 0u;
                } else if (level == 1) {
                    const unsigned a = prune ? fi >> GF_a1 & 1u : 0u, d = cached ? fi >> Dx & 1u : 0u;
                    from_j = (cached && valid && (fj >> Dx & 1u)) ? 1u << GF_D2 : 0u;
                    self = a << GF_a2 | d << GF_D2 | d << GF_S1;
                    to_j = valid ? (a << GF_a2 | d << GF_S1) : 0u;
                } else {
                    const unsigned a = prune ? fi >> GF_a2 & 1u : 0u, d = cached ? fi >> GF_D2 & 1u : 0u;
                    from_j = 0u;
                    self = a << GF_a3 | d << GF_S2;
                    to_j = valid ? (a << GF_a3 | d << GF_S2) : 0u;
                }
                // OR of from_j over the node's 32 lanes: every flag it can carry in this level is one bit, so a ballot per level does
                const unsigned long long bal = __ballot(from_j != 0u);
                const bool any = ((hi ? bal >> 32 : bal) & 0xffffffffull) != 0ull;
                const unsigned own = in ? (self | (any ? (level == 0 ? 1u << GF_d1 : 1u << GF_D2) : 0u)) : 0u;
                if (e0 && own) atomicOr(&F[i], own);
                if (__ballot(to_j != 0u) != 0ull) {
                    if (to_j) atomicOr(&F[j], to_j);
                }
            }
            }
        }
        __syncthreads();
    }
    if (d1_out)
        for (int k = tid; k < n; k += 1024) d1_out[gs + k] = (uint8_t)(F[k] >> GF_d1 & 1u);
    // compaction of every job, 1024 nodes of the graph at a time: per job one returning atomic per pass (list order is irrelevant)
    for (int base = 0; base < n; base += 1024) {
        const int k = base + tid;
        const unsigned fk = k < n ? F[k] : 0u;
        unsigned member = 0u;       // bit job
        for (int job = 0; job < jobs.n_jobs; ++job) {
            const int fa = jobs.flag[job], fb = jobs.flag2[job];
            bool a = k < n && (fa == GF_ALL || (fk >> (fa == GF_D1IN ? Dx : (unsigned)fa) & 1u));
            if (a && fb != GF_ALL) a = ((fk >> (fb == GF_D1IN ? Dx : (unsigned)fb) & 1u) != 0u) == (jobs.want2[job] != 0);
            const unsigned long long m = __ballot(a);
            if (lane == 0) s_cnt[job][wave] = __popcll(m);
            member |= a ? 1u << job : 0u;
        }
        __syncthreads();
        if (tid < jobs.n_jobs) {
            int tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const int c = s_cnt[tid][w]; s_cnt[tid][w] = tot; tot += c; }   // exclusive prefix
            s_base[tid] = tot ? atomicAdd(jobs.count[tid], tot) : 0;
        }
        __syncthreads();
        for (int job = 0; job < jobs.n_jobs; ++job) {
            const bool a = (member >> job) & 1u;
            const unsigned long long m = __ballot(a);
            if (a) jobs.list[job][s_base[job] + s_cnt[job][wave] + __popcll(m & ((1ull << lane) - 1ull))] = gs + k;
        }
        __syncthreads();
    }
}

hipError_t launch_graph_lists(const uint8_t* gen, const uint8_t* lig, const uint8_t* d1_in, const int32_t* nbr, const int32_t* deg,
                              const int32_t* graph_ptr, int n_graphs, const GraphListJobs& jobs, bool cached, bool prune,
                              uint8_t* d1_out, hipStream_t s) {
    if (n_graphs == 0 || jobs.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(graph_lists_kernel, dim3(n_graphs), dim3(1024), 0, s, gen, lig, d1_in, nbr, deg, graph_ptr, jobs,
                       cached ? 1 : 0, prune ? 1 : 0, d1_out);
    return hipGetLastError();
}

hipError_t launch_list_level(const GraphFlags& f, const int32_t* nbr, const int32_t* deg, int n, int level, bool cached, bool prune,
                             hipStream_t s) {
    if (n == 0) return hipSuccess;
    const long threads = (long)n * 32;
    hipLaunchKernelGGL(list_level_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, f, nbr, deg, n, level,
                       cached ? 1 : 0, prune ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_build_lists(const ListJobs& jobs, int n, hipStream_t s) {
    if (n == 0 || jobs.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(build_lists_kernel, dim3((n + BA_THREADS - 1) / BA_THREADS, jobs.n_jobs), dim3(BA_THREADS), 0, s, jobs, n);
    return hipGetLastError();
}

// ---- receptive-field pruning helpers ---------------------------------------------------------------
__global__ void mark_seed_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int n,
                                 uint8_t* __restrict__ m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m[i] = (a[i] | b[i]) ? 1 : 0;
}

// m[j] = 1 for every in-neighbour j of the listed nodes (benign write race: every writer stores 1)
__global__ void mark_nbr_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg,
                                uint8_t* __restrict__ m) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t >> 5, e = t & 31;
    if (k >= *count) return;
    const int i = list[k];
    if (e < deg[i]) m[nbr[(size_t)i * KNN + e]] = 1;
}

// out[i] |= 1 if any in-neighbour of i is flagged (forward propagation of "differs from the ligand-free pocket");
// out must already hold the flags themselves.  One thread per (node, slot); benign write race (every writer stores 1).
__global__ void mark_from_nbr_kernel(const uint8_t* __restrict__ flag, const int32_t* __restrict__ nbr,
                                     const int32_t* __restrict__ deg, int n, uint8_t* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t >> 5), e = (int)(t & 31);
    if (i >= n || e >= deg[i]) return;
    if (flag[nbr[(size_t)i * KNN + e]]) out[i] = 1;
}

hipError_t launch_mark_from_nbr(const uint8_t* flag, const int32_t* nbr, const int32_t* deg, int n, uint8_t* out,
                                hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(out, flag, (size_t)n, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    const long threads = (long)n * 32;
    hipLaunchKernelGGL(mark_from_nbr_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, flag, nbr, deg, n, out);
    return hipGetLastError();
}

hipError_t launch_mark_seed(const uint8_t* a, const uint8_t* b, int n, uint8_t* m, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(mark_seed_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, m);
    return hipGetLastError();
}

hipError_t launch_mark_nbr(const int* list, const int* count, int n_upper, const int32_t* nbr, const int32_t* deg,
                           uint8_t* m, hipStream_t s) {
    if (n_upper == 0) return hipSuccess;
    const long threads = (long)n_upper * 32;
    hipLaunchKernelGGL(mark_nbr_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, list, count, nbr, deg, m);
    return hipGetLastError();
}

hipError_t launch_build_active(const uint8_t* flag, int n, int* list, int* count, hipStream_t s, bool counter_zeroed) {
    if (!counter_zeroed) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(build_active_kernel, dim3((n + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, s, flag, n, list, count);
    return hipGetLastError();
}

constexpr unsigned CHUNKS_ALL = 0x3ffu;   // PDk PDv PSk PSv qh
constexpr unsigned CHUNKS_PS = 0x0f0u;    // PSk | PSv  (columns 256..511): what *neighbours* contribute
constexpr unsigned CHUNKS_OWN = 0x30fu;   // PDk | PDv | qh: what the destination node itself contributes

// node stage of one attention block: P, q (scratch), Qt.
// `act` / `act_count` (optional): only the listed destination nodes will be processed by the edge kernel (h2x: nodes
// that can move; x2h in the last layers: nodes whose features can still reach an output).  `src` / `src_count`
// (optional): the nodes that can be *sources* of those destinations; without it PS is produced for every node.

bool add_node_stage_jobs(NodeStageJobs& jobs, const float* att, float* P, float* qbuf, float* Qt, const int* act,
                         const int* act_count, const int* src, const int* src_count, const uint8_t* fold_flag) {
    if (jobs.n + (act ? 2 : 1) > NS_JOBS_MAX) return false;
    jobs.j[jobs.n++] = NodeStageJob{att, P, qbuf, Qt, act, act_count, act ? CHUNKS_OWN : CHUNKS_ALL, 0, fold_flag};
    if (act) jobs.j[jobs.n++] = NodeStageJob{att, P, qbuf, Qt, src, src_count, CHUNKS_PS, 1, nullptr};
    return true;
}

// Workgroup size.  Measured (profiles/small_r05k.log): the 8-wave variant (two workgroups per CU) wins at every input size -- at
// one graph (28 row tiles) 15.6 against 16.8 us per launch, at ten graphs (276 tiles, ~440 busy (tile, job) pairs: more than the 256
// CUs a 16-wave workgroup needs one each of) 22.4 against 26.1.  CBGX_NODE_STAGE_WAVES = 4 | 8 | 16 forces a variant (A/B runs and
// the bit-identity test).
static void launch_node_stage_grid(const NodeStageJobs& jobs, const float* h, const uint8_t* lig, int n_nodes, hipStream_t s) {
    static const int forced = [] { const char* e = getenv("CBGX_NODE_STAGE_WAVES"); return e ? atoi(e) : 0; }();
    const int tiles = (n_nodes + 15) / 16;
    const int nw = (forced == 4 || forced == 8 || forced == 16) ? forced : NODE_STAGE_WAVES;
    const dim3 grid(min(tiles, 1024), jobs.n);
    if (nw == 4) hipLaunchKernelGGL(node_stage_kernel<4>, grid, dim3(256), 0, s, jobs, h, lig, n_nodes);
    else if (nw == 8) hipLaunchKernelGGL(node_stage_kernel<8>, grid, dim3(512), 0, s, jobs, h, lig, n_nodes);
    else hipLaunchKernelGGL(node_stage_kernel<16>, grid, dim3(1024), 0, s, jobs, h, lig, n_nodes);
}

hipError_t launch_node_stage_jobs(const NodeStageJobs& jobs, const float* h, const uint8_t* lig, int n_nodes, hipStream_t s) {
    if (n_nodes == 0 || jobs.n == 0) return hipSuccess;
    if (n_nodes > NODE_STAGE_MAX_ROWS || jobs.n > NS_JOBS_MAX) return hipErrorInvalidValue;
    profile_mark_begin(K_NODE_QUERY, s);
    launch_node_stage_grid(jobs, h, lig, n_nodes, s);
    profile_mark_end(s);
    return hipGetLastError();
}

// `fold` / `fold_count` (optional, x2h blocks of the inference path): the rows whose folded query Qt the edge stage will read -- the
// GENERAL list of the layer; protein-only destinations fold in registers (edge_mfma.hip) and only need q.  Without it Qt is
// produced for every destination.
hipError_t launch_node_mfma(const float* att, const float* h, const uint8_t* lig, int n_nodes, float* P, float* qbuf,
                            float* Qt, const int* act, const int* act_count, const int* src, const int* src_count,
                            hipStream_t s, bool large_lists, const int* fold, const int* fold_count) {
    if (n_nodes == 0) return hipSuccess;
    const int tiles = (n_nodes + 63) / 64;
    const int grid = min(tiles, 512);
    // few row tiles (small batches, or a work list): spread the heads of the fold / the halves of the query MLP over workgroups too
    const bool small = tiles <= 128 || (act != nullptr && !large_lists);
    // node_proj_kernel: one workgroup per (row-tile column x, selected chunk); ALL of them resident at once -- the kernel's
    // __launch_bounds__(256, 3) give three 4-wave workgroups per CU = 768 slots (a grid of 960 ran as two rounds, the second a
    // quarter full: 100 us instead of 67) -- and gridDim.x a multiple of 8 so that the workgroups of one row tile -- linear ids
    // x + y gridDim.x -- land on one XCD and its L2 serves the tile's rows to all of them
    auto proj_grid = [&](unsigned mask) {
        const int py = (__builtin_popcount(mask) + NP_CPW - 1) / NP_CPW;
        int gx = NP_RESIDENT_WGS / py;
        if (gx >= 8) gx &= ~7;
        return dim3((unsigned)min(tiles, gx), (unsigned)py);
    };
    // Small inputs (n_nodes bounds a work list's length too): ONE launch does projection (own columns, or all of them when
    // every node is a destination), query MLP and query fold -- such launches are bound by kernel boundaries, not throughput.
    // (A short destination list on a LARGE input -- the h2x blocks, ~5 k movable atoms of a 99.5 k-node batch -- stays on the
    // three-kernel chain: the fused kernel was measured there in round 4, 40 us against 10 + 8 + 12: its 16-wave workgroups, one
    // per CU, need two rounds for 313 row tiles.)
    const bool fused = n_nodes <= NODE_STAGE_MAX_ROWS;
    profile_mark_begin(K_NODE_GEMM, s);
    // fused with a destination list: the source rows' PS columns are the second job of the node_stage_kernel launch below
    const bool two_jobs = fused && act != nullptr;
    if (!act) {
        if (!fused)
            hipLaunchKernelGGL(node_proj_kernel<false>, proj_grid(CHUNKS_ALL), dim3(256), 0, s, att, h, lig, P, n_nodes,
                               (const int*)nullptr, (const int*)nullptr, CHUNKS_ALL);
    } else {
        if (!two_jobs) {     // without a source list the PS columns are produced for every node
            if (src)
                hipLaunchKernelGGL(node_proj_kernel<true>, proj_grid(CHUNKS_PS), dim3(256), 0, s, att, h, lig, P, n_nodes, src,
                                   src_count, CHUNKS_PS);
            else
                hipLaunchKernelGGL(node_proj_kernel<false>, proj_grid(CHUNKS_PS), dim3(256), 0, s, att, h, lig, P, n_nodes,
                                   (const int*)nullptr, (const int*)nullptr, CHUNKS_PS);
        }
        if (!fused)
            hipLaunchKernelGGL(node_proj_kernel<true>, proj_grid(CHUNKS_OWN), dim3(256), 0, s, att, h, lig, P, n_nodes, act, act_count,
                               CHUNKS_OWN);
    }
    profile_mark_end(s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    profile_mark_begin(K_NODE_QUERY, s);
    if (fused) {
        NodeStageJobs jobs;
        jobs.n = 0;
        add_node_stage_jobs(jobs, att, P, qbuf, Qt, act, act_count, src, src_count);
        launch_node_stage_grid(jobs, h, lig, n_nodes, s);
    } else {
        hipLaunchKernelGGL(node_qmlp_kernel, dim3(grid, small ? 2 : 1), dim3(256), 0, s, att, P, qbuf, n_nodes, act, act_count);
        if (fold)    // heads spread over four workgroups per row tile: the list is a fraction of the nodes, of unknown length
            hipLaunchKernelGGL(node_qfold_kernel<4>, dim3(grid, 4), dim3(256), 0, s, att, qbuf, Qt, n_nodes, fold, fold_count);
        else if (small)
            hipLaunchKernelGGL(node_qfold_kernel<4>, dim3(grid, 4), dim3(256), 0, s, att, qbuf, Qt, n_nodes, act, act_count);
        else
            hipLaunchKernelGGL(node_qfold_kernel<16>, dim3(grid, 1), dim3(256), 0, s, att, qbuf, Qt, n_nodes, act, act_count);
    }
    profile_mark_end(s);
    return hipGetLastError();
}

}  // namespace cbgx

// libcbgx -- node-level helpers of the denoiser's backward (training): the fold of the
// output gradient through the second v Linear, column sums, two-level slab reductions into the reference's tensor layouts,
// a small SGEMM, the gate MLP's backward, ShiftedSoftplus' backward.  The fused edge backward and the MFMA node kernels live
// in train_bwd_mfma.hip; the first-generation VALU kernels (test-only cross-check) in tests/xcheck/csrc/train_bwd_v1.hip.
// Math follows the reference modules (autograd of x2h_attention.py:43-97, h2x_attention.py:34-73,
// common.py:151-171); oracle/training.py + torch.autograd is the checker (tests/test_gpu_training.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edge_common.h"
#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

__constant__ float c_mu_b[G] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};



// ------------------------------------------------------------------------------------------------
// x2h: fold the output gradient through the second v Linear (the mirror image of the query fold):
//   Gt[i][a][m] = sum_c G[i][8a+c] Wbv[8a+c][m],   gb[i][a] = sum_c G[i][8a+c] bbv[8a+c]
// ------------------------------------------------------------------------------------------------
// LISTED (the pruned last x2h blocks of a backward): only the rows of a device-side list -- the edge kernel reads no other row's fold;
// a row's result depends on that row alone, so a listed launch writes the bits a full one would
// Persistent workgroups of 256 threads over tiles of 16 rows: thread (mq, ag) owns the columns 4 mq .. 4 mq + 3 of the heads 2 ag, 2 ag + 1
// and keeps its 64 weights in registers across tiles; the 8 KB a row produces leave as float4 stores (the first version -- 128 threads, one
// column each, its weights re-read per head with a 512-byte stride, 4-byte stores -- took 77 us per 16.5 k-row launch against the ~35 us the
// 135 MB it writes need).  Every output is the same fma chain over c = 0 .. 7 as before: the bits do not move.
constexpr int FOLD_GRAD_GRID = 512;     // two workgroups per CU
template <bool LISTED>
__global__ __launch_bounds__(256, 4) void fold_grad_kernel(const float* __restrict__ att, const float* __restrict__ Gr,
                                                        int n_nodes, float* __restrict__ Gt, float* __restrict__ gb,
                                                        const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    __shared__ __attribute__((aligned(16))) float sG[16][H];
    __shared__ int sRow[16];
    const int count = LISTED ? *n_rows_ptr : n_nodes;
    const int t = threadIdx.x, mq = t & 31, ag = t >> 5;
    const int n_tiles = (count + 15) / 16;
    if ((int)blockIdx.x >= n_tiles) return;
    float w[2][4][DH];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* src = att + A_WBV + (size_t)(4 * mq + j) * H + (2 * ag + hh) * DH;      // x2h layout [m][n]
            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            w[hh][j][0] = lo.x; w[hh][j][1] = lo.y; w[hh][j][2] = lo.z; w[hh][j][3] = lo.w;
            w[hh][j][4] = hi.x; w[hh][j][5] = hi.y; w[hh][j][6] = hi.z; w[hh][j][7] = hi.w;
        }
    float bb[DH];
#pragma unroll
    for (int cc = 0; cc < DH; ++cc) bb[cc] = att[A_BBV + (t & 15) * DH + cc];
    auto node = [&](int r) { return sRow[r]; };   // (also when not LISTED: with affine row numbers the compiler pipelines the row loop into 256 registers)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 16, live = min(16, count - row0);
        __syncthreads();                    // the previous tile's readers are done
        if (t < 16) sRow[t] = t < live ? (LISTED ? rows[row0 + t] : row0 + t) : 0;
        __syncthreads();
#pragma unroll
        for (int u = t; u < 16 * H / 4; u += 256) {
            const int r = u >> 5, c4 = u & 31;
            const size_t nd = (size_t)node(r);
            *reinterpret_cast<float4*>(&sG[r][4 * c4]) =
                r < live ? *reinterpret_cast<const float4*>(Gr + nd * H + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll 2
        for (int r = 0; r < live; ++r) {
            const size_t nd = (size_t)node(r);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int a = 2 * ag + hh;
                const float4 g0 = *reinterpret_cast<const float4*>(&sG[r][a * DH]), g1 = *reinterpret_cast<const float4*>(&sG[r][a * DH + 4]);
                const float g[DH] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sacc = 0.f;
#pragma unroll
                    for (int cc = 0; cc < DH; ++cc) sacc = fmaf(g[cc], w[hh][j][cc], sacc);
                    o[j] = sacc;
                }
                *reinterpret_cast<float4*>(Gt + (nd * HEADS + a) * H + 4 * mq) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        {   // gb: thread (r, a) = (t >> 4, t & 15)
            const int r = t >> 4, a = t & 15;
            if (r < live) {
                const size_t nd = (size_t)node(r);
                float sacc = 0.f;
#pragma unroll
                for (int cc = 0; cc < DH; ++cc) sacc = fmaf(sG[r][a * DH + cc], bb[cc], sacc);
                gb[nd * HEADS + a] = sacc;
            }
        }
    }
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
    // eight loads in flight per thread, added in slab order (the same sum, bit for bit, as the one-load-at-a-time loop the
    // compiler otherwise emits: load -> wait -> add, 32 dependent round trips per launch)
    float acc = 0.f;
    int s = g;
    for (; s + 7 * groups < n_slabs; s += 8 * groups) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u * groups) * slab_stride + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < n_slabs; s += groups) acc += src[(size_t)s * slab_stride + c];
    dst[(size_t)g * size + c] = acc;
}

// the same fold for several slab sets in one launch (blockIdx.z = job): bit-identical sums, one launch instead of one per set
__global__ __launch_bounds__(256) void slab_fold_multi_kernel(FoldBatch b, int groups) {
    const FoldJob& jb = b.j[blockIdx.z];
    const int c = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (c >= jb.size) return;
    const float* __restrict__ src = jb.src;
    const size_t slab_stride = jb.stride;
    const int n_slabs = jb.n_slabs;
    float acc = 0.f;
    int s = g;
    for (; s + 7 * groups < n_slabs; s += 8 * groups) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u * groups) * slab_stride + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < n_slabs; s += groups) acc += src[(size_t)s * slab_stride + c];
    jb.dst[(size_t)g * jb.size + c] = acc;
}

// dst[r][c] (or dst[c][r] if transpose) = sum_s src[s * slab_stride + r * src_ld + c]
__global__ void reduce_store_kernel(const float* __restrict__ src, int n_slabs, size_t slab_stride, int src_ld,
                                    int rows, int cols, float* __restrict__ dst, int dst_ld, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx % cols;
    float acc = 0.f;
    int s = 0;
    for (; s + 7 < n_slabs; s += 8) {       // eight loads in flight, added in slab order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * slab_stride + (size_t)r * src_ld + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < n_slabs; ++s) acc += src[(size_t)s * slab_stride + (size_t)r * src_ld + c];
    if (transpose) dst[(size_t)c * dst_ld + r] = acc; else dst[(size_t)r * dst_ld + c] = acc;
}

// several reduce_store pieces in one launch (blockIdx.y = piece)
__global__ void reduce_store_multi_kernel(RsBatch b) {
    const RsPiece& pc = b.p[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pc.rows * pc.cols) return;
    const int r = idx / pc.cols, c = idx % pc.cols;
    const float* src = pc.src + (size_t)r * pc.src_ld + c;
    float acc = 0.f;
    int s = 0;
    for (; s + 7 < pc.n_slabs; s += 8) {    // eight loads in flight, added in slab order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * pc.stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < pc.n_slabs; ++s) acc += src[(size_t)s * pc.stride];
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

// ONE kernel on the matrix pipe.  A wave owns tiles of 16 edge slots (half a node):
//   Y = R W1^T          [16 edges x 160 units], k = 20 rbf      50 v_mfma_f32_16x16x4_f32, Y stays in 40 registers
//                       D layout: lane (j, q), tile nt, register r <-> edge 4 q + r, unit 16 nt + j
//   per edge: LayerNorm statistics, the gate value and its gradient, the two sums of the LayerNorm backward -- reductions over the
//             units = over the 16 lanes of a row (DPP) and the 10 tiles
//   d W1 += dpre^T R    k = the 16 edges                        80 MFMAs into 20 accumulator tiles (register r of the D layout of
//                       dpre is the A operand of k-slot q, the rbf of edge 4 q + r at g = lane & 15 the B operand)
//   per unit: d b1, d gamma, d beta, d W2 accumulate in the lanes that own the unit.
// (Rounds 1 - 3 ran two VALU kernels: an edge pass that evaluated the 160 x 20 products four times per edge, 292 us for 528 k edge
// slots, and parked eight scalars per edge for a weight pass of 205 us.)  Nothing is parked here.  One slab (train.h GB_*) per WAVE.
// Pinned by the reference's recorded gradients of the six gate tensors (tests/test_gpu_training.py) and, lane by lane, by
// tests/test_lanesim_gate.py.
constexpr int GATE_WAVES = 4;
__global__ __launch_bounds__(GATE_WAVES * 64) void gate_bwd_mfma_kernel(const float* __restrict__ wts, const float* __restrict__ x,
                                                                      const int32_t* __restrict__ nbr,
                                                                      const int32_t* __restrict__ deg, int n_nodes,
                                                                      const float* __restrict__ de_w, float* __restrict__ partial,
                                                                      const int* __restrict__ rows, const int* __restrict__ n_rows_ptr) {
    // `rows` (optional): the nodes whose edges can carry a gate gradient (an h2x stack adds to de_w on its movable rows only)
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    constexpr int NT = GH / 16;      // 10 unit tiles
    constexpr int KS = G / 4;        // 5 k-steps of the first product
    __shared__ float sP[4][GH];      // b1 | gamma | beta | W2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    for (int u = tid; u < 4 * GH; u += GATE_WAVES * 64) (&sP[0][0])[u] = wts[GATE_B1 + u];      // B1, LNG, LNB, W2 are contiguous
    __syncthreads();
    // B operand of Y = R W1^T: lane (j, q) supplies W1[unit 16 nt + j][g = 4 s + q]
    float w1[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS; ++s) w1[nt][s] = wts[GATE_W1 + (size_t)(16 * nt + j) * G + 4 * s + q];
    const float b2 = wts[GATE_B2];
    float mu_a[KS];                  // rbf centres of this lane's A slots (g = 4 s + q) and B slots (g = j, 16 + (j & 3))
#pragma unroll
    for (int s = 0; s < KS; ++s) mu_a[s] = c_mu_b[4 * s + q];
    const float mu_b0 = c_mu_b[j], mu_b1 = c_mu_b[16 + (j & 3)];
    floatx4 dw[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dw[nt][0] = floatx4{0.f, 0.f, 0.f, 0.f}; dw[nt][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
    float aB1[NT], aG[NT], aBe[NT], aW2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { aB1[nt] = 0.f; aG[nt] = 0.f; aBe[nt] = 0.f; aW2[nt] = 0.f; }
    float aB2 = 0.f;
    const long n_tiles = (long)(rows ? *n_rows_ptr : n_nodes) * (KNN / 16);
    const long stride = (long)gridDim.x * GATE_WAVES;
    long tile = (long)blockIdx.x * GATE_WAVES + wave;
    // geometry of a tile: lane (j, *) <-> edge slot 16 (tile & 1) + j of node tile >> 1; fetched one tile ahead
    float dist_n = 0.f, dew_n = 0.f;
    auto fetch = [&](long tl) {
        const int i = rows ? rows[tl >> 1] : (int)(tl >> 1), sl = 16 * (int)(tl & 1) + j;
        const bool valid = sl < deg[i];
        const int nb = valid ? nbr[(size_t)i * KNN + sl] : i;
        const float dx = x[3 * i] - x[3 * nb], dy = x[3 * i + 1] - x[3 * nb + 1], dz = x[3 * i + 2] - x[3 * nb + 2];
        dist_n = sqrtf(dx * dx + dy * dy + dz * dz);
        dew_n = valid ? de_w[(size_t)i * KNN + sl] : 0.f;       // invalid slots: zero upstream gradient -> zero everywhere below
    };
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += stride) {
        const float dist = dist_n, dew = dew_n;
        fetch(tile + stride < n_tiles ? tile + stride : tile);
        // ---- Y = R W1^T + b1
        floatx4 y[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) y[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float t = dist - mu_a[s];
            const float ra = expf(-0.5f * (t * t));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) y[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra, w1[nt][s], y[nt], 0, 0, 0);
        }
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float b1 = sP[0][16 * nt + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) { y[nt][r] += b1; sum[r] += y[nt][r]; }
        }
        float mean[4], rstd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mean[r] = row16_sum(sum[r]) * (1.f / GH);
        float var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { y[nt][r] -= mean[r]; var[r] = fmaf(y[nt][r], y[nt][r], var[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) rstd[r] = 1.f / sqrtf(row16_sum(var[r]) * (1.f / GH) + 1e-5f);
        // ---- gate value -> d L / d (pre-sigmoid) of the four edges 4 q + r
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float gam = sP[1][16 * nt + j], bet = sP[2][16 * nt + j], w2 = sP[3][16 * nt + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[nt][r] *= rstd[r];                                      // y = the normalised pre-activation n from here on
                acc[r] = fmaf(w2, fmaxf(fmaf(y[nt][r], gam, bet), 0.f), acc[r]);
            }
        }
        float dacc[4], dist_e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = row16_sum(acc[r]) + b2;
            const float ew = 1.f / (1.f + expf(-a));
            dacc[r] = __shfl(dew, 4 * q + r, 64) * ew * (1.f - ew);
            dist_e[r] = __shfl(dist, 4 * q + r, 64);
        }
        // ---- LayerNorm backward: d n, its two per-edge sums, d pre (in place in y); per-unit sums on the way
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float gam = sP[1][16 * nt + j], bet = sP[2][16 * nt + j], w2 = sP[3][16 * nt + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n = y[nt][r];
                const float ya = fmaf(n, gam, bet);
                aW2[nt] = fmaf(dacc[r], fmaxf(ya, 0.f), aW2[nt]);
                const float dy = ya > 0.f ? dacc[r] * w2 : 0.f;
                aG[nt] = fmaf(dy, n, aG[nt]);
                aBe[nt] += dy;
                const float dn = dy * gam;
                s1[r] += dn;
                s2[r] = fmaf(dn, n, s2[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]) * (1.f / GH); s2[r] = row16_sum(s2[r]) * (1.f / GH); }
        if (j == 0) aB2 += (dacc[0] + dacc[1]) + (dacc[2] + dacc[3]);
        // B operand of d W1: rbf of edge 4 q + r at g = j (tile 0) and g = 16 + j (tile 1, j < 4)
        float rb0[4], rb1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t0 = dist_e[r] - mu_b0, t1 = dist_e[r] - mu_b1;
            rb0[r] = expf(-0.5f * (t0 * t0));
            rb1[r] = j < 4 ? expf(-0.5f * (t1 * t1)) : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float gam = sP[1][16 * nt + j], bet = sP[2][16 * nt + j], w2 = sP[3][16 * nt + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n = y[nt][r];
                const float dn = fmaf(n, gam, bet) > 0.f ? dacc[r] * w2 * gam : 0.f;
                const float dp = rstd[r] * (dn - s1[r] - n * s2[r]);
                aB1[nt] += dp;
                dw[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp, rb0[r], dw[nt][0], 0, 0, 0);
                dw[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp, rb1[r], dw[nt][1], 0, 0, 0);
            }
        }
    }
    // ---- this wave's slab.  d W1 tiles: lane (gc = j, qq = q), register rr <-> unit 16 nt + 4 qq + rr, g = 16 gt + gc
    float* slab = partial + ((size_t)blockIdx.x * GATE_WAVES + wave) * GB_SIZE;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int u = 16 * nt + 4 * q + rr;
            slab[GB_W1 + u * G + j] = dw[nt][0][rr];
            if (j < 4) slab[GB_W1 + u * G + 16 + j] = dw[nt][1][rr];
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {     // per-unit sums: over the four q groups
        const float v1 = xrow_sum(aB1[nt]), v2 = xrow_sum(aG[nt]), v3 = xrow_sum(aBe[nt]), v4 = xrow_sum(aW2[nt]);
        if (q == 0) {
            slab[GB_B1 + 16 * nt + j] = v1;
            slab[GB_LNG + 16 * nt + j] = v2;
            slab[GB_LNB + 16 * nt + j] = v3;
            slab[GB_W2 + 16 * nt + j] = v4;
        }
    }
    const float vb2 = xrow_sum(aB2);
    if (lane == 0) slab[GB_B2] = vb2;
}

// classifier: d(pre) = d(act) * sigmoid(pre)   (derivative of softplus(x) - ln 2)
__global__ void ssp_backward_kernel(const float* __restrict__ pre, const float* __restrict__ dact, long n,
                                    float* __restrict__ dpre) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dpre[idx] = dact[idx] / (1.f + expf(-pre[idx]));
}

// the same on a row list (device-side count): only the listed rows of pre / dact are read, only those rows of dpre written
__global__ __launch_bounds__(H) void ssp_backward_rows_kernel(const float* __restrict__ pre, const float* __restrict__ dact,
                                                              const int* __restrict__ rows, const int* __restrict__ n_rows_ptr,
                                                              float* __restrict__ dpre) {
    const int count = *n_rows_ptr;
    for (int k = blockIdx.x; k < count; k += gridDim.x) {
        const size_t idx = (size_t)rows[k] * H + threadIdx.x;
        dpre[idx] = dact[idx] / (1.f + expf(-pre[idx]));
    }
}

// classifier.2 on a row list: dW1[c][k] = sum over the listed rows i of dlogits[i][c] act[i][k]; one workgroup per class c, thread =
// (k, row group g of 8): group g walks the rows g, g + 8, ... four at a time (a single chain of ~800 dependent row gathers took
// 92 us), the eight partial sums are added in group order through LDS (no atomics, no slabs: deterministic for a given row list --
// the list itself is compacted with atomics, build_active_kernel, so its order, and with it the last bits of these sums, can differ
// from run to run)
constexpr int CLS_GROUPS = 8;
__global__ __launch_bounds__(H * CLS_GROUPS) void cls_w1_grad_rows_kernel(const float* __restrict__ dlogits, int C,
                                                                          const float* __restrict__ act,
                                                                          const int* __restrict__ rows,
                                                                          const int* __restrict__ n_rows_ptr,
                                                                          float* __restrict__ dW1) {
    __shared__ float part[CLS_GROUPS][H];
    const int count = *n_rows_ptr, c = blockIdx.x, k = threadIdx.x & (H - 1), g = threadIdx.x >> 7;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};       // four independent chains: the loads of four rows in flight
    int it = g;
    for (; it + 3 * CLS_GROUPS < count; it += 4 * CLS_GROUPS) {
        int r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = rows[it + u * CLS_GROUPS];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(dlogits[(size_t)r[u] * C + c], act[(size_t)r[u] * H + k], acc[u]);
    }
    for (; it < count; it += CLS_GROUPS) { const int r = rows[it]; acc[0] = fmaf(dlogits[(size_t)r * C + c], act[(size_t)r * H + k], acc[0]); }
    part[g][k] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (g == 0) {
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < CLS_GROUPS; ++u) sum += part[u][k];
        dW1[(size_t)c * H + k] = sum;
    }
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


hipError_t launch_fold_grad(const float* att, const float* Gr, int n_nodes, float* Gt, float* gb, hipStream_t s, const int* rows,
                            const int* n_rows) {
    if (n_nodes <= 0) return hipSuccess;
    const int tiles = (n_nodes + 15) / 16, grid = tiles < FOLD_GRAD_GRID ? tiles : FOLD_GRAD_GRID;
    if (rows) hipLaunchKernelGGL(fold_grad_kernel<true>, dim3(grid), dim3(256), 0, s, att, Gr, n_nodes, Gt, gb, rows, n_rows);
    else hipLaunchKernelGGL(fold_grad_kernel<false>, dim3(grid), dim3(256), 0, s, att, Gr, n_nodes, Gt, gb, rows, n_rows);
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

hipError_t launch_slab_fold_multi(const FoldBatch& b, int groups, hipStream_t s) {
    if (b.n == 0) return hipSuccess;
    int mx = 0;
    for (int k = 0; k < b.n; ++k) mx = b.j[k].size > mx ? b.j[k].size : mx;
    hipLaunchKernelGGL(slab_fold_multi_kernel, dim3((mx + 255) / 256, groups, b.n), dim3(256), 0, s, b, groups);
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

// product path: one fused kernel; `grid` slabs of GB_SIZE floats in `partial`, every one written (grid a multiple of GATE_WAVES)
hipError_t launch_gate_backward_mfma(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                                     const float* de_w, float* partial, int grid, hipStream_t s, const int* rows, const int* n_rows) {
    hipLaunchKernelGGL(gate_bwd_mfma_kernel, dim3(grid / GATE_WAVES), dim3(GATE_WAVES * 64), 0, s, packed, x, nbr, deg, n_nodes,
                       de_w, partial, rows, n_rows);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}


hipError_t launch_ssp_backward(const float* pre, const float* dact, long n, float* dpre, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ssp_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, dact, n, dpre);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_ssp_backward_rows(const float* pre, const float* dact, const int* rows, const int* n_rows, int max_rows,
                                    float* dpre, hipStream_t s) {
    if (max_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(ssp_backward_rows_kernel, dim3(max_rows < 2048 ? max_rows : 2048), dim3(H), 0, s, pre, dact, rows, n_rows,
                       dpre);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_cls_w1_grad_rows(const float* dlogits, int C, const float* act, const int* rows, const int* n_rows, float* dW1,
                                   hipStream_t s) {
    hipLaunchKernelGGL(cls_w1_grad_rows_kernel, dim3(C), dim3(H * CLS_GROUPS), 0, s, dlogits, C, act, rows, n_rows, dW1);
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

// libcbgx -- the neighbour-row gradient of the x2h edge backward without atomics (round 6).
//
// The backward of  pre[e] = PD[i] + PS[j_e] + ...  (x2h_attention.py:56-62 through the factored first Linear)
// scatters d pre[e] [256] (k | v) of every edge to the projection-gradient row of its SOURCE node j_e.  As fp32 atomics that is 128
// 64-lane global_atomic_add_f32 per destination node, and the compute unit's atomic path prices each at 57 - 65 ns whatever its shape
// (scripts/ubench/vmem.hip): 0.53 ms of a 0.83 ms launch.  Here the edge kernel stores d pre of edge (i, slot) into row 32 i + slot of
// `dE` (plain stores, 5.5 ns each) and this file turns the scatter into a gather:
//   * the incoming-edge lists of every source node (a CSR over edge ids, ascending) -- built ONCE per backward call, the graph is the
//     same for all 2 L attention blocks: count -> exclusive scan -> fill -> per-source sort (the fill's order depends on the atomics
//     that hand out the slots; the sort makes the summation order, and with it every bit of dP, reproducible);
//   * edge_rows_reduce_kernel: one wavefront per source node, lane l owns columns 4 l .. 4 l + 3 of the 1 KB row, eight rows in
//     flight, summed in list order -> dP[j][256:512].  HBM-bound: N x 32 KB read once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

__global__ __launch_bounds__(256) void rin_count_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg,
                                                         int n_nodes, int* __restrict__ cnt) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n_nodes * KNN) return;
    const int i = (int)(e >> 5), slot = (int)(e & 31);
    if (slot >= deg[i]) return;
    const int j = nbr[e];
    if ((unsigned)j < (unsigned)n_nodes) atomicAdd(&cnt[j], 1);
}

// exclusive scan of cnt [n] into ptr [n + 1] by ONE workgroup (once per training step: 16.5 k entries are three passes); cnt is
// zeroed on the way -- the fill kernel reuses it as the per-source cursor
constexpr int SCAN_T = 1024, SCAN_PER = 8;
__global__ __launch_bounds__(SCAN_T) void rin_scan_kernel(int* __restrict__ cnt, int n, int* __restrict__ ptr) {
    __shared__ int wsum[SCAN_T / 64];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += SCAN_T * SCAN_PER) {
        const int k0 = base + t * SCAN_PER;
        int v[SCAN_PER], s = 0;
#pragma unroll
        for (int u = 0; u < SCAN_PER; ++u) {
            v[u] = k0 + u < n ? cnt[k0 + u] : 0;
            s += v[u];
        }
        int incl = s;       // inclusive scan of the thread sums within the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        int run = carry_s + wbase + incl - s;
#pragma unroll
        for (int u = 0; u < SCAN_PER; ++u) {
            if (k0 + u < n) { ptr[k0 + u] = run; cnt[k0 + u] = 0; }
            run += v[u];
        }
        __syncthreads();
        if (t == SCAN_T - 1) carry_s = run;
        __syncthreads();
    }
    if (t == 0) ptr[n] = carry_s;
}

__global__ __launch_bounds__(256) void rin_fill_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg,
                                                        int n_nodes, const int* __restrict__ ptr, int* __restrict__ cursor,
                                                        int* __restrict__ tmp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n_nodes * KNN) return;
    const int i = (int)(e >> 5), slot = (int)(e & 31);
    if (slot >= deg[i]) return;
    const int j = nbr[e];
    if ((unsigned)j < (unsigned)n_nodes) tmp[ptr[j] + atomicAdd(&cursor[j], 1)] = (int)e;
}

// one wavefront per source node: its incoming edge ids in ascending order (rank sort: the ids are distinct).  Lists of up to 64
// entries (a kNN graph's in-degree is ~32) live in one register per lane; longer ones re-read the list from memory.
__global__ __launch_bounds__(256) void rin_sort_kernel(const int* __restrict__ ptr, const int* __restrict__ tmp, int n_nodes,
                                                        int* __restrict__ edges) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_nodes) return;
    const int b = ptr[j], len = ptr[j + 1] - b;
    if (len <= 64) {
        const int a = lane < len ? tmp[b + lane] : 0x7fffffff;
        int rank = 0;
        for (int m = 0; m < len; ++m) rank += __shfl(a, m, 64) < a ? 1 : 0;
        if (lane < len) edges[b + rank] = a;
    } else {
        for (int k = lane; k < len; k += 64) {
            const int a = tmp[b + k];
            int rank = 0;
            for (int m = 0; m < len; ++m) rank += tmp[b + m] < a ? 1 : 0;
            edges[b + rank] = a;
        }
    }
}

typedef float floatx4s __attribute__((ext_vector_type(4)));

// dP[j][256 + 4 l ..] = sum over the incoming edges of j, in list order, of dE[edge][4 l ..]
__global__ __launch_bounds__(256) void edge_rows_reduce_kernel(const float* __restrict__ dE, const int* __restrict__ rin_ptr,
                                                                const int* __restrict__ rin_edge, int n_nodes,
                                                                float* __restrict__ dP) {
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= n_nodes) return;
    const int b = __builtin_amdgcn_readfirstlane(rin_ptr[j]), e = __builtin_amdgcn_readfirstlane(rin_ptr[j + 1]);
    const floatx4s* rows = reinterpret_cast<const floatx4s*>(dE) + lane;
    floatx4s acc = {0.f, 0.f, 0.f, 0.f};
    int k = b;
    for (; k + 8 <= e; k += 8) {      // eight 1 KB rows in flight, added in list order
        int id[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) id[u] = __builtin_amdgcn_readfirstlane(rin_edge[k + u]);
        floatx4s v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(rows + (size_t)id[u] * (2 * H / 4));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (k < e) {
        const int rem = e - k;      // 1 .. 7 (wave-uniform)
        floatx4s v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int id = __builtin_amdgcn_readfirstlane(rin_edge[u < rem ? k + u : k]);
            v[u] = __builtin_nontemporal_load(rows + (size_t)id * (2 * H / 4));
        }
#pragma unroll
        for (int u = 0; u < 7; ++u)
            if (u < rem) acc += v[u];
    }
    *reinterpret_cast<floatx4s*>(dP + (size_t)j * PROW + 2 * H + 4 * lane) = acc;
}

// zero the listed rows of A [.][ld] (ld a multiple of 4): the projection gradient of an h2x block is non-zero -- and read -- on
// gen | nbr(gen) only, a fill of all N rows was 42 MB per block
// (+ `copy_dst` [copy_n] = `copy_src`, when given: the identity part of an h2x block's coordinate gradient rides along instead of
// taking a launch of its own between the two edge kernels of a layer)
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ A, int ld, const int* __restrict__ rows,
                                                         const int* __restrict__ n_rows_ptr, float* __restrict__ copy_dst,
                                                         const float* __restrict__ copy_src, int copy_n) {
    const int count = *n_rows_ptr, lane = threadIdx.x & 63;
    const int first = blockIdx.x * 4 + (threadIdx.x >> 6), stride = gridDim.x * 4;
    for (int it0 = first; it0 < count; it0 += 4 * stride) {
        int r[4];       // a wavefront's row numbers first, all in flight; then the stores (one dependent round trip, not one per row)
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = it0 + u * stride < count ? rows[it0 + u * stride] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r[u] >= 0) {
                floatx4s* row = reinterpret_cast<floatx4s*>(A + (size_t)r[u] * ld);
                for (int k = lane; k < ld / 4; k += 64) row[k] = floatx4s{0.f, 0.f, 0.f, 0.f};
            }
    }
    for (int k = blockIdx.x * 256 + threadIdx.x; k < copy_n; k += gridDim.x * 256) copy_dst[k] = copy_src[k];
}

// m[i] = 1 for every row of g [n][128] with a non-zero entry (one wavefront per row): the support of a caller's dL/dh_out, so that
// the backward's receptive-field pruning also applies when such a gradient exists (it is exactly zero outside: DiffBP's centre-of-mass
// head reads the features of the movable atoms and their neighbours only)
// (`cols` <= 128 columns per row; `set`: m[i] = 0 / 1 instead of only raising it)
__global__ __launch_bounds__(256) void mark_nonzero_rows_kernel(const float* __restrict__ g, int n, int cols, int set,
                                                                 uint8_t* __restrict__ m) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* row = g + (size_t)i * cols;
    const float a = lane < cols ? row[lane] : 0.f, b = lane + 64 < cols ? row[lane + 64] : 0.f;
    const bool any = __ballot(a != 0.f || b != 0.f) != 0;
    if (lane == 0) { if (set) m[i] = any ? 1 : 0; else if (any) m[i] = 1; }
}

hipError_t launch_mark_nonzero_rows(const float* g, int n, uint8_t* m, hipStream_t s, int cols, int set) {
    if (n <= 0) return hipSuccess;
    if (cols < 1 || cols > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mark_nonzero_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, g, n, cols, set, m);
    return hipGetLastError();
}

hipError_t launch_zero_rows(float* A, int ld, const int* rows, const int* n_rows_ptr, int max_rows, hipStream_t s, float* copy_dst,
                            const float* copy_src, int copy_n) {
    if (max_rows <= 0 && copy_n <= 0) return hipSuccess;
    // two workgroups per CU: a listed block has ~2 k rows, and 2048 workgroups that mostly read the count and leave took 12 us
    const int g = (max_rows + 3) / 4;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(g < 1 ? 1 : (g < 512 ? g : 512)), dim3(256), 0, s, A, ld, rows, n_rows_ptr, copy_dst,
                       copy_src, copy_dst ? copy_n : 0);
    return hipGetLastError();
}

hipError_t launch_rin_build(const int32_t* nbr, const int32_t* deg, int n_nodes, int* cnt, int* ptr, int* tmp, int* edges,
                            hipStream_t s) {
    if (n_nodes <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)n_nodes * sizeof(int), s);
    if (e != hipSuccess) return e;
    const unsigned eg = (unsigned)(((long)n_nodes * KNN + 255) / 256);
    hipLaunchKernelGGL(rin_count_kernel, dim3(eg), dim3(256), 0, s, nbr, deg, n_nodes, cnt);
    hipLaunchKernelGGL(rin_scan_kernel, dim3(1), dim3(SCAN_T), 0, s, cnt, n_nodes, ptr);
    hipLaunchKernelGGL(rin_fill_kernel, dim3(eg), dim3(256), 0, s, nbr, deg, n_nodes, ptr, cnt, tmp);
    hipLaunchKernelGGL(rin_sort_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, s, ptr, tmp, n_nodes, edges);
    return hipGetLastError();
}

hipError_t launch_edge_rows_reduce(const float* dE, const int* rin_ptr, const int* rin_edge, int n_nodes, float* dP, hipStream_t s) {
    if (n_nodes <= 0) return hipSuccess;
    profile_mark_begin(K_EDGE_ROWS_REDUCE, s);
    hipLaunchKernelGGL(edge_rows_reduce_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, s, dE, rin_ptr, rin_edge, n_nodes, dP);
    profile_mark_end(s);
    return hipGetLastError();
}

}  // namespace cbgx

// libcbgx -- the training step's input side as one launch, and its weight gradients as two:
//   PLContextEmbedder (repo/modules/context_emb.py:137-230, the shipped configuration: linear atom / residue / indicator embeddings)
//       protein row:  h = W_pa feat + b_pa + W_res onehot(aa) + b_res + (W_ind 0 + b_ind)
//       ligand row:   h = W_la c + b_la + (W_ind 1 + b_ind)
//   compose_context (repo/modules/common.py:189-214): cat(protein, ligand)[sort_idx] of the coordinates, the features and the movable
//       flag, sort_idx = stable argsort of the graph ids.
// Both embeddings are ONE product with a stacked weight matrix: every composed row gets an "extended input"
//       ext = [ feat (F) | onehot aa (A) | 1 if protein | c (C) | 1 if ligand | 0 ... ]          (EMB_LD columns)
// and h = ext . Wext with Wext rows = W_pa^T | W_res^T | b_pa + b_res + b_ind | W_la^T | b_la + W_ind + b_ind.  The forward kernel
// writes ext next to h; the backward is then the weight gradient of a single Linear,  dWext[c][j] = sum_r dh[r][c] ext[r][j]
// (wgrad_mfma_kernel + one reduce), from whose columns every parameter's gradient is a slice.  The tensor path takes ~20 launches
// forward and ~35 backward for the same numbers -- index_put's sort, three GEMMs with K = 27 / 20 / C that run at ~1 TFLOP/s, five
// column reductions -- 0.45 ms of a 14.7 ms step (profiles/trace_train_r06/).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

static_assert(H == 128 && EMB_LD == 128, "two columns per lane");

// One wave per composed row, lane l owns the columns 2l, 2l+1 of h and the entries l, l + 64 of ext; the stacked weights sit in LDS.
__global__ __launch_bounds__(256) void embed_compose_kernel(
    const float* __restrict__ x_rec, const float* __restrict__ x_lig, const float* __restrict__ feat, const int64_t* __restrict__ aa,
    const float* __restrict__ c_lig, const int64_t* __restrict__ sort_idx, const uint8_t* __restrict__ gen_rec,
    const uint8_t* __restrict__ gen_lig, int n_rec, int n_lig, int F, int A, int C, EmbedParams p, float* __restrict__ x,
    float* __restrict__ h, float* __restrict__ ext, uint8_t* __restrict__ gen) {
    extern __shared__ float smem[];
    const int J = F + A + C + 2, jp = F + A, jl = J - 1;
    float* Wt = smem;                     // [J][H]
    float* erow = smem + J * H;           // [4 waves][EMB_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < H * F; e += 256) Wt[(e % F) * H + e / F] = p.w_pa[e];
    for (int e = tid; e < H * A; e += 256) Wt[(F + e % A) * H + e / A] = p.w_res[e];
    for (int e = tid; e < H * C; e += 256) Wt[(jp + 1 + e % C) * H + e / C] = p.w_la[e];
    if (tid < H) {
        Wt[jp * H + tid] = (p.b_pa[tid] + p.b_res[tid]) + p.b_ind[tid];
        Wt[jl * H + tid] = p.b_la[tid] + (p.w_ind[tid] + p.b_ind[tid]);
    }
    __syncthreads();
    float* er = erow + w * EMB_LD;
    const int n = n_rec + n_lig;
    for (int r = blockIdx.x * 4 + w; r < n; r += gridDim.x * 4) {
        const long src = __builtin_amdgcn_readfirstlane((int)sort_idx[r]);
        const bool prot = src < n_rec;
        const long ls = src - n_rec;
        float e0 = 0.f, e1 = 0.f;          // ext[lane], ext[lane + 64]
        if (prot) {
            const int a = (int)aa[src];
            auto entry = [&](int j) { return j < F ? feat[src * F + j] : (j < jp ? (j - F == a ? 1.f : 0.f) : (j == jp ? 1.f : 0.f)); };
            e0 = entry(lane);
            e1 = entry(lane + 64);
        } else {
            auto entry = [&](int j) { return j > jp && j < jl ? c_lig[ls * C + (j - jp - 1)] : (j == jl ? 1.f : 0.f); };
            e0 = entry(lane);
            e1 = entry(lane + 64);
        }
        ext[(size_t)r * EMB_LD + lane] = e0;
        ext[(size_t)r * EMB_LD + 64 + lane] = e1;
        er[lane] = e0;
        er[lane + 64] = e1;
        __builtin_amdgcn_wave_barrier();   // the wave's own LDS row: written above, read below by all of its lanes, in program order
        if (lane < 3) x[(size_t)r * 3 + lane] = prot ? x_rec[src * 3 + lane] : x_lig[ls * 3 + lane];
        if (gen && lane == 0) gen[r] = prot ? (gen_rec ? gen_rec[src] : 0) : gen_lig[ls];
        const int j0 = prot ? 0 : jp + 1, j1 = prot ? jp + 1 : J;
        float2 acc = make_float2(0.f, 0.f);
        for (int j = j0; j < j1; ++j) {
            const float e = er[j];
            const float2 wv = *reinterpret_cast<const float2*>(Wt + j * H + 2 * lane);
            acc.x = fmaf(e, wv.x, acc.x);
            acc.y = fmaf(e, wv.y, acc.y);
        }
        *reinterpret_cast<float2*>(h + (size_t)r * H + 2 * lane) = acc;
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_embed_compose(const float* x_rec, const float* x_lig, const float* feat, const int64_t* aa, const float* c_lig,
                                const int64_t* sort_idx, const uint8_t* gen_rec, const uint8_t* gen_lig, int n_rec, int n_lig, int F,
                                int A, int C, const EmbedParams& p, float* x, float* h, float* ext, uint8_t* gen, hipStream_t s) {
    const int n = n_rec + n_lig;
    if (n <= 0) return hipSuccess;
    const int J = F + A + C + 2;
    const size_t lds = (size_t)(J * H + 4 * EMB_LD) * sizeof(float);
    // four rows per wave: a row is a chain of three dependent memory round trips (sort_idx -> its inputs -> the stores) that the row
    // loop does not overlap -- at 16 rows per wave the launch took 47 us for 16.5 k rows; the weight staging (J x 128 floats per
    // workgroup, from L2) costs one more round trip per workgroup whatever the grid
    int grid = (n + 15) / 16;
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    hipLaunchKernelGGL(embed_compose_kernel, dim3(grid), dim3(256), lds, s, x_rec, x_lig, feat, aa, c_lig, sort_idx, gen_rec, gen_lig,
                       n_rec, n_lig, F, A, C, p, x, h, ext, gen);
    return hipGetLastError();
}

// grad_out: [H x F | H x A | H | H x C | H] = dW_pa, dW_res, the protein bias column, dW_la, the ligand bias column (see the header)
hipError_t launch_embed_compose_backward(const float* grad_h, const float* ext, int n, int F, int A, int C, float* partial, int groups,
                                         float* grad_out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipError_t e = launch_wgrad_mfma(grad_h, H, ext, EMB_LD, n, 1, partial, EMB_LD, (size_t)H * EMB_LD, groups, s);
    if (e != hipSuccess) return e;
    const int off[5] = {0, F, F + A, F + A + 1, F + A + C + 1}, cols[5] = {F, A, 1, C, 1};
    RsBatch b;
    b.n = 5;
    float* dst = grad_out;
    for (int k = 0; k < 5; ++k) {
        b.p[k] = RsPiece{partial + off[k], dst, (size_t)H * EMB_LD, groups, EMB_LD, H, cols[k], cols[k], 0};
        dst += (size_t)H * cols[k];
    }
    return launch_reduce_store_multi(b, s);
}

// ------------------------------------------------------------------------------------------------
// compose_context's index work (repo/modules/common.py:189-214): the stable argsort of cat(protein graph ids, ligand graph ids) and what
// the callers derive from it -- the sorted ids, the ligand flag, the composed rows of the ligand atoms, the CSR offsets of the graphs --
// as three launches (count, scan, place) instead of the ~20 of the tensor path (radix sort, scatter, cumsum ...).  A stable sort by
// graph id is a counting sort: row = offset(graph) [+ protein atoms of the graph, for a ligand atom] + rank among the EARLIER atoms of
// the same array with the same id.  Collated batches have non-decreasing ids, where that rank is i - (atoms of earlier graphs); the count
// pass checks it, and an array that is not sorted takes the literal definition (a scan over the earlier entries: slow, exact).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compose_count_kernel(const int64_t* __restrict__ br, const int64_t* __restrict__ bl, int n_rec, int n_lig,
                                                            int B, int* __restrict__ cnt, int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool in = i < n_rec + n_lig;
    const bool lig = i >= n_rec;
    const int64_t* a = lig ? bl : br;
    const int k = lig ? i - n_rec : i;
    const int64_t g = in ? a[k] : 0;
    const bool ok = in && g >= 0 && g < B;
    if (in && !ok) atomicOr(flag, 4);
    if (ok && k > 0 && a[k - 1] > g) atomicOr(flag, lig ? 2 : 1);
    // one atomic per distinct id of a wavefront, not per atom: a training batch has 32 graphs, and 16.5 k adds onto 32 counters took
    // 98 us (profiles/kernel_stats_train_r06fin.csv); collated ids come in runs, so a wavefront holds one or two
    const int slot = ok ? (lig ? B : 0) + (int)g : -1;
    bool active = ok;
    for (;;) {
        const unsigned long long m = __ballot(active);
        if (!m) break;
        const int leader = __ffsll((long long)m) - 1;
        const int ls = __shfl(slot, leader, 64);
        const bool same = active && slot == ls;
        const unsigned long long sm = __ballot(same);
        if (lane == leader) atomicAdd(&cnt[ls], __popcll(sm));
        active = active && !same;
    }
}

// one workgroup: exclusive prefix sums over the graphs -- pre[0][g] / pre[1][g] = protein / ligand atoms of the graphs before g,
// graph_ptr[g] = both
__global__ __launch_bounds__(1024) void compose_scan_kernel(const int* __restrict__ cnt, int B, int* __restrict__ pre, int32_t* __restrict__ graph_ptr) {
    __shared__ int s_part[2][1024];
    const int t = threadIdx.x, per = (B + 1023) / 1024;
    const int g0 = t * per, g1 = min(B, g0 + per);
    int s0 = 0, s1 = 0;
    for (int g = g0; g < g1; ++g) { s0 += cnt[g]; s1 += cnt[B + g]; }
    s_part[0][t] = s0;
    s_part[1][t] = s1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan of the per-thread totals
        const int a0 = t >= off ? s_part[0][t - off] : 0, a1 = t >= off ? s_part[1][t - off] : 0;
        __syncthreads();
        s_part[0][t] += a0;
        s_part[1][t] += a1;
        __syncthreads();
    }
    int p0 = s_part[0][t] - s0, p1 = s_part[1][t] - s1;
    for (int g = g0; g < g1; ++g) {
        pre[g] = p0;
        pre[B + g] = p1;
        graph_ptr[g] = p0 + p1;
        p0 += cnt[g];
        p1 += cnt[B + g];
    }
    if (t == 1023) graph_ptr[B] = s_part[0][1023] + s_part[1][1023];
}

__global__ __launch_bounds__(256) void compose_place_kernel(const int64_t* __restrict__ br, const int64_t* __restrict__ bl, int n_rec, int n_lig,
                                                            int B, const int* __restrict__ cnt, const int* __restrict__ pre,
                                                            const int* __restrict__ flag, int64_t* __restrict__ sort_idx,
                                                            int64_t* __restrict__ batch_idx, uint8_t* __restrict__ lig_flag,
                                                            int64_t* __restrict__ lig_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rec + n_lig) return;
    const bool lig = i >= n_rec;
    const int64_t* a = lig ? bl : br;
    const int k = lig ? i - n_rec : i;
    const int64_t g64 = a[k];
    if (g64 < 0 || g64 >= B) return;
    const int g = (int)g64;
    int rank;
    if (*flag & (lig ? 2 : 1)) {
        rank = 0;
        for (int j = 0; j < k; ++j) rank += a[j] == g64 ? 1 : 0;
    } else
        rank = k - pre[(lig ? B : 0) + g];
    const int pos = pre[g] + pre[B + g] + (lig ? cnt[g] : 0) + rank;
    sort_idx[pos] = i;
    batch_idx[pos] = g64;
    lig_flag[pos] = lig ? 1 : 0;
    if (lig) lig_rows[k] = pos;
}

hipError_t launch_compose_plan(const int64_t* br, const int64_t* bl, int n_rec, int n_lig, int B, int* scratch, int64_t* sort_idx,
                               int64_t* batch_idx, uint8_t* lig_flag, int64_t* lig_rows, int32_t* graph_ptr, hipStream_t s) {
    // scratch: cnt [2 B] | pre [2 B] | flag [1]
    int *cnt = scratch, *pre = scratch + 2 * B, *flag = scratch + 4 * B;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(int) * (4 * (size_t)B + 1), s);
    if (e != hipSuccess) return e;
    const int n = n_rec + n_lig, grid = (n + 255) / 256;
    if (n > 0) hipLaunchKernelGGL(compose_count_kernel, dim3(grid), dim3(256), 0, s, br, bl, n_rec, n_lig, B, cnt, flag);
    hipLaunchKernelGGL(compose_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, B, pre, graph_ptr);
    if (n > 0)
        hipLaunchKernelGGL(compose_place_kernel, dim3(grid), dim3(256), 0, s, br, bl, n_rec, n_lig, B, cnt, pre, flag, sort_idx, batch_idx,
                           lig_flag, lig_rows);
    return hipGetLastError();
}

}  // namespace cbgx

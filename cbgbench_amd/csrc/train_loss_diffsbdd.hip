// libcbgx -- DiffSBDD's training arithmetic around the denoiser call (repo/models/diffusion/diffsbdd.py:91-195, training mode) as three
// launches, with the gradients of both losses with respect to the network outputs left behind by the same pass:
//   diffsbdd_noise_kernel   before the network: ligand centred on its mean, q(z_t | x) for coordinates and one-hot / 4 types with the
//                           pocket re-centred on the noisy ligand (DiffsbddVariationalScheduler.forward_pos_center_noise /
//                           forward_type_add_noise, diffusion_scheduler.py:740-790), and every loss term that does not depend on the
//                           network: the two KL priors (:846-868) and the discretised-Gaussian reconstruction term of the types at t = 0
//                           (:930-945)
//   diffsbdd_loss_kernel    after it: per graph 0.5 sum(err^2) [t != 0] / (n dim) + 0.5 sum(err^2) [t == 0] (coordinates) resp.
//                           -log p(. | z_0) [t == 0] (types) + KL prior (:886-900), and d loss / d prediction
//   diffsbdd_finish_kernel  mean over the graphs
// One workgroup per graph on the COMPOSED row order (per graph: protein rows, then ligand rows; `sort_idx` maps a composed row to its
// index in cat(protein, ligand)), so no assumption on the order of the caller's graph ids; sums over a graph's atoms are block
// reductions in a fixed order (the tensor path's index_add uses float atomics).  alpha(t) / sigma(t) come from tables the host makes
// with the tensor path's own operations (sqrt(sigmoid(-/+gamma))): the noised inputs agree with that path to the last bit of a mean.
// The tensor path (CBGX_FUSED_TRAINING_OPS=0, and evaluation mode always) takes ~150 small launches and their autograd for the same
// numbers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

constexpr int SB_GDATA = 4;       // per graph: n ligand atoms, KL prior (coordinates), KL prior (types), -log p(c | z_0) [t == 0]

// sum of v over the workgroup's 256 threads, in a fixed order; the result in every thread (s_red: [4] scratch)
__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__device__ __forceinline__ float std_normal_cdf(float v) { return 0.5f * (1.f + erff(v * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256) void diffsbdd_noise_kernel(
    const float* __restrict__ x0, const float* __restrict__ x_rec, const int64_t* __restrict__ v0, const float* __restrict__ eps_x,
    const float* __restrict__ eps_c, const uint8_t* __restrict__ gen, const int64_t* __restrict__ t, const int64_t* __restrict__ sort_idx,
    const int32_t* __restrict__ graph_ptr, int n_rec_total, int C, const float* __restrict__ alpha_tab, const float* __restrict__ sigma_tab,
    int T, float* __restrict__ x_t, float* __restrict__ xr_t, float* __restrict__ c_t, float* __restrict__ gdata) {
    __shared__ float s_red[4];
    __shared__ int s_cnt;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int r0 = graph_ptr[g], r1 = graph_ptr[g + 1];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int c = 0;
    for (int r = r0 + tid; r < r1; r += 256) c += sort_idx[r] >= n_rec_total ? 1 : 0;
    if (c) atomicAdd(&s_cnt, c);
    __syncthreads();
    const int nl = s_cnt, l0 = r1 - nl;                 // ligand rows are the tail of the graph's range
    const float cnt = (float)(nl > 0 ? nl : 1);
    const int tb = (int)t[g];
    const float a = alpha_tab[tb], s = sigma_tab[tb], aT = alpha_tab[T], sT = sigma_tab[T];
    // mean of the ligand, then of the noised centred ligand
    float m0[3], m2[3];
    {
        float p[3] = {0.f, 0.f, 0.f};
        for (int r = l0 + tid; r < r1; r += 256) {
            const long ai = sort_idx[r] - n_rec_total;
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] += x0[3 * ai + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) m0[k] = block_sum_256(p[k], s_red) / cnt;
        float q[3] = {0.f, 0.f, 0.f};
        for (int r = l0 + tid; r < r1; r += 256) {
            const long ai = sort_idx[r] - n_rec_total;
#pragma unroll
            for (int k = 0; k < 3; ++k) q[k] += a * (x0[3 * ai + k] - m0[k]) + s * eps_x[3 * ai + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) m2[k] = block_sum_256(q[k], s_red) / cnt;
    }
    for (int r = r0 + tid; r < l0; r += 256) {          // pocket: moved with the ligand twice (remove_mean_batch, then the re-centring)
        const long src = sort_idx[r];
#pragma unroll
        for (int k = 0; k < 3; ++k) xr_t[3 * src + k] = (x_rec[3 * src + k] - m0[k]) - m2[k];
    }
    float mu2p = 0.f, mu2a = 0.f, l0a = 0.f;
    const float sig0 = s * 4.f;
    for (int r = l0 + tid; r < r1; r += 256) {
        const long ai = sort_idx[r] - n_rec_total;
        const bool gn = gen[ai] != 0;
        const int v = (int)v0[ai];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xc = x0[3 * ai + k] - m0[k];
            const float xn = (a * xc + s * eps_x[3 * ai + k]) - m2[k];
            x_t[3 * ai + k] = gn ? xn : xc;
            const float mu = aT * xc;
            mu2p = fmaf(mu, mu, mu2p);
        }
        // types: one-hot / 4; the reconstruction term reads the noised types z_t (only graphs at t = 0 keep it)
        float mx = -INFINITY, se = 0.f, lv = 0.f;     // log-sum-exp over the classes, running (no per-class array)
        for (int k = 0; k < C; ++k) {
            const float c0 = k == v ? 0.25f : 0.f;
            const float cn = a * c0 + s * eps_c[ai * C + k];
            const float ct = gn ? cn : c0;
            c_t[ai * C + k] = ct;
            const float mu = aT * c0;
            mu2a = fmaf(mu, mu, mu2a);
            const float ctr = ct * 4.f - 1.f;
            const float lp = logf(std_normal_cdf((ctr + 0.5f) / sig0) - std_normal_cdf((ctr - 0.5f) / sig0) + 1e-10f);
            if (k == v) lv = lp;
            if (lp > mx) { se = se * expf(mx - lp) + 1.f; mx = lp; }
            else se += expf(lp - mx);
        }
        l0a -= (lv - (mx + logf(se))) * (0.25f * 4.f);
    }
    mu2p = block_sum_256(mu2p, s_red);
    mu2a = block_sum_256(mu2a, s_red);
    l0a = block_sum_256(l0a, s_red);
    if (tid == 0) {
        const float t0 = tb == 0 ? 1.f : 0.f;
        const float dp = (float)((nl - 1) * 3), da = 1.f;
        gdata[g * SB_GDATA + 0] = (float)nl;
        gdata[g * SB_GDATA + 1] = dp * logf(1.f / sT) + 0.5f * (dp * (sT * sT) + mu2p) - 0.5f * dp;
        gdata[g * SB_GDATA + 2] = da * logf(1.f / sT) + 0.5f * (da * (sT * sT) + mu2a) - 0.5f * da;
        gdata[g * SB_GDATA + 3] = l0a * t0;
    }
}

__global__ __launch_bounds__(256) void diffsbdd_loss_kernel(
    const float* __restrict__ x_out, const float* __restrict__ logits, const float* __restrict__ eps_x, const float* __restrict__ eps_c,
    const int64_t* __restrict__ t, const int64_t* __restrict__ sort_idx, const int32_t* __restrict__ graph_ptr, int n_rec_total, int B,
    int C, const float* __restrict__ gdata, float* __restrict__ glosses, float* __restrict__ x_pred, float* __restrict__ c_pred,
    float* __restrict__ gpos, float* __restrict__ gz) {
    __shared__ float s_red[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int r1 = graph_ptr[g + 1];
    const float n = gdata[g * SB_GDATA + 0];
    const int nl = (int)n, l0 = r1 - nl;
    const float t0 = t[g] == 0 ? 1.f : 0.f;
    const float inv_b = 1.f / (float)B;
    // d loss / d pred = -(tgt - pred) coef / B,   coef = (1 - t0) / (n dim) [+ t0 for the coordinates' reconstruction term]
    const float cp = ((1.f - t0) / (n * 3.f) + t0) * inv_b, ca = ((1.f - t0) / (n * (float)C)) * inv_b;
    float ep = 0.f, ea = 0.f;
    for (int r = l0 + tid; r < r1; r += 256) {
        const long ai = sort_idx[r] - n_rec_total;
        float sp = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p = x_out[3 * (long)r + k], d = eps_x[3 * ai + k] - p;
            x_pred[3 * ai + k] = p;
            gpos[3 * ai + k] = -d * cp;
            sp += d * d;
        }
        ep += sp;
        float sa = 0.f;
        for (int k = 0; k < C; ++k) {
            const float p = logits[(long)r * C + k], d = eps_c[ai * C + k] - p;
            c_pred[ai * C + k] = p;
            gz[ai * C + k] = -d * ca;
            sa += d * d;
        }
        ea += sa;
    }
    ep = block_sum_256(ep, s_red);
    ea = block_sum_256(ea, s_red);
    if (tid == 0) {
        glosses[2 * g + 0] = (0.5f * ep * (1.f - t0) / (n * 3.f) + 0.5f * ep * t0) + gdata[g * SB_GDATA + 1];
        glosses[2 * g + 1] = (0.5f * ea * (1.f - t0) / (n * (float)C) + gdata[g * SB_GDATA + 3]) + gdata[g * SB_GDATA + 2];
    }
}

__global__ __launch_bounds__(64) void diffsbdd_finish_kernel(const float* __restrict__ glosses, int B, float* __restrict__ losses) {
    const int k = threadIdx.x;
    if (k < 2) {
        float s = 0.f;
        for (int g = 0; g < B; ++g) s += glosses[2 * g + k];
        losses[k] = s / (float)B;
    }
}

hipError_t launch_diffsbdd_noise(const float* x0, const float* x_rec, const int64_t* v0, const float* eps_x, const float* eps_c,
                                 const uint8_t* gen, const int64_t* t, const int64_t* sort_idx, const int32_t* graph_ptr, int n_rec, int B,
                                 int C, const float* alpha_tab, const float* sigma_tab, int T, float* x_t, float* xr_t, float* c_t,
                                 float* gdata, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(diffsbdd_noise_kernel, dim3(B), dim3(256), 0, s, x0, x_rec, v0, eps_x, eps_c, gen, t, sort_idx, graph_ptr, n_rec, C,
                       alpha_tab, sigma_tab, T, x_t, xr_t, c_t, gdata);
    return hipGetLastError();
}

hipError_t launch_diffsbdd_loss(const float* x_out, const float* logits, const float* eps_x, const float* eps_c, const int64_t* t,
                                const int64_t* sort_idx, const int32_t* graph_ptr, int n_rec, int B, int C, const float* gdata,
                                float* glosses, float* losses, float* x_pred, float* c_pred, float* gpos, float* gz, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(diffsbdd_loss_kernel, dim3(B), dim3(256), 0, s, x_out, logits, eps_x, eps_c, t, sort_idx, graph_ptr, n_rec, B, C,
                       gdata, glosses, x_pred, c_pred, gpos, gz);
    hipLaunchKernelGGL(diffsbdd_finish_kernel, dim3(1), dim3(64), 0, s, glosses, B, losses);
    return hipGetLastError();
}

}  // namespace cbgx

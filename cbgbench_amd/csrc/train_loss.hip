// libcbgx -- the arithmetic of TargetDiff's training step around the denoiser, three launches instead of ~260 small ones:
//   train_noise_kernel      q(x_t | x_0), q(v_t | v_0) by Gumbel-argmax            targetdiff.py:87-93
//                           CTNVPScheduler.forward_add_noise  diffusion_scheduler.py:117-134
//                           TypeVPScheduler.forward_add_noise :339-365 (+ models/utils/categorical.py:5-32)
//   train_loss_kernel       position loss (type 'denoise', :185-201) and atom-type loss (KL of the categorical posteriors, decoder NLL at
//                           t = 0, :380-441), both scatter_mean over the graphs then mean over the graphs (targetdiff.py:103-121),
//                           and -- in the same pass -- their gradients with respect to the denoiser outputs
//   train_loss_bwd_kernel   scatters those gradients, scaled by the upstream gradients of the two losses, into dL/dx_out [N,3] and
//                           dL/dlogits [N,C] (zeros on protein rows)
// The training step is bound by the NUMBER of launches as much as by their duration (a dependent launch costs ~4.7 us on an
// otherwise busy queue: profiles/trace_train_r04n); nothing here is heavy.  One workgroup computes the losses (a few hundred
// to a few thousand ligand atoms, the class loops unrolled over registers): the per-graph sums live in LDS.  Sums over atoms use LDS
// float atomics, so the losses are reproducible up to summation order (as the reference's index_add on a GPU).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"
#include "train.h"

namespace cbgx {

// atom classes are register arrays indexed statically: the class loops run to the template bound LC (16 or 32) with k < C guards

__device__ __forceinline__ float lae(float a, float b) {      // _log_add_exp of the reference: max + log(exp(a - max) + exp(b - max))
    const float mx = fmaxf(a, b);
    return mx + logf(expf(a - mx) + expf(b - mx));
}
constexpr float LOG_TINY = -69.07755278982137f;     // log(1e-30): index_to_log_onehot clamps the one-hot at 1e-30

__global__ __launch_bounds__(256) void train_noise_kernel(
    const float* __restrict__ x0, const int64_t* __restrict__ v0, const int64_t* __restrict__ t,
    const int64_t* __restrict__ batch, const uint8_t* __restrict__ gen, int n_lig, int C,
    const float* __restrict__ acp, const float* __restrict__ log_acp, const float* __restrict__ log_1m_acp, float log_c,
    const float* __restrict__ eps, const float* __restrict__ u, float* __restrict__ x_t, float* __restrict__ c_t,
    int64_t* __restrict__ v_t) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_lig) return;
    const int tb = (int)t[batch[a]];
    const bool g = gen[a] != 0;
    const float ab = acp[tb];
    const float sa = sqrtf(ab), sb = sqrtf(1.0f - ab);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = x0[3 * a + k];
        // a.sqrt() * x + (1 - a).sqrt() * noise, each product rounded (no contraction into an fma: the reference's three kernels)
        const float xn = __fadd_rn(__fmul_rn(sa, x), __fmul_rn(sb, eps[3 * a + k]));
        x_t[3 * a + k] = g ? xn : x;
    }
    const int v = (int)v0[a];
    const float la = log_acp[tb], lb = log_1m_acp[tb] - log_c;
    int best = 0;
    float best_v = -INFINITY;
    for (int k = 0; k < C; ++k) {
        const float lq = lae((k == v ? 0.f : LOG_TINY) + la, lb);
        const float gum = -logf(-logf(u[(size_t)a * C + k] + 1e-30f) + 1e-30f);
        const float s = gum + lq;
        if (s > best_v) { best_v = s; best = k; }
    }
    const int vn = g ? best : v;
    for (int k = 0; k < C; ++k) c_t[(size_t)a * C + k] = k == vn ? 1.f : 0.f;
    v_t[a] = vn;
}

// LDS: s_pos [B], s_typ [B], cnt [B], one int (largest masked graph id), the block reduction [2][16], one int (bad graph id seen)
template <int LC, int THREADS>
__global__ __launch_bounds__(THREADS) void train_loss_kernel(
    const float* __restrict__ x_out, const float* __restrict__ logits, const int64_t* __restrict__ lig_rows,
    const float* __restrict__ x0, const int64_t* __restrict__ v0, const int64_t* __restrict__ vt,
    const int64_t* __restrict__ t, const int64_t* __restrict__ batch, const uint8_t* __restrict__ gen, int n_lig, int B, int C,
    const float* __restrict__ log_alpha, const float* __restrict__ log_1m_alpha, const float* __restrict__ log_acp,
    const float* __restrict__ log_1m_acp, float log_c, float* __restrict__ losses, float* __restrict__ x_pred,
    float* __restrict__ c_pred, float* __restrict__ gpos, float* __restrict__ gz) {
    extern __shared__ float lds[];
    float* s_pos = lds;
    float* s_typ = lds + B;
    float* cnt = lds + 2 * B;
    int* top = reinterpret_cast<int*>(lds + 3 * B);
    float* red = lds + 3 * B + 1;       // [2][16] block reduction
    int* bad = reinterpret_cast<int*>(lds + 3 * B + 1 + 32);      // an atom's graph id was outside [0, B)
    const int tid = threadIdx.x;
    for (int b = tid; b < 3 * B; b += blockDim.x) lds[b] = 0.f;
    if (tid == 0) { *top = 0; *bad = 0; }
    __syncthreads();
    for (int a = tid; a < n_lig; a += blockDim.x) {
        const int row = (int)lig_rows[a];
        // a malformed batch (num_graphs smaller than the largest graph id + 1) must not write outside the per-graph sums: such an
        // atom is skipped and both losses come back NaN (the tensor path raises an index error there)
        const int b_raw = (int)batch[a];
        const bool in_range = b_raw >= 0 && b_raw < B;
        if (!in_range) *bad = 1;
        const int b = in_range ? b_raw : 0;
        const int tb = (int)t[b];
        const bool g = gen[a] != 0 && in_range;
        // ---- positions: mse = sum_k (x_pred - x0)^2, d mse / d x_pred = 2 (x_pred - x0)
        float mse = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xp = x_out[3 * (size_t)row + k];
            const float d = xp - x0[3 * a + k];
            mse = __fadd_rn(mse, __fmul_rn(d, d));
            gpos[3 * a + k] = 2.f * d;
            if (x_pred) x_pred[3 * a + k] = xp;
        }
        // ---- atom types
        const float* z = logits + (size_t)row * C;
        float lp[LC];       // log_softmax(z)
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < LC; ++k) { lp[k] = k < C ? z[k] : -INFINITY; mx = fmaxf(mx, lp[k]); }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < LC; ++k) se += k < C ? expf(lp[k] - mx) : 0.f;
        const float lse = mx + logf(se);
        const int c0 = (int)v0[a], ct = (int)vt[a];
        const int tm1 = tb > 0 ? tb - 1 : 0;
        const float a0 = log_acp[tm1], b0 = log_1m_acp[tm1] - log_c;
        const float a1 = log_alpha[tb], b1 = log_1m_alpha[tb] - log_c;
        float unp[LC], unq[LC], Ap[LC];
        float pmx = -INFINITY, qmx = -INFINITY;
#pragma unroll
        for (int k = 0; k < LC; ++k) {
            lp[k] = lp[k] - lse;
            Ap[k] = lae(lp[k] + a0, b0);                                        // q_v_pred(log_c_pred, t - 1)
            const float Aq = lae((k == c0 ? 0.f : LOG_TINY) + a0, b0);         // q_v_pred(log_c0, t - 1)
            const float Bk = lae((k == ct ? 0.f : LOG_TINY) + a1, b1);         // q_v_pred_one_timestep(log_ct, t)
            unp[k] = k < C ? Ap[k] + Bk : -INFINITY;
            unq[k] = k < C ? Aq + Bk : -INFINITY;
            pmx = fmaxf(pmx, unp[k]);
            qmx = fmaxf(qmx, unq[k]);
        }
        float ps = 0.f, qs = 0.f;
#pragma unroll
        for (int k = 0; k < LC; ++k) {
            ps += k < C ? expf(unp[k] - pmx) : 0.f;
            qs += k < C ? expf(unq[k] - qmx) : 0.f;
        }
        const float plse = pmx + logf(ps), qlse = qmx + logf(qs);
        const float m0 = tb == 0 ? 1.f : 0.f;
        float kl = 0.f, nll = 0.f, G = 0.f;
        float gk[LC];       // d l / d log_p[k], then d l / d un_p[k], then d l / d log_softmax(z)[k]
#pragma unroll
        for (int k = 0; k < LC; ++k) {
            const float logp = unp[k] - plse, logq = unq[k] - qlse;
            const float q = expf(logq);
            const float e = k == c0 ? 1.f : 1e-30f;        // exp(log_c0)
            if (k < C) {
                kl += q * (logq - logp);
                nll -= e * logp;
                gk[k] = -(m0 * e + (1.f - m0) * q);
                G += gk[k];
            } else
                gk[k] = 0.f;
        }
        const float l_typ = m0 * nll + (1.f - m0) * kl;
        // through log_p = un - logsumexp(un): du[k] = g[k] - p[k] G.  Over k it sums to G (1 - sum p) = 0, and for the class that
        // holds almost all of p (the noisy type itself at most t: p and q both ~1 there) the direct form is the difference of two
        // numbers of size 1 that agree to 4 - 7 digits -- in fp32 that entry, the largest of the row, keeps 1 - 3 digits.  It is
        // therefore taken as minus the sum of the others, which are small numbers known to full relative precision.
        float du_others = 0.f;
        unsigned dom = 0u;      // one-hot: the first class at the maximum of un_p
#pragma unroll
        for (int k = 0; k < LC; ++k) {
            if (k < C) {
                gk[k] = gk[k] - expf(unp[k] - plse) * G;
                const bool is_dom = dom == 0u && unp[k] == pmx;
                dom |= is_dom ? 1u << k : 0u;
                du_others += is_dom ? 0.f : gk[k];
            }
        }
        float R = 0.f;
#pragma unroll
        for (int k = 0; k < LC; ++k) {
            if (k < C) {
                const float du = ((dom >> k) & 1u) ? -du_others : gk[k];
                gk[k] = du * expf((lp[k] + a0) - Ap[k]);             // through the log-add-exp: d A / d log_c_pred
                R += gk[k];
            }
        }
#pragma unroll
        for (int k = 0; k < LC; ++k) {
            if (k < C) {
                const float sm = expf(lp[k]);                         // softmax(z): also the reported c_pred
                gz[(size_t)a * C + k] = gk[k] - sm * R;               // through the log_softmax
                if (c_pred) c_pred[(size_t)a * C + k] = sm;
            }
        }
        if (g) {
            atomicAdd(&s_pos[b], mse);
            atomicAdd(&s_typ[b], l_typ);
            atomicAdd(&cnt[b], 1.f);
            atomicMax(top, b);
        }
    }
    __syncthreads();
    // losses = sum_b (s_b / max(cnt_b, 1)) / (largest masked graph id + 1): graph sums added in graph order per thread, then a tree
    const float n_eff = (float)(*top + 1);
    float lpz = 0.f, ltz = 0.f;
    for (int b = tid; b < B; b += blockDim.x) {
        const float c = fmaxf(cnt[b], 1.f);
        lpz += s_pos[b] / c;
        ltz += s_typ[b] / c;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lpz += __shfl_xor(lpz, o, 64); ltz += __shfl_xor(ltz, o, 64); }
    const int wave = tid >> 6, nw = blockDim.x >> 6;
    if ((tid & 63) == 0) { red[wave] = lpz; red[16 + wave] = ltz; }
    __syncthreads();
    if (tid == 0) {
        float sp = 0.f, st = 0.f;
        for (int w = 0; w < nw; ++w) { sp += red[w]; st += red[16 + w]; }
        // no generated atom at all: the mean over an empty set of graphs -- NaN, as the reference's scatter_mean of an empty
        // selection gives (diffusion_scheduler.py:199) -- not the 0 that n_eff = 1 would produce
        float any = 0.f;
        for (int b = 0; b < B; ++b) any += cnt[b];
        const bool undefined = *bad != 0 || any == 0.f;
        losses[0] = undefined ? __builtin_nanf("") : sp / n_eff;
        losses[1] = undefined ? __builtin_nanf("") : st / n_eff;
    }
    // per-atom gradients get the weight of their atom in the loss: gen / (max(cnt_b, 1) n_eff)
    for (int a = tid; a < n_lig; a += blockDim.x) {       // (each thread revisits the atoms it wrote)
        const int b_raw = (int)batch[a];
        const bool in_range = b_raw >= 0 && b_raw < B;
        const int b = in_range ? b_raw : 0;
        const float wgt = (gen[a] && in_range) ? 1.f / (fmaxf(cnt[b], 1.f) * n_eff) : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) gpos[3 * a + k] *= wgt;
        for (int k = 0; k < C; ++k) gz[(size_t)a * C + k] *= wgt;
    }
}

__global__ __launch_bounds__(256) void train_loss_bwd_kernel(const float* __restrict__ gpos, const float* __restrict__ gz,
                                                             const int64_t* __restrict__ sort_idx, int n_rec, int n_nodes, int C,
                                                             const float* __restrict__ g_pos, const float* __restrict__ g_typ,
                                                             float* __restrict__ grad_x, float* __restrict__ grad_logits) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_nodes) return;
    const int64_t src = sort_idx[r];      // composed row r holds entry `src` of cat(protein, ligand)
    const int a = (int)(src - n_rec);
    const float gp = g_pos ? *g_pos : 0.f, gt = g_typ ? *g_typ : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) grad_x[3 * (size_t)r + k] = a >= 0 ? gp * gpos[3 * a + k] : 0.f;
    for (int k = 0; k < C; ++k) grad_logits[(size_t)r * C + k] = a >= 0 ? gt * gz[(size_t)a * C + k] : 0.f;
}

#define CBGX_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return _e;               \
    } while (0)

hipError_t launch_train_noise(const float* x0, const int64_t* v0, const int64_t* t, const int64_t* batch, const uint8_t* gen,
                              int n_lig, int C, const float* acp, const float* log_acp, const float* log_1m_acp, float log_c,
                              const float* eps, const float* u, float* x_t, float* c_t, int64_t* v_t, hipStream_t s) {
    hipLaunchKernelGGL(train_noise_kernel, dim3((n_lig + 255) / 256), dim3(256), 0, s, x0, v0, t, batch, gen, n_lig, C, acp,
                       log_acp, log_1m_acp, log_c, eps, u, x_t, c_t, v_t);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_train_loss(const float* x_out, const float* logits, const int64_t* lig_rows, const float* x0,
                             const int64_t* v0, const int64_t* vt, const int64_t* t, const int64_t* batch, const uint8_t* gen,
                             int n_lig, int B, int C, const float* const* tables, float log_c, float* losses, float* x_pred,
                             float* c_pred, float* gpos, float* gz, hipStream_t s) {
    const size_t lds = ((size_t)3 * B + 1 + 32 + 1) * sizeof(float);
    // up to 16 classes (every shipped config): 16-wide class loops leave room for 512 threads, one or two atoms each at a
    // training batch's few hundred ligand atoms; up to 32 classes: 256 threads
    if (C <= 16)
        hipLaunchKernelGGL((train_loss_kernel<16, 512>), dim3(1), dim3(512), lds, s, x_out, logits, lig_rows, x0, v0, vt, t, batch,
                           gen, n_lig, B, C, tables[0], tables[1], tables[2], tables[3], log_c, losses, x_pred, c_pred, gpos, gz);
    else
        hipLaunchKernelGGL((train_loss_kernel<32, 256>), dim3(1), dim3(256), lds, s, x_out, logits, lig_rows, x0, v0, vt, t, batch,
                           gen, n_lig, B, C, tables[0], tables[1], tables[2], tables[3], log_c, losses, x_pred, c_pred, gpos, gz);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_train_loss_bwd(const float* gpos, const float* gz, const int64_t* sort_idx, int n_rec, int n_nodes, int C,
                                 const float* g_pos, const float* g_typ, float* grad_x, float* grad_logits, hipStream_t s) {
    hipLaunchKernelGGL(train_loss_bwd_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, gpos, gz, sort_idx, n_rec, n_nodes, C,
                       g_pos, g_typ, grad_x, grad_logits);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cbgx

// libcbgx -- the parts of one reverse-diffusion step that surround the network calls.  TargetDiff, as two kernels:
//   prologue: scatter the current ligand state into the composed node arrays
//             x[row] = x_lig ; h[row] = ligand_atom_emb(c_lig) + ligand_indicator(1)     (context_emb.py:179-230)
//   epilogue: posterior sampling of the next ligand state from the denoiser output
//             positions  CTNVPScheduler.backward_remove_noise(type='denoise')   diffusion_scheduler.py:144-165
//             atom types TypeVPScheduler.backward_remove_noise(pred_logit=True) :367-378, 407-441 + Gumbel argmax
//                        (models/utils/categorical.py:26-32)
// DiffBP (diffbp_epilogue_kernel) and DiffSBDD (diffsbdd_step_kernel): one kernel per step each, one wave per graph.
// Noise (eps ~ N(0,1), u ~ U(0,1)) is an input, so the host decides the generator and tests can replay a tape.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"

namespace cbgx {

constexpr int MAXC = 32;

__global__ __launch_bounds__(128) void step_prologue_kernel(const float* __restrict__ x_lig, const float* __restrict__ c_lig,
                                                            const int32_t* __restrict__ lig_rows, int n_lig, int C,
                                                            const float* __restrict__ emb_w /*[128][C]*/,
                                                            const float* __restrict__ emb_b, const float* __restrict__ ind_w,
                                                            const float* __restrict__ ind_b, float* __restrict__ x,
                                                            float* __restrict__ h, const int32_t* __restrict__ t_ptr) {
    const int a = blockIdx.x, m = threadIdx.x;
    if (a >= n_lig) return;
    if (t_ptr) {   // trajectory mode: x_lig / c_lig are the bases of [T+1] slot arrays; the current state is slot t + 1
        const size_t slot = (size_t)(*t_ptr + 1);
        x_lig += slot * n_lig * 3;
        c_lig += slot * n_lig * C;
    }
    const int row = lig_rows[a];
    float acc = 0.f;
    for (int k = 0; k < C; ++k) acc = fmaf(emb_w[m * C + k], c_lig[(size_t)a * C + k], acc);
    h[(size_t)row * H + m] = (acc + emb_b[m]) + (ind_w[m] + ind_b[m]);
    if (m < 3) x[3 * row + m] = x_lig[3 * a + m];
}

__device__ __forceinline__ float log_add_exp(float a, float b) {
    const float mx = fmaxf(a, b);
    return mx + logf(expf(a - mx) + expf(b - mx));
}

__global__ __launch_bounds__(256) void step_epilogue_kernel(
    const float* __restrict__ x_den, const float* __restrict__ logits, const int32_t* __restrict__ lig_rows,
    const float* __restrict__ x_lig, const float* __restrict__ c_lig, const uint8_t* __restrict__ gen_lig, int n_lig, int C,
    int t, const float* __restrict__ c0_tab, const float* __restrict__ ct_tab, const float* __restrict__ logvar_tab,
    const float* __restrict__ log_alpha, const float* __restrict__ log_1m_alpha, const float* __restrict__ log_acp,
    const float* __restrict__ log_1m_acp, float log_c, const float* __restrict__ eps, const float* __restrict__ u,
    float* __restrict__ x_next, float* __restrict__ c_next, int32_t* __restrict__ v_next,
    const int32_t* __restrict__ t_ptr) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_lig) return;
    if (t_ptr) {   // trajectory mode: state of slot t + 1 -> slot t of the same [T+1] arrays
        t = *t_ptr;
        x_lig += (size_t)(t + 1) * n_lig * 3;
        c_lig += (size_t)(t + 1) * n_lig * C;
        x_next += (size_t)t * n_lig * 3;
        c_next += (size_t)t * n_lig * C;
    }
    const int row = lig_rows[a];
    const bool gen = gen_lig[a] != 0;
    // ---- positions
    const float c0 = c0_tab[t], ct = ct_tab[t];
    const float sigma = t > 0 ? expf(0.5f * logvar_tab[t]) : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float xt = x_lig[3 * a + k];
        const float xs = (c0 * x_den[3 * row + k] + ct * xt) + sigma * eps[3 * a + k];
        x_next[3 * a + k] = gen ? xs : xt;
    }
    // ---- atom types
    const float* lg = logits + (size_t)row * C;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, lg[k]);
    float se = 0.f;
    for (int k = 0; k < C; ++k) se += expf(lg[k] - mx);
    const float lse = mx + logf(se);
    const int tm1 = t > 0 ? t - 1 : 0;
    const float a0 = log_acp[tm1], b0 = log_1m_acp[tm1] - log_c;
    const float a1 = log_alpha[t], b1 = log_1m_alpha[t] - log_c;
    float un[MAXC];
    float umx = -INFINITY;
    int cur = 0;
    float cur_v = -INFINITY;
    for (int k = 0; k < C; ++k) {
        const float ck = c_lig[(size_t)a * C + k];
        if (ck > cur_v) { cur_v = ck; cur = k; }
        const float lq0 = log_add_exp((lg[k] - lse) + a0, b0);
        const float lq1 = log_add_exp(logf(ck + 1e-8f) + a1, b1);
        un[k] = lq0 + lq1;
        umx = fmaxf(umx, un[k]);
    }
    float us = 0.f;
    for (int k = 0; k < C; ++k) us += expf(un[k] - umx);
    const float ulse = umx + logf(us);
    int best = 0;
    float best_v = -INFINITY;
    for (int k = 0; k < C; ++k) {
        const float g = -logf(-logf(u[(size_t)a * C + k] + 1e-30f) + 1e-30f);
        const float s = g + (un[k] - ulse);
        if (s > best_v) { best_v = s; best = k; }
    }
    const int v = gen ? best : cur;
    for (int k = 0; k < C; ++k) c_next[(size_t)a * C + k] = k == v ? 1.f : 0.f;
    if (v_next) v_next[a] = v;
}

// ---- TargetDiff, epilogue of step t and prologue of step t - 1 in ONE launch (round 5) ---------------------------------------------
// The prologue of a step consumes exactly what the epilogue of the step before it produced, so a sampler that runs step after step
// can have the epilogue write the composed rows as well: x[row] = x_next, h[row] = ligand_atom_emb(c_next) + ligand_indicator(1).
// One 128-thread workgroup per ligand atom (the prologue's shape): lanes 0..C-1 of its first wave evaluate the type posterior one
// class each -- the one-thread-per-atom epilogue walked 13 classes x 7 transcendental calls serially, 12 us for 25 atoms -- with
// every sum over the classes taken in the serial kernel's order (a broadcast loop), so the two kernels agree bit for bit; then all
// 128 threads write the feature row.  11.8 + 5.3 us of a 550 us one-graph step become one launch.
__global__ __launch_bounds__(128) void step_boundary_kernel(
    const float* __restrict__ x_den, const float* __restrict__ logits, const int32_t* __restrict__ lig_rows,
    const float* __restrict__ x_lig, const float* __restrict__ c_lig, const uint8_t* __restrict__ gen_lig, int n_lig, int C,
    int t, const float* __restrict__ c0_tab, const float* __restrict__ ct_tab, const float* __restrict__ logvar_tab,
    const float* __restrict__ log_alpha, const float* __restrict__ log_1m_alpha, const float* __restrict__ log_acp,
    const float* __restrict__ log_1m_acp, float log_c, const float* __restrict__ eps, const float* __restrict__ u,
    float* __restrict__ x_next, float* __restrict__ c_next, const float* __restrict__ emb_w, const float* __restrict__ emb_b,
    const float* __restrict__ ind_w, const float* __restrict__ ind_b, float* __restrict__ x, float* __restrict__ h) {
    __shared__ float s_c[MAXC];
    const int a = blockIdx.x, m = threadIdx.x;
    if (a >= n_lig) return;
    const int row = lig_rows[a];
    const bool gen = gen_lig[a] != 0;
    if (m < 64) {      // the first wave: lane k <-> class k (lanes >= C idle along)
        const int k = m;
        const bool live = k < C;
        const int kc = live ? k : 0;
        // ---- positions (lanes 0..2)
        if (k < 3) {
            const float c0 = c0_tab[t], ct = ct_tab[t];
            const float sigma = t > 0 ? expf(0.5f * logvar_tab[t]) : 0.f;
            const float xt = x_lig[3 * a + k];
            const float xs = (c0 * x_den[3 * row + k] + ct * xt) + sigma * eps[3 * a + k];
            const float xn = gen ? xs : xt;
            x_next[3 * a + k] = xn;
            x[3 * row + k] = xn;
        }
        // ---- atom types: the serial kernel's arithmetic, one class per lane; sums over classes in index order
        const float lgk = logits[(size_t)row * C + kc];
        float mx = live ? lgk : -INFINITY;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float ek = expf(lgk - mx);
        float se = 0.f;
        for (int j = 0; j < C; ++j) se += __shfl(ek, j, 64);
        const float lse = mx + logf(se);
        const int tm1 = t > 0 ? t - 1 : 0;
        const float a0 = log_acp[tm1], b0 = log_1m_acp[tm1] - log_c;
        const float a1 = log_alpha[t], b1 = log_1m_alpha[t] - log_c;
        const float ck = c_lig[(size_t)a * C + kc];
        const float lq0 = log_add_exp((lgk - lse) + a0, b0);
        const float lq1 = log_add_exp(logf(ck + 1e-8f) + a1, b1);
        const float un = lq0 + lq1;
        float umx = live ? un : -INFINITY;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) umx = fmaxf(umx, __shfl_xor(umx, o, 64));
        const float euk = expf(un - umx);
        float us = 0.f;
        for (int j = 0; j < C; ++j) us += __shfl(euk, j, 64);
        const float ulse = umx + logf(us);
        const float g = -logf(-logf(u[(size_t)a * C + kc] + 1e-30f) + 1e-30f);
        const float sk = g + (un - ulse);
        // first index of the maximum (the serial scan's strict '>'): of the current type vector and of the perturbed log-posterior
        float cmx = live ? ck : -INFINITY, smx = live ? sk : -INFINITY;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { cmx = fmaxf(cmx, __shfl_xor(cmx, o, 64)); smx = fmaxf(smx, __shfl_xor(smx, o, 64)); }
        const unsigned long long mc = __ballot(live && ck == cmx), ms = __ballot(live && sk == smx);
        const int cur = mc ? __builtin_ctzll(mc) : 0, best = ms ? __builtin_ctzll(ms) : 0;
        const int v = gen ? best : cur;
        if (live) {
            const float cn = k == v ? 1.f : 0.f;
            c_next[(size_t)a * C + k] = cn;
            s_c[k] = cn;
        }
    }
    __syncthreads();
    // ---- feature row of the next step (step_prologue_kernel)
    float acc = 0.f;
    for (int k = 0; k < C; ++k) acc = fmaf(emb_w[m * C + k], s_c[k], acc);
    h[(size_t)row * H + m] = (acc + emb_b[m]) + (ind_w[m] + ind_b[m]);
}

hipError_t launch_step_boundary(const float* x_den, const float* logits, const int32_t* lig_rows, const float* x_lig,
                                const float* c_lig, const uint8_t* gen_lig, int n_lig, int C, int t, const float* const* tabs,
                                float log_c, const float* eps, const float* u, float* x_next, float* c_next, const float* emb_w,
                                const float* emb_b, const float* ind_w, const float* ind_b, float* x, float* h, hipStream_t s) {
    if (n_lig == 0) return hipSuccess;
    hipLaunchKernelGGL(step_boundary_kernel, dim3(n_lig), dim3(128), 0, s, x_den, logits, lig_rows, x_lig, c_lig, gen_lig, n_lig, C,
                       t, tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], tabs[5], tabs[6], log_c, eps, u, x_next, c_next, emb_w, emb_b,
                       ind_w, ind_b, x, h);
    return hipGetLastError();
}

// ---- DiffBP: one wave per graph ----------------------------------------------------------------------------------------
// diffbp.py:262-297 after the two network calls:
//   noise  = x_den[lig] - x_in[lig],  noise -= mean_graph(noise)                 (CoMPredictor, diffbp.py:79-101)
//   shift  = mean_graph(x_com[lig] - x_in[lig])                                   (the H2X stack's mean displacement)
//   x_pred = noise + shift  ->  score step  CTNVPScheduler.backward_remove_noise(type='score')  diffusion_scheduler.py:154-158
//   types: MaskTypeSchedule.backward_remove_noise  :475-496  (masked + generated atoms take the argmax with prob (T - t)/T)
// The per-graph means are wave reductions in a fixed order (the reference's index_add is order-free).
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64) void diffbp_epilogue_kernel(
    const float* __restrict__ x_den, const float* __restrict__ x_com, const float* __restrict__ x_in,
    const float* __restrict__ logits, const int32_t* __restrict__ lig_rows, const int32_t* __restrict__ lig_ptr,
    const float* __restrict__ x_lig, const float* __restrict__ c_lig, const uint8_t* __restrict__ gen_lig, int C, int t, int T,
    const float* __restrict__ acp_tab, const float* __restrict__ beta_tab, int absorbing, const float* __restrict__ eps,
    const float* __restrict__ u, float* __restrict__ x_next, float* __restrict__ c_next) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int a0 = lig_ptr[g], a1 = lig_ptr[g + 1];
    float sn[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
    for (int a = a0 + lane; a < a1; a += 64) {
        const int row = lig_rows[a];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xi = x_in[3 * row + k];
            sn[k] += x_den[3 * row + k] - xi;
            sd[k] += x_com[3 * row + k] - xi;
        }
    }
    const float inv = 1.f / (float)max(a1 - a0, 1);
    float mn[3], sh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { mn[k] = wave_sum64(sn[k]) * inv; sh[k] = wave_sum64(sd[k]) * inv; }
    const float acp = acp_tab[t], beta = beta_tab[t];
    const float rs = 1.f / sqrtf(1.f - acp), rb = 1.f / sqrtf(1.f - beta), sb = t != 0 ? sqrtf(beta) : 0.f;
    const float prob = fminf(fmaxf((float)(T - t) / (float)T, 0.f), 1.f);
    for (int a = a0 + lane; a < a1; a += 64) {
        const int row = lig_rows[a];
        const bool gen = gen_lig[a] != 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xt = x_lig[3 * a + k];
            const float xp = ((x_den[3 * row + k] - x_in[3 * row + k]) - mn[k]) + sh[k];
            const float score = -xp * rs;
            const float xs = (xt + beta * score) * rb + sb * eps[3 * a + k];
            x_next[3 * a + k] = gen ? xs : xt;
        }
        // argmax of the current one-hot state and of softmax(logits) (= argmax of the logits, first index on ties)
        int cur = 0, best = 0;
        float cur_v = -INFINITY, best_v = -INFINITY;
        for (int k = 0; k < C; ++k) {
            const float ck = c_lig[(size_t)a * C + k], lk = logits[(size_t)row * C + k];
            if (ck > cur_v) { cur_v = ck; cur = k; }
            if (lk > best_v) { best_v = lk; best = k; }
        }
        const bool change = (u[a] < prob) && gen && cur == absorbing;
        const int v = change ? best : cur;
        for (int k = 0; k < C; ++k) c_next[(size_t)a * C + k] = k == v ? 1.f : 0.f;
    }
}

// ---- DiffSBDD: one wave per graph ---------------------------------------------------------------------------------------
// one iteration of diffsbdd.py:296-304 after the network call (DiffsbddVariationalScheduler.sample_p_zs_given_zt,
// diffusion_scheduler.py:1012-1040, for positions with com=True and for types with com=False):
//   mu = z_t * inv_alpha - coef * eps_pred;  z_s = mu + sigma * eps;   positions: z_s -= mean_ligand(z_s) and the whole pocket
//   of the graph is translated by the same vector.  inv_alpha / coef / sigma depend on the step only and come from tables
//   the host builds once with the scheduler's own expressions (cbgbench_amd/diffsbdd.py::step_tables).
// Besides the next ligand state the kernel leaves the NEXT step's composed inputs in place: x[rows of the graph] (moved
// pocket, new ligand positions) and h[ligand rows] = ligand_atom_emb(c) + ligand_indicator (context_emb.py:179-230).
// FRAMED mode (`frame` != NULL, [B][3], in / out): the composed x stays in the POCKET'S OWN FRAME -- the pocket rows are never
// touched, so the denoiser's static-context cache (its kNN lists, gate values and ligand-free layer-0/1 features) holds for all T
// steps -- and the translation the reference applies to the pocket every step is carried as the per-graph sum S of the removed
// means: true position = frame position - S.  The denoiser is translation-equivariant in x, so x_den (frame) - S is what the
// reference's network call returns; the ligand state x_next is kept in TRUE coordinates (trajectory output), the composed ligand
// rows get x_next + S_new = z_s + S_old.
__global__ __launch_bounds__(64) void diffsbdd_step_kernel(
    const float* __restrict__ x_den, const float* __restrict__ logits, const int32_t* __restrict__ graph_ptr,
    const int32_t* __restrict__ lig_rows, const int32_t* __restrict__ lig_ptr, const uint8_t* __restrict__ lig_flag,
    const float* __restrict__ x_lig, const float* __restrict__ c_lig, int C, float inv_alpha, float coef, float sigma,
    int do_x, int do_c, const float* __restrict__ eps_x, const float* __restrict__ eps_c, const float* __restrict__ emb_w,
    const float* __restrict__ emb_b, const float* __restrict__ ind_w, const float* __restrict__ ind_b,
    float* __restrict__ x_next, float* __restrict__ c_next, float* __restrict__ x, float* __restrict__ h,
    float* __restrict__ shift_out, float* __restrict__ frame) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int a0 = lig_ptr[g], a1 = lig_ptr[g + 1];
    float sm[3] = {0.f, 0.f, 0.f};
    float S[3] = {0.f, 0.f, 0.f};
    if (frame) { S[0] = frame[3 * g]; S[1] = frame[3 * g + 1]; S[2] = frame[3 * g + 2]; }
    if (do_x) {
        for (int a = a0 + lane; a < a1; a += 64) {
            const int row = lig_rows[a];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float zs = (x_lig[3 * a + k] * inv_alpha - coef * (x_den[3 * row + k] - S[k])) + sigma * eps_x[3 * a + k];
                x_next[3 * a + k] = zs;     // mean removed below
                sm[k] += zs;
            }
        }
    }
    const float inv = 1.f / (float)max(a1 - a0, 1);
    float mean[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) mean[k] = do_x ? wave_sum64(sm[k]) * inv : 0.f;
    if (lane < 3 && shift_out) shift_out[3 * g + lane] = mean[lane];
    float Sn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Sn[k] = S[k] + mean[k];
    __syncthreads();     // (one wave) every lane has read frame[] before lanes 0..2 overwrite it
    if (lane < 3 && frame) frame[3 * g + lane] = Sn[lane];
    // ligand atoms: centred positions, next types, and the composed rows of the next step
    for (int a = a0 + lane; a < a1; a += 64) {
        const int row = lig_rows[a];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = do_x ? x_next[3 * a + k] - mean[k] : x_lig[3 * a + k];
            x_next[3 * a + k] = v;
            x[3 * row + k] = v + Sn[k];      // (Sn = 0 without a frame)
        }
        for (int k = 0; k < C; ++k) {
            const float ck = c_lig[(size_t)a * C + k];
            c_next[(size_t)a * C + k] = do_c ? (ck * inv_alpha - coef * logits[(size_t)row * C + k]) + sigma * eps_c[(size_t)a * C + k] : ck;
        }
    }
    // pocket atoms of this graph move with the ligand's centre of mass (framed mode: they stay, S carries the move)
    if (do_x && !frame) {
        for (int row = graph_ptr[g] + lane; row < graph_ptr[g + 1]; row += 64)
            if (!lig_flag[row]) {
#pragma unroll
                for (int k = 0; k < 3; ++k) x[3 * row + k] -= mean[k];
            }
    }
    __syncthreads();   // one wave: orders this wave's c_next stores before the reads below
    for (int a = a0; a < a1; ++a) {
        const int row = lig_rows[a];
        for (int m = lane; m < H; m += 64) {
            float acc = 0.f;
            for (int k = 0; k < C; ++k) acc = fmaf(emb_w[m * C + k], c_next[(size_t)a * C + k], acc);
            h[(size_t)row * H + m] = (acc + emb_b[m]) + (ind_w[m] + ind_b[m]);
        }
    }
}

hipError_t launch_step_prologue(const float* x_lig, const float* c_lig, const int32_t* lig_rows, int n_lig, int C,
                                const float* emb_w, const float* emb_b, const float* ind_w, const float* ind_b, float* x,
                                float* h, hipStream_t s, const int32_t* t_ptr) {
    if (n_lig == 0) return hipSuccess;
    hipLaunchKernelGGL(step_prologue_kernel, dim3(n_lig), dim3(128), 0, s, x_lig, c_lig, lig_rows, n_lig, C, emb_w, emb_b,
                       ind_w, ind_b, x, h, t_ptr);
    return hipGetLastError();
}

__global__ void step_counter_kernel(int32_t* t_ptr) { *t_ptr -= 1; }

hipError_t launch_step_epilogue(const float* x_den, const float* logits, const int32_t* lig_rows, const float* x_lig,
                                const float* c_lig, const uint8_t* gen_lig, int n_lig, int C, int t,
                                const float* const* tabs, float log_c, const float* eps, const float* u, float* x_next,
                                float* c_next, int32_t* v_next, hipStream_t s, int32_t* t_ptr) {
    if (n_lig == 0) return hipSuccess;
    hipLaunchKernelGGL(step_epilogue_kernel, dim3((n_lig + 255) / 256), dim3(256), 0, s, x_den, logits, lig_rows, x_lig,
                       c_lig, gen_lig, n_lig, C, t, tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], tabs[5], tabs[6], log_c, eps,
                       u, x_next, c_next, v_next, t_ptr);
    if (t_ptr) hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(1), 0, s, t_ptr);   // next step: t - 1
    return hipGetLastError();
}

hipError_t launch_diffbp_epilogue(const float* x_den, const float* x_com, const float* x_in, const float* logits,
                                  const int32_t* lig_rows, const int32_t* lig_ptr, const float* x_lig, const float* c_lig,
                                  const uint8_t* gen_lig, int n_graphs, int C, int t, int T, const float* acp_tab,
                                  const float* beta_tab, int absorbing, const float* eps, const float* u, float* x_next,
                                  float* c_next, hipStream_t s) {
    if (n_graphs == 0) return hipSuccess;
    hipLaunchKernelGGL(diffbp_epilogue_kernel, dim3(n_graphs), dim3(64), 0, s, x_den, x_com, x_in, logits, lig_rows, lig_ptr,
                       x_lig, c_lig, gen_lig, C, t, T, acp_tab, beta_tab, absorbing, eps, u, x_next, c_next);
    return hipGetLastError();
}

hipError_t launch_diffsbdd_step(const float* x_den, const float* logits, const int32_t* graph_ptr, const int32_t* lig_rows,
                                const int32_t* lig_ptr, const uint8_t* lig_flag, const float* x_lig, const float* c_lig,
                                int n_graphs, int C, float inv_alpha, float coef, float sigma, int do_x, int do_c,
                                const float* eps_x, const float* eps_c, const float* emb_w, const float* emb_b,
                                const float* ind_w, const float* ind_b, float* x_next, float* c_next, float* x, float* h,
                                float* shift_out, float* frame, hipStream_t s) {
    if (n_graphs == 0) return hipSuccess;
    hipLaunchKernelGGL(diffsbdd_step_kernel, dim3(n_graphs), dim3(64), 0, s, x_den, logits, graph_ptr, lig_rows, lig_ptr,
                       lig_flag, x_lig, c_lig, C, inv_alpha, coef, sigma, do_x, do_c, eps_x, eps_c, emb_w, emb_b, ind_w, ind_b,
                       x_next, c_next, x, h, shift_out, frame);
    return hipGetLastError();
}

}  // namespace cbgx

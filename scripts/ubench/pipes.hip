// micro-benchmarks behind the split-f16 design of the x2h edge kernels: issue rates of the VALU forms the
// kernel uses, of the f32 / f16 MFMA forms, whether f16 MFMAs overlap VALU work (other wave of the SIMD / same wave), and
// whether a 4-byte-aligned ds_read_b128 is usable.
// build: hipcc --offload-arch=gfx950 -O3 pipes.hip -o pipes ; run on the GPU box (prints cycles per instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

enum { V_FMA, V_PKFMA, V_PKMUL, V_CVTPK, V_EXP, V_MIX, V_MAX, M_F32, M_F16K32, M_F16K16, N_MODES };
static const char* mode_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cvt_pkrtz_f16_f32", "v_exp_f32",
                                  "v_fma_mixlo_f16", "v_max_f32", "mfma_f32_16x16x4_f32", "mfma_f32_16x16x32_f16",
                                  "mfma_f32_16x16x16_f16"};
static const int per_iter[] = {32, 32, 32, 32, 32, 32, 32, 8, 8, 8};

template <int MODE>
__device__ __forceinline__ void body(float (&f)[8], float2v (&p)[8], floatx4 (&c)[8], half8 ha, half8 hb, float a, float b) {
    if (MODE == V_FMA) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(f[i], b, a);
    } else if (MODE == V_PKFMA) {
        const float2v b2 = {b, b}, a2 = {a, a};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], b2, a2);
    } else if (MODE == V_PKMUL) {
        const float2v b2 = {b, b};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = p[i] * b2;
    } else if (MODE == V_CVTPK) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned r;
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(f[i]), "v"(f[(i + 1) & 7]));
                f[i] = __uint_as_float(r);
            }
    } else if (MODE == V_EXP) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_exp2f(f[i]);
    } else if (MODE == V_MIX) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned r = __float_as_uint(f[i]);
                // lo half <- f16(f32 a * f32 b + f32 f[i+1])
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(r) : "v"(a), "v"(b), "v"(f[(i + 1) & 7]));
                f[i] = __uint_as_float(r);
            }
    } else if (MODE == V_MAX) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], b);
    } else if (MODE == M_F32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    } else if (MODE == M_F16K32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
    } else if (MODE == M_F16K16) {
        const half4 a4 = __builtin_shufflevector(ha, ha, 0, 1, 2, 3), b4 = __builtin_shufflevector(hb, hb, 0, 1, 2, 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[i], 0, 0, 0);
    }
}

// SPLIT: waves 0-3 (8-11) run mode MA, waves 4-7 (12-15) mode MB -- a workgroup's waves are dealt to the four SIMDs
// cyclically, so waves w and w + 4 share a SIMD; otherwise every wave runs MA then MB in each iteration.
// FINE: one MA instruction group followed by MB's, interleaved at single-instruction granularity inside one wave.
template <int MA, int MB, bool SPLIT>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float f[8];
    float2v p[8];
    floatx4 c[8];
    half8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = a + i; p[i] = float2v{a + i, a - i}; c[i] = floatx4{0, 0, 0, 0}; ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b * i); }
    const bool first = !SPLIT || ((wave >> 2) & 1) == 0, second = MB >= 0 && (!SPLIT || ((wave >> 2) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (first) body<MA>(f, p, c, ha, hb, a, b);
        if (MB >= 0) { if (second) body<(MB >= 0 ? MB : 0)>(f, p, c, ha, hb, a, b); }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y + c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one f16 MFMA followed by NV dependent-free v_fma_f32, eight times per iteration, in ONE wave: how many VALU hide behind an MFMA?
template <int NV, bool F32>
__global__ __launch_bounds__(1024) void kfine(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float f[8];
    floatx4 c[8];
    half8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = a + i; c[i] = floatx4{0, 0, 0, 0}; ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (F32) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
            else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) f[(i + v) & 7] = __builtin_fmaf(f[(i + v) & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool F32>
static void fine(float* d) {
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((kfine<NV, F32>), dim3(256), dim3(threads), 0, 0, d, 50);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((kfine<NV, F32>), dim3(256), dim3(threads), 0, 0, d, 20000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s MFMA + %d v_fma_f32 interleaved, %d wave/SIMD: %6.1f cyc per (MFMA + %d VALU) per SIMD\n", F32 ? "f32 16x16x4 " : "f16 16x16x32",
               NV, threads / 256, ms * 1e-3f * 2.4e9f / 20000 / 8 / (threads / 256) , NV);
    }
}

template <int MA, int MB, bool SPLIT>
static float run(float* d, int threads, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MA, MB, SPLIT>), dim3(256), dim3(threads), 0, 0, d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MA, MB, SPLIT>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e-3f * 2.4e9f / iters;   // cycles per iteration at 2.4 GHz
}

template <int M>
static void single(float* d) {
    for (int threads : {256, 512, 1024}) {
        const float cyc = run<M, -1, false>(d, threads, 20000);
        const int wps = threads / 256;
        printf("%-24s %d wave/SIMD: %8.1f cyc/iter  -> %6.2f cyc per instruction per SIMD\n", mode_name[M], wps, cyc,
               cyc / (per_iter[M] * wps));
    }
}

template <int MA, int MB>
static void pair(float* d) {
    const float a = run<MA, -1, false>(d, 256, 20000), b = run<MB, -1, false>(d, 256, 20000);
    const float sp = run<MA, MB, true>(d, 512, 20000), il = run<MA, MB, false>(d, 256, 20000), il2 = run<MA, MB, false>(d, 512, 20000);
    printf("%-22s + %-20s alone %7.1f / %7.1f | other wave of the SIMD %7.1f | same wave %7.1f | 2 waves, each both %7.1f  (cyc/iter)\n",
           mode_name[MA], mode_name[MB], a, b, sp, il, il2);
}

// ---- misaligned ds_read_b128 ------------------------------------------------------------------------------------------
__global__ void lds_kernel(float* out, int iters, int byte_off, int stride_dw) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    floatx4 acc = {0, 0, 0, 0};
    const unsigned base = (unsigned)(size_t)(lds) + lane * stride_dw * 4 + byte_off;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            floatx4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base), "n"(u * 1024));
            acc += v;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + 2 * acc.y + 3 * acc.z + 4 * acc.w;
}

static void lds_test(float* d) {
    float h[64];
    for (int stride : {4, 5, 6, 8}) for (int off : {0, 4, 8}) {
        hipLaunchKernelGGL(lds_kernel, dim3(1), dim3(64), 0, 0, d, 1, off, stride);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        bool ok = true;
        for (int l = 0; l < 64; ++l) {
            float e = 0;
            for (int u = 0; u < 8; ++u) { const int b = l * stride + off / 4 + u * 256; e += b + 2.f * (b + 1) + 3.f * (b + 2) + 4.f * (b + 3); }
            ok = ok && e == h[l];
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(256), 0, 0, d, 20000, off, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ds_read_b128 lane stride %d dwords, byte offset %d: %s, %6.1f cyc per (dependent) read, 4 waves/CU\n", stride, off,
               ok ? "values correct" : "VALUES WRONG", ms * 1e-3f * 2.4e9f / (20000 * 8));
    }
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    single<V_FMA>(d); single<V_PKFMA>(d); single<V_PKMUL>(d); single<V_CVTPK>(d); single<V_EXP>(d); single<V_MIX>(d); single<V_MAX>(d);
    single<M_F32>(d); single<M_F16K32>(d); single<M_F16K16>(d);
    pair<M_F32, V_FMA>(d); pair<M_F16K32, V_FMA>(d); pair<M_F16K32, V_PKFMA>(d); pair<M_F16K16, V_FMA>(d); pair<M_F16K32, M_F32>(d);
    pair<M_F16K32, V_CVTPK>(d); pair<M_F16K32, V_EXP>(d);
    fine<0, false>(d); fine<2, false>(d); fine<4, false>(d); fine<6, false>(d); fine<8, false>(d);
    fine<0, true>(d); fine<4, true>(d); fine<8, true>(d); fine<12, true>(d);
    return 0;
}

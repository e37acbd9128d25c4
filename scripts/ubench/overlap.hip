// micro-benchmark: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and fp32 VALU work overlap on one SIMD?
// build: hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// mode: 0 all waves MFMA, 1 all waves VALU, 2 even waves MFMA / odd waves VALU, 3 every wave interleaves both
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
    const bool do_mfma = mode == 0 || mode == 3 || (mode == 2 && (wave & 1) == 0);
    const bool do_valu = mode == 1 || mode == 3 || (mode == 2 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
            c0 = MFMA(a, b, c0); c1 = MFMA(a, b, c1); c2 = MFMA(a, b, c2); c3 = MFMA(a, b, c3);
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f0 = fmaf(f0, b, a); f1 = fmaf(f1, b, a); f2 = fmaf(f2, b, a); f3 = fmaf(f3, b, a);
                f4 = fmaf(f4, b, a); f5 = fmaf(f5, b, a); f6 = fmaf(f6, b, a); f7 = fmaf(f7, b, a);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[] = {"all waves MFMA (4/iter)", "all waves VALU (32 fma/iter)", "even MFMA / odd VALU", "every wave both"};
    for (int threads : {256, 512}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 100, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per-wave cycles at 2.4 GHz per iteration
            printf("threads/block=%d (%d waves/SIMD) mode %d %-30s %8.3f ms  %7.1f cycles/iter\n", threads, threads / 256, mode,
                   names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
        }
    }
    return 0;
}

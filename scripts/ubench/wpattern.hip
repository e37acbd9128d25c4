// Microbenchmark: HBM write bandwidth of the node projection's output (N rows of 640 floats = 2 560 B, 255 MB at N = 99 543) under
// different store schedules, no compute.  Question: is node_proj_kernel's 2.4 - 2.9 TB/s the price of its
// store pattern -- a wave writes one 256-byte piece of each of 16 rows per column chunk, the ten pieces of a row microseconds apart?
//   pattern 0  node_proj: workgroup = 4 waves x 16 rows; for chunk in 0..9: lane (c, q) writes float4 at row 4q + r, col 64 ch + 4c
//              (r = 0..3): per instruction 4 rows x 256 B; a row's ten pieces are `gap` dummy cycles apart
//   pattern 1  same pieces, chunk loop innermost: the ten 256-byte pieces of a row quad back to back (row-complete)
//   pattern 2  full rows: a wave writes 64 lanes x 16 B = 1 KB contiguous, 2.5 instructions per row (what an LDS-staged tile allows)
//   pattern 3  half rows: 1 280-byte runs (five chunks staged), lanes 0..79 of a 5-instruction group
//   pattern 4  node_qfold: 8 KB per row, lane (c, q) writes float4 at row 4q + r, offset 256 (2a + g) + 16 c bytes, 32 (a, g) steps
//   hipcc --offload-arch=gfx950 -O3 -o wpattern wpattern.hip && ./wpattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 640;

__device__ __forceinline__ void spin(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8); }

template <int PATTERN>
__global__ __launch_bounds__(256) void wkernel(float* __restrict__ P, int n_rows, int gap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int n_tiles = (n_rows + 63) / 64;
    const floatx4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 64 + wave * 16;
        if (PATTERN == 0) {
            for (int ch = 0; ch < 10; ++ch) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 4 * q + r;
                    if (row < n_rows) *reinterpret_cast<floatx4*>(P + (size_t)row * ROWF + 64 * ch + 4 * c) = v;
                }
                spin(gap);
            }
        } else if (PATTERN == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                for (int ch = 0; ch < 10; ++ch)
                    if (row < n_rows) *reinterpret_cast<floatx4*>(P + (size_t)row * ROWF + 64 * ch + 4 * c) = v;
            }
            spin(10 * gap);
        } else if (PATTERN == 2) {
            // 16 rows x 2560 B = 40 KB per wave = 40 instructions of 1 KB
            for (int k = 0; k < 40; ++k) {
                const size_t off = (size_t)row0 * ROWF + (size_t)k * 256 + 4 * lane;
                if (off + 4 <= (size_t)n_rows * ROWF) *reinterpret_cast<floatx4*>(P + off) = v;
            }
            spin(10 * gap);
        } else if (PATTERN == 3) {
            for (int half = 0; half < 2; ++half) {
                for (int row = 0; row < 16; ++row)
                    for (int k = 0; k < 2; ++k) {       // 1 280 B = 320 floats = 80 lanes x 4: one full + one quarter instruction
                        const int f = 256 * k + 4 * lane;
                        const int rr = row0 + row;
                        if (f < 320 && rr < n_rows) *reinterpret_cast<floatx4*>(P + (size_t)rr * ROWF + 320 * half + f) = v;
                    }
                spin(5 * gap);
            }
        }
    }
}

__global__ __launch_bounds__(256) void wkernel_fold(float* __restrict__ Q, int n_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int n_tiles = (n_rows + 63) / 64;
    const floatx4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * 64 + wave * 16;
        for (int ag = 0; ag < 32; ++ag)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                if (row < n_rows) *reinterpret_cast<floatx4*>(Q + (size_t)row * 2048 + 64 * ag + 4 * c) = v;
            }
    }
}

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main() {
    const int N = 99543;
    float *P, *Q;
    hipMalloc(&P, (size_t)N * ROWF * 4);
    hipMalloc(&Q, (size_t)N * 2048 * 4);
    const double mb = (double)N * ROWF * 4 / 1e6, mbq = (double)N * 2048 * 4 / 1e6;
    for (int grid : {512, 1024, 2048}) {
        for (int gap : {0, 4, 16}) {
            float t0 = time_us([&] { hipLaunchKernelGGL(wkernel<0>, dim3(grid), dim3(256), 0, 0, P, N, gap); }, 20);
            float t1 = time_us([&] { hipLaunchKernelGGL(wkernel<1>, dim3(grid), dim3(256), 0, 0, P, N, gap); }, 20);
            float t2 = time_us([&] { hipLaunchKernelGGL(wkernel<2>, dim3(grid), dim3(256), 0, 0, P, N, gap); }, 20);
            float t3 = time_us([&] { hipLaunchKernelGGL(wkernel<3>, dim3(grid), dim3(256), 0, 0, P, N, gap); }, 20);
            printf("grid %4d gap %2d | proj pieces %6.1f us %5.2f TB/s | row-complete %6.1f us %5.2f | full rows %6.1f us %5.2f | 1280-B runs %6.1f us %5.2f\n",
                   grid, gap, t0, mb / t0, t1, mb / t1, t2, mb / t2, t3, mb / t3);
        }
        float tf = time_us([&] { hipLaunchKernelGGL(wkernel_fold, dim3(grid), dim3(256), 0, 0, Q, N); }, 20);
        printf("grid %4d        | fold pattern (8 KB rows) %6.1f us %5.2f TB/s\n", grid, tf, mbq / tf);
    }
    return 0;
}

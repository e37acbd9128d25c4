// Microbenchmark: cost of vector-memory instruction shapes through one CU's texture addresser / L1 with 8 wavefronts competing, the
// occupancy of the x2h backward (one 512-thread workgroup per CU).  Each wave issues REPS batches of 16 independent instructions of
// one shape against an L2-resident buffer and reports cycles per instruction per CU (wall cycles of the workgroup / instructions of
// ONE wave: with 8 waves sharing the addresser, that is 8 x the addresser cost per instruction).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o vmem vmem.hip && ./vmem
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
constexpr int ROWF = 640;        // floats per row (the projection row of libcbgx)
constexpr int NROWS = 16384;     // 42 MB (the size of dP at 16.5 k nodes): L2 / MALL resident
constexpr int REPS = 200;

// shape 0: dwordx4, lane (c, q): row r_c, 16 bytes at 4q floats (+16 t): 16 rows x 64 B per instruction   (edge-major gather)
// shape 1: dword,   lane (c, q): row r_q', 4 bytes at c floats: 4 rows x 64 B per instruction
// shape 2: dwordx4, lanes 0..31 one row (512 B contiguous), lanes 32..63 another: 2 rows x 512 B
// shape 3: dwordx4, 64 lanes x 16 B contiguous (1 KB of one row pair)
// shape 4: dwordx2, 64 lanes x 8 B contiguous (512 B)
// shape 5: fp32 atomic add, 4 rows x 64 B per instruction (no return)
// shape 6: fp32 atomic add, 16 rows x 16 B (stride-2 dwords as in a pair labeling): lane (c,q): row r_q, dword 2c
// shape 7: dword store, 4 rows x 64 B
// shape 8: fp32 atomic add, ONE row x 256 B per instruction (lane = dword of the row)
// shape 9: fp32 atomic add, 2 rows x 128 B
// shapes 12 - 15: integer / 64-bit atomics (is the atomic path priced per lane-operation or per byte?)
// shapes 10, 11: as 5 and 8 on rows PRIVATE to the workgroup (64 rows each, shared by its 8 waves): what the x2h backward's neighbour
//            rows look like under the XCD-aware partition -- the rows of a graph are touched by one workgroup's neighbourhood only
template <int SHAPE>
__global__ __launch_bounds__(512) void vmem_kernel(float* buf, const int* rows, unsigned long long* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int* rw = rows + (blockIdx.x * 8 + wave) * 64;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int rep = 0; rep < REPS; ++rep) {
        const int base = (rep * 7) & 31;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (SHAPE == 0) {
                const int r = rw[(c + base) & 63];
                acc += *reinterpret_cast<const floatx4*>(buf + (size_t)r * ROWF + 16 * (k & 7) + 4 * q + 128 * (k >> 3));
            } else if (SHAPE == 1) {
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                acc[0] += buf[(size_t)r * ROWF + 16 * (k >> 2) + c];
            } else if (SHAPE == 2) {
                const int r = rw[(2 * k + (lane >> 5) + base) & 63];
                acc += *reinterpret_cast<const floatx4*>(buf + (size_t)r * ROWF + 4 * (lane & 31));
            } else if (SHAPE == 3) {
                const int r = rw[(k + base) & 63];
                acc += *reinterpret_cast<const floatx4*>(buf + (size_t)r * ROWF + 4 * lane);
            } else if (SHAPE == 4) {
                const int r = rw[(k + base) & 63];
                const floatx2 v = *reinterpret_cast<const floatx2*>(buf + (size_t)r * ROWF + 2 * lane);
                acc[0] += v[0]; acc[1] += v[1];
            } else if (SHAPE == 5) {
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                atomicAdd(buf + (size_t)r * ROWF + 16 * (k >> 2) + c, 1.0f);
            } else if (SHAPE == 6) {
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                atomicAdd(buf + (size_t)r * ROWF + 32 * (k >> 2) + 2 * c, 1.0f);
            } else if (SHAPE == 7) {
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                buf[(size_t)r * ROWF + 256 + 16 * (k >> 2) + c] = (float)k;
            } else if (SHAPE == 8) {
                const int r = rw[(k + base) & 63];
                atomicAdd(buf + (size_t)r * ROWF + 64 * (rep & 3) + lane, 1.0f);
            } else if (SHAPE == 9) {
                const int r = rw[(2 * k + (lane >> 5) + base) & 63];
                atomicAdd(buf + (size_t)r * ROWF + 32 * (rep & 7) + (lane & 31), 1.0f);
            } else if (SHAPE == 10) {
                const int r = blockIdx.x * 64 + (rw[(4 * (k & 3) + q + base) & 63] & 63);
                atomicAdd(buf + (size_t)r * ROWF + 16 * (k >> 2) + c, 1.0f);
            } else if (SHAPE == 11) {
                const int r = blockIdx.x * 64 + (rw[(k + base) & 63] & 63);
                atomicAdd(buf + (size_t)r * ROWF + 64 * (rep & 3) + lane, 1.0f);
            } else if (SHAPE == 12) {       // u64 integer add, 4 rows x 128 B (16 lanes x 8 B): two 32-bit fixed-point columns per lane-operation
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                atomicAdd(reinterpret_cast<unsigned long long*>(buf + (size_t)r * ROWF + 32 * (k >> 2) + 2 * c), 0x0000000100000001ull);
            } else if (SHAPE == 13) {       // f64 add, same addresses
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                atomicAdd(reinterpret_cast<double*>(buf + (size_t)r * ROWF + 32 * (k >> 2) + 2 * c), 1.0);
            } else if (SHAPE == 14) {       // u32 integer add, 4 rows x 64 B
                const int r = rw[(4 * (k & 3) + q + base) & 63];
                atomicAdd(reinterpret_cast<unsigned*>(buf + (size_t)r * ROWF + 16 * (k >> 2) + c), 1u);
            } else {                        // u64 integer add, 2 rows x 256 B (32 lanes x 8 B)
                const int r = rw[(2 * k + (lane >> 5) + base) & 63];
                atomicAdd(reinterpret_cast<unsigned long long*>(buf + (size_t)r * ROWF + 64 * (rep & 3) + 2 * (lane & 31)), 0x0000000100000001ull);
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) buf[0] = 1.f;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
static void run(const char* name, float* buf, int* rows, unsigned long long* out) {
    hipLaunchKernelGGL(vmem_kernel<SHAPE>, dim3(256), dim3(512), 0, 0, buf, rows, out);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(vmem_kernel<SHAPE>, dim3(256), dim3(512), 0, 0, buf, rows, out);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), out, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per = s / 256 / (REPS * 16.0);
    printf("%-58s %8.1f clock64 ticks / instruction / wave (8 waves per CU)   kernel %.3f ms -> %.1f ns per instruction per CU\n", name, per, ms,
           ms * 1e6 / (REPS * 16.0 * 8));
}

int main() {
    float* buf; int* rows; unsigned long long* out;
    hipMalloc(&buf, (size_t)NROWS * ROWF * sizeof(float));
    hipMemset(buf, 0, (size_t)NROWS * ROWF * sizeof(float));
    std::vector<int> h(256 * 8 * 64);
    srand(1);
    for (auto& v : h) v = rand() % NROWS;
    hipMalloc(&rows, h.size() * sizeof(int));
    hipMemcpy(rows, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * sizeof(unsigned long long));
    run<0>("0 dwordx4 gather, 16 rows x 64 B (edge-major)", buf, rows, out);
    run<1>("1 dword gather, 4 rows x 64 B", buf, rows, out);
    run<2>("2 dwordx4, 2 rows x 512 B", buf, rows, out);
    run<3>("3 dwordx4, 1 KB contiguous", buf, rows, out);
    run<4>("4 dwordx2, 512 B contiguous", buf, rows, out);
    run<5>("5 atomic add f32, 4 rows x 64 B", buf, rows, out);
    run<6>("6 atomic add f32, 4 rows x 16 lanes stride 2 dwords", buf, rows, out);
    run<7>("7 dword store, 4 rows x 64 B", buf, rows, out);
    run<8>("8 atomic add f32, 1 row x 256 B", buf, rows, out);
    run<9>("9 atomic add f32, 2 rows x 128 B", buf, rows, out);
    run<10>("10 atomic add f32, 4 rows x 64 B, workgroup-private rows", buf, rows, out);
    run<11>("11 atomic add f32, 1 row x 256 B, workgroup-private rows", buf, rows, out);
    run<12>("12 atomic add u64, 4 rows x 128 B", buf, rows, out);
    run<13>("13 atomic add f64, 4 rows x 128 B", buf, rows, out);
    run<14>("14 atomic add u32, 4 rows x 64 B", buf, rows, out);
    run<15>("15 atomic add u64, 2 rows x 256 B", buf, rows, out);
    return 0;
}

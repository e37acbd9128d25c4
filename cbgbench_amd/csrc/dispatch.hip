// libcbgx -- stage dispatch and weight-packing copies.
//
// The product library (libcbgx.so) contains one implementation of every stage: the MFMA / register-resident kernels of
// graph_mfma.hip, node_mfma.hip and edge_mfma.hip.  The first-generation VALU kernels (tests/xcheck/csrc/: kernels_v1.hip, train_bwd_v1.hip)
// are compiled only into the test-only library libcbgx_xcheck.so (-DCBGX_XCHECK), where cbgx_debug_set_edge_kernel(1)
// (include/cbgx_xcheck.h) routes the same entry points through them as an independent on-device cross-check.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "layout.h"

namespace cbgx {

#ifdef CBGX_XCHECK
int g_edge_impl = 0;  // 0: what libcbgx.so runs; 1: first-generation VALU kernels; 2: as 0 with the second-generation x2h backward
#endif

#define CBGX_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return _e;               \
    } while (0)

hipError_t launch_knn(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int32_t* nbr,
                      int32_t* deg, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
#ifdef CBGX_XCHECK
    if (g_edge_impl == 1) return launch_knn_v1(x, graph_ptr, n_graphs, n_nodes, nbr, deg, s);
#endif
    return launch_knn_reg(x, graph_ptr, n_graphs, n_nodes, nbr, deg, s);
}

hipError_t launch_gate(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                       float* e_w, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
#ifdef CBGX_XCHECK
    if (g_edge_impl == 1) return launch_gate_v1(packed, x, nbr, deg, n_nodes, e_w, s);
#endif
    return launch_gate_mfma(packed, x, nbr, deg, n_nodes, e_w, s);
}

// node projection + query fold + fused edge kernel of one attention block
hipError_t launch_attention(bool x2h, const float* att, const float* x, const float* h, const int32_t* nbr,
                            const int32_t* deg, const uint8_t* lig, const uint8_t* gen, const float* e_w, int n_nodes,
                            float* P, float* Qt, float* qbuf, float* out, float* dx_out, const int* act, const int* act_count,
                            const int* src, const int* src_count, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
#ifdef CBGX_XCHECK
    if (g_edge_impl == 1) return launch_attention_v1(x2h, att, x, h, nbr, deg, lig, gen, e_w, n_nodes, P, Qt, out, dx_out, s);
#endif
    hipError_t e0 = launch_node_mfma(att, h, lig, n_nodes, P, qbuf, Qt, act, act_count, src, src_count, s, x2h);
    if (e0 != hipSuccess) return e0;
    return launch_edge_mfma(x2h, att, x, h, P, Qt, nbr, deg, lig, gen, e_w, n_nodes, out, dx_out, act, act_count, s);
}

// ---- strided / transposed copies of cbgx_pack_weights ---------------------------------------------------------------
__global__ void pack_copy_kernel(const float* __restrict__ src, int src_ld, int src_off, int transpose,
                                 float* __restrict__ dst, int dst_ld, int rows, int cols) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    int i = idx / cols, j = idx % cols;
    dst[(size_t)i * dst_ld + j] = transpose ? src[(size_t)j * src_ld + i + src_off] : src[(size_t)i * src_ld + j + src_off];
}

// many copies in one launch (blockIdx.y = piece): weight packing is ~40 small copies per attention block
__global__ void pack_copy_multi_kernel(PackBatch b) {
    const PackPiece& pc = b.p[blockIdx.y];
    const int total = pc.rows * pc.cols;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx / pc.cols, j = idx % pc.cols;
        pc.dst[(size_t)i * pc.dst_ld + j] =
            pc.transpose ? pc.src[(size_t)j * pc.src_ld + i + pc.src_off] : pc.src[(size_t)i * pc.src_ld + j + pc.src_off];
    }
}

hipError_t launch_pack_copy_multi(const PackBatch& b, hipStream_t s) {
    if (b.n == 0) return hipSuccess;
    int mx = 0;
    for (int k = 0; k < b.n; ++k) mx = b.p[k].rows * b.p[k].cols > mx ? b.p[k].rows * b.p[k].cols : mx;
    int gx = (mx + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pack_copy_multi_kernel, dim3(gx, b.n), dim3(256), 0, s, b);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_pack_copy(const float* src, int src_ld, int src_off, int transpose, float* dst, int dst_ld,
                            int rows, int cols, hipStream_t s) {
    int total = rows * cols;
    hipLaunchKernelGGL(pack_copy_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, src_ld, src_off, transpose,
                       dst, dst_ld, rows, cols);
    CBGX_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cbgx

/* libcbgx_xcheck.so -- TEST-ONLY build of libcbgx (same sources compiled with -DCBGX_XCHECK).
 *
 * It exports everything include/cbgx.h declares plus the switch below, and additionally contains the first-generation
 * VALU kernels (tests/xcheck/csrc/kernels_v1.hip, train_bwd_v1.hip).  Those implement the same stages as the MFMA
 * kernels of libcbgx.so with different code, which makes them an independent on-device cross-check at sizes the CPU
 * oracle cannot reach (tests/test_gpu_parity.py, tests/test_gpu_training.py).  The product library libcbgx.so contains
 * neither the switch nor those kernels. */
#ifndef CBGX_XCHECK_H
#define CBGX_XCHECK_H

#ifdef __cplusplus
extern "C" {
#endif

/* 0 = the kernels libcbgx.so always runs, 1 = first-generation VALU kernels, 2 = as 0 but the x2h backward of the
 * second generation (one workgroup per node, train_bwd_mfma.hip) instead of train_bwd_x2h.hip.
 * Returns the previous setting (>= 0) or CBGX_E_INVALID.  Process-wide. */
int cbgx_debug_set_edge_kernel(int impl);

#ifdef __cplusplus
}
#endif
#endif

// internal declarations of the training (backward) kernels, train_bwd.hip / api_train.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace cbgx {

// per-workgroup partial gradient slab of edge_backward_kernel (floats)
constexpr int PB_WR = 0;                          // [4][20][256]  rbf columns of the first Linear (k | v)
constexpr int PB_WT = PB_WR + NT * G * 2 * H;     // [4][256]      type columns
constexpr int PB_LNG = PB_WT + NT * 2 * H;        // [256]         LayerNorm gamma (k | v)
constexpr int PB_LNB = PB_LNG + 2 * H;            // [256]         LayerNorm beta
constexpr int PB_WBV16 = PB_LNB + 2 * H;          // h2x: [16][128] second v Linear
constexpr int PB_BBV16 = PB_WBV16 + HEADS * H;    // h2x: [16]
constexpr int PB_SIZE = PB_BBV16 + HEADS;

// per-workgroup slab of the node-level reductions of one attention block
constexpr int NS_WBK = 0;                      // [128][128] second k Linear
constexpr int NS_WBV = NS_WBK + H * H;         // [128][128] x2h second v Linear
constexpr int NS_WQ1 = NS_WBV + H * H;         // [128][128] second q Linear
constexpr int NS_V1B = NS_WQ1 + H * H;         // [128] x2h second v bias
constexpr int NS_Q1B = NS_V1B + H;             // [128]
constexpr int NS_DP = NS_Q1B + H;              // [640] column sums of dP (first-Linear biases)
constexpr int NS_SIZE = NS_DP + PROW;

// partial slab of gate_bwd_weight_kernel
constexpr int GB_W1 = 0;                 // [160][20]
constexpr int GB_B1 = GB_W1 + GH * G;
constexpr int GB_LNG = GB_B1 + GH;
constexpr int GB_LNB = GB_LNG + GH;
constexpr int GB_W2 = GB_LNB + GH;
constexpr int GB_B2 = GB_W2 + GH;
constexpr int GB_SIZE = GB_B2 + 4;

#ifdef CBGX_XCHECK
hipError_t launch_edge_backward(bool x2h, const float* att, const float* x, const float* P, const float* Qt,
                                const float* Gt, const float* gb, const float* gx_out, const int32_t* nbr,
                                const int32_t* deg, const uint8_t* lig, const float* e_w, const int* rows,
                                const int* n_rows, int n_nodes, float* T, float* S, float* sw, float* dP, float* dx,
                                float* de_w, float* partial, int grid, hipStream_t s);
#endif
hipError_t launch_edge_backward_mfma(bool x2h, const float* att, const float* x, const float* P, const float* Qt,
                                     const float* Gt, const float* gb, const float* gx_out, const int32_t* nbr,
                                     const int32_t* deg, const uint8_t* lig, const float* e_w, const int* rows,
                                     const int* n_rows, int n_nodes, float* T, float* S, float* sw, float* dP, float* dx,
                                     float* de_w, float* partial, int grid, hipStream_t s, int centred = 0);
// third-generation x2h backward (train_bwd_x2h.hip): one wavefront per node, 8 nodes in flight per workgroup; P must be the
// centred projection of the MFMA node kernel.  grid = edge_grid_x2h(n) workgroups, one PB_SIZE slab each.  work_ctr: 8 ints, ZERO at
// launch (the per-XCD counters of the last, partial round of nodes).
hipError_t launch_edge_backward_x2h(const float* att, const float* x, const float* P, const float* Qt, const float* Gt,
                                    const float* gb, const int32_t* nbr, const int32_t* deg, const uint8_t* lig,
                                    const float* e_w, const int* rows, const int* n_rows, int n_nodes, float* T, float* S,
                                    float* sw, float* dP, float* dx, float* de_w, float* partial, float* nk_scratch,
                                    int* work_ctr, int grid, hipStream_t s, float* dE = nullptr);
// Round 6, the neighbour-row gradients without atomics (train_scatter.hip).  `dE` [N][32][256]: the edge backward (full launches)
// stores d pre of every edge (k | v) in the edge's own row; the incoming-edge lists of every SOURCE node -- `rin_ptr` [N + 1] offsets,
// `rin_edge` edge ids 32 i + slot in ascending order -- are built once per backward call (the graph is the same for all layers), and
// launch_edge_rows_reduce writes dP[j][256:512] = sum over the incoming edges of j, in list order (bit-reproducible).
// launch_rin_build: `cnt` [N] and `tmp` [32 N] are scratch; `ptr` [N + 1], `edges` [32 N] the result.
hipError_t launch_rin_build(const int32_t* nbr, const int32_t* deg, int n_nodes, int* cnt, int* ptr, int* tmp, int* edges,
                            hipStream_t s);
hipError_t launch_mark_nonzero_rows(const float* g, int n, uint8_t* m, hipStream_t s, int cols = H, int set = 0);
hipError_t launch_zero_rows(float* A, int ld, const int* rows, const int* n_rows_ptr, int max_rows, hipStream_t s,
                            float* copy_dst = nullptr, const float* copy_src = nullptr, int copy_n = 0);
hipError_t launch_edge_rows_reduce(const float* dE, const int* rin_ptr, const int* rin_edge, int n_nodes, float* dP, hipStream_t s);
// floats of nk_scratch: two slots (key | value path) per wave of the largest grid: the normalised pre-activation of a path, parked by its
// forward part and read back by the backward sweep in the same labeling (the key path also across two phases of a node)
constexpr size_t BX_NK_FLOATS = (size_t)256 * 8 * 2 * (KNN * H + 128);
hipError_t launch_fold_grad(const float* att, const float* Gr, int n_nodes, float* Gt, float* gb, hipStream_t s,
                            const int* rows = nullptr, const int* n_rows = nullptr);
hipError_t launch_outer_accum_mfma(bool headed, const float* Lm, const float* R, const int* rows, const int* n_rows,
                                   int n_nodes, float* partial, size_t slab_stride, int grid, hipStream_t s);
// (`rows` / `n_rows`, wgrad and dgrad: restrict the node dimension to a device-side list -- the rows outside it are known to be zero)
hipError_t launch_wgrad_mfma(const float* Lm, int ldl, const float* R, int ldr, int n_nodes, int col_blocks, float* partial,
                             int ldo, size_t slab_stride, int groups, hipStream_t s, const int* rows = nullptr,
                             const int* n_rows = nullptr);
// `C_in` (accumulate only): the addend is read from C_in instead of C (same leading dimension) -- C = C_in + A B^T out of place
hipError_t launch_dgrad_mfma(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int K,
                             int accumulate, hipStream_t s, const float* C_in = nullptr, const int* rows = nullptr,
                             const int* n_rows = nullptr);
hipError_t launch_q_backward_mfma(const float* att, const float* P, const float* T, const int* rows, const int* n_rows,
                                  int n_nodes, float* qs, float* dqb, float* zb, float* dP, float* partial, int grid,
                                  hipStream_t s);
#ifdef CBGX_XCHECK
hipError_t launch_q_backward(const float* att, const float* P, const float* T, const int* rows, const int* n_rows,
                             int n_nodes, float* qs, float* dqb, float* zb, float* dP, float* partial, int grid,
                             hipStream_t s);
#endif
#ifdef CBGX_XCHECK
hipError_t launch_outer_accum(bool headed, const float* Lm, const float* R, const int* rows, const int* n_rows,
                              int n_nodes, float* partial, size_t slab_stride, int grid, hipStream_t s);
#endif
hipError_t launch_compose_plan(const int64_t* br, const int64_t* bl, int n_rec, int n_lig, int B, int* scratch, int64_t* sort_idx,
                               int64_t* batch_idx, uint8_t* lig_flag, int64_t* lig_rows, int32_t* graph_ptr, hipStream_t s);
// train_loss_diffsbdd.hip: DiffSBDD's noising + data-only loss terms (one launch), its two losses and their gradients (two)
hipError_t launch_diffsbdd_noise(const float* x0, const float* x_rec, const int64_t* v0, const float* eps_x, const float* eps_c,
                                 const uint8_t* gen, const int64_t* t, const int64_t* sort_idx, const int32_t* graph_ptr, int n_rec, int B,
                                 int C, const float* alpha_tab, const float* sigma_tab, int T, float* x_t, float* xr_t, float* c_t,
                                 float* gdata, hipStream_t s);
hipError_t launch_diffsbdd_loss(const float* x_out, const float* logits, const float* eps_x, const float* eps_c, const int64_t* t,
                                const int64_t* sort_idx, const int32_t* graph_ptr, int n_rec, int B, int C, const float* gdata,
                                float* glosses, float* losses, float* x_pred, float* c_pred, float* gpos, float* gz, hipStream_t s);
// train_embed.hip: PLContextEmbedder + compose_context of a training step in one launch; its weight gradients through wgrad_mfma
constexpr int EMB_LD = 128;            // columns of the extended input rows (feat | onehot aa | 1 | c | 1 | zeros)
constexpr int EMB_MAX_J = 120;         // F + A + C + 2: the stacked weights (J x 128 floats) + four rows fit 64 KB of LDS
struct EmbedParams { const float *w_pa, *b_pa, *w_res, *b_res, *w_la, *b_la, *w_ind, *b_ind; };
hipError_t launch_embed_compose(const float* x_rec, const float* x_lig, const float* feat, const int64_t* aa, const float* c_lig,
                                const int64_t* sort_idx, const uint8_t* gen_rec, const uint8_t* gen_lig, int n_rec, int n_lig, int F,
                                int A, int C, const EmbedParams& p, float* x, float* h, float* ext, uint8_t* gen, hipStream_t s);
hipError_t launch_embed_compose_backward(const float* grad_h, const float* ext, int n, int F, int A, int C, float* partial, int groups,
                                         float* grad_out, hipStream_t s);
hipError_t launch_colsum(const float* A, int lda, int cols, const float* scale, const int* rows, const int* n_rows_ptr,
                         int n_rows, float* partial, size_t slab_stride, int grid, hipStream_t s);
// up to RS_MAX reduce_store pieces in one launch
constexpr int RS_MAX = 32;
struct RsPiece {
    const float* src; float* dst; size_t stride; int n_slabs, src_ld, rows, cols, dst_ld, transpose;
};
struct RsBatch {
    RsPiece p[RS_MAX]; int n;
};
hipError_t launch_reduce_store_multi(const RsBatch& b, hipStream_t s);
hipError_t launch_slab_fold(const float* src, int n_slabs, size_t slab_stride, int size, int groups, float* dst,
                            hipStream_t s);
// up to FOLD_JOBS_MAX first-level slab folds in one launch (blockIdx.z = job); every job folds into `groups` slabs
constexpr int FOLD_JOBS_MAX = 4;
struct FoldJob { const float* src; float* dst; size_t stride; int n_slabs, size; };
struct FoldBatch { FoldJob j[FOLD_JOBS_MAX]; int n; };
hipError_t launch_slab_fold_multi(const FoldBatch& b, int groups, hipStream_t s);
hipError_t launch_reduce_store(const float* src, int n_slabs, size_t slab_stride, int src_ld, int rows, int cols,
                               float* dst, int dst_ld, int transpose, hipStream_t s);
// C[z] (+)= op(A) op(B); A is [M,K] (ta: stored [K,M]), B is [K,N] (tb: stored [N,K]); `splits` partitions K and
// writes split z to C + z * c_split_stride (accumulate must be 0 when splits > 1)
hipError_t launch_sgemm(bool ta, bool tb, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                        int N, int K, int splits, size_t c_split_stride, int accumulate, hipStream_t s);
hipError_t launch_gate_backward_mfma(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                                     const float* de_w, float* partial, int grid, hipStream_t s, const int* rows = nullptr,
                                     const int* n_rows = nullptr);
hipError_t launch_ssp_backward(const float* pre, const float* dact, long n, float* dpre, hipStream_t s);
// train_loss.hip: TargetDiff's forward noising, its two losses with their gradients, and the scatter of those gradients
hipError_t launch_train_noise(const float* x0, const int64_t* v0, const int64_t* t, const int64_t* batch, const uint8_t* gen,
                              int n_lig, int C, const float* acp, const float* log_acp, const float* log_1m_acp, float log_c,
                              const float* eps, const float* u, float* x_t, float* c_t, int64_t* v_t, hipStream_t s);
hipError_t launch_train_loss(const float* x_out, const float* logits, const int64_t* lig_rows, const float* x0,
                             const int64_t* v0, const int64_t* vt, const int64_t* t, const int64_t* batch, const uint8_t* gen,
                             int n_lig, int B, int C, const float* const* tables, float log_c, float* losses, float* x_pred,
                             float* c_pred, float* gpos, float* gz, hipStream_t s);
hipError_t launch_train_loss_bwd(const float* gpos, const float* gz, const int64_t* sort_idx, int n_rec, int n_nodes, int C,
                                 const float* g_pos, const float* g_typ, float* grad_x, float* grad_logits, hipStream_t s);
// train_loss_diffbp.hip: DiffBP's four losses and their gradients with respect to the network outputs (composed row order)
hipError_t launch_diffbp_loss(const float* x_out, const float* x_in, const float* x_stack, const float* logits, const int64_t* sort_idx,
                              const int32_t* graph_ptr, const uint8_t* lig, const float* pos_noise, const float* com_noise,
                              const int64_t* v0, const uint8_t* type_flag, const uint8_t* gen, const int64_t* t, int n_rec, int n_lig,
                              int B, int C, const float* acp, const float* betas, float rho, float gamma, float* gstats, float* losses,
                              float* scal, float* a_pos, float* a_int, float* b_com, float* b_int, float* z_atom, int* bad,
                              hipStream_t s);
hipError_t launch_ssp_backward_rows(const float* pre, const float* dact, const int* rows, const int* n_rows, int max_rows,
                                    float* dpre, hipStream_t s);
hipError_t launch_cls_w1_grad_rows(const float* dlogits, int C, const float* act, const int* rows, const int* n_rows, float* dW1,
                                   hipStream_t s);
hipError_t launch_add_inplace(float* dst, const float* src, long n, hipStream_t s);

}  // namespace cbgx

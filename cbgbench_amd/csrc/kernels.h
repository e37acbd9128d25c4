// internal launcher declarations (host side)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cbgx {

// Auxiliary stream of a caller stream (api.hip: one per (host thread, caller stream), created on first use, reused for the life of
// the thread): the forward's two-stream schedule forks / joins it with `fork` / `join`; the training backward runs the weight-gradient
// part of a block on it and records `done[set]` when the block's buffer set is free again (api_train.hip).  NULL when unavailable.
struct AuxStream {
    hipStream_t owner = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, done[2] = {nullptr, nullptr};
};
AuxStream* aux_for(hipStream_t caller);
typedef AuxStream AuxLane;

// ---- optional per-kernel timing (cbgx_profile_begin / cbgx_profile_end) ----
enum KernelClass { K_KNN = 0, K_GATE, K_NODE_GEMM, K_NODE_QUERY, K_EDGE_X2H, K_EDGE_H2X, K_EDGE_X2H_LISTED, K_EDGE_X2H_BWD, K_EDGE_H2X_BWD,
                   K_TRAIN_GEMM, K_EDGE_X2H_BWD_LISTED, K_EDGE_ROWS_REDUCE, K_NUM_CLASSES };
void profile_mark_begin(int cls, hipStream_t s);
void profile_mark_end(hipStream_t s);
bool profile_is_on();

hipError_t launch_knn(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int32_t* nbr,
                      int32_t* deg, hipStream_t s);
hipError_t launch_gate(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                       float* e_w, hipStream_t s);
hipError_t launch_node_gemm(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M,
                            int nout, int act, hipStream_t s, const int* rows = nullptr, const int* n_rows = nullptr);
// node projection + query fold + fused edge kernel of one attention block.
// x2h: out = h_out[N,128]; h2x: out = x_out[N,3], dx_out optional.
hipError_t launch_attention(bool x2h, const float* att, const float* x, const float* h, const int32_t* nbr,
                            const int32_t* deg, const uint8_t* lig, const uint8_t* gen, const float* e_w, int n_nodes,
                            float* P, float* Qt, float* qbuf, float* out, float* dx_out, const int* act, const int* act_count,
                            const int* src, const int* src_count, hipStream_t s);
// which generation of kernels the stage dispatchers (dispatch.hip) use: always the MFMA one in libcbgx.so; the test-only
// library libcbgx_xcheck.so (-DCBGX_XCHECK) can switch to the first-generation VALU kernels at run time
#ifdef CBGX_XCHECK
extern int g_edge_impl;
hipError_t launch_knn_v1(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int32_t* nbr, int32_t* deg,
                         hipStream_t s);
hipError_t launch_gate_v1(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                          float* e_w, hipStream_t s);
hipError_t launch_node_gemm_v1(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M,
                               int nout, int act, hipStream_t s, const int* rows, const int* n_rows);
hipError_t launch_node_query_v1(const float* att, const float* P, float* Qt, int n_nodes, hipStream_t s);
hipError_t launch_attention_v1(bool x2h, const float* att, const float* x, const float* h, const int32_t* nbr,
                               const int32_t* deg, const uint8_t* lig, const uint8_t* gen, const float* e_w, int n_nodes,
                               float* P, float* Qt, float* out, float* dx_out, hipStream_t s);
#else
constexpr int g_edge_impl = 0;
#endif
// ---- weight packing, batched over attention blocks (the weights are re-packed every training step: 18 blocks x 12 small kernels
// were 216 launches and 1.5 ms per step; one launch per kernel for all blocks is ~20) ----------------------------------------------
constexpr int PACK_BLOCKS_MAX = 20;
struct PackBlocks {                       // per attention block: the reference tensors the pack kernels read and the block's ATT region
    const float* wk0[PACK_BLOCKS_MAX];    // k first Linear [128][340], its bias
    const float* bk0[PACK_BLOCKS_MAX];
    const float* wv0[PACK_BLOCKS_MAX];
    const float* bv0[PACK_BLOCKS_MAX];
    const float* wq0[PACK_BLOCKS_MAX];    // q first Linear [128][128], its bias
    const float* bq0[PACK_BLOCKS_MAX];
    const float* wq1[PACK_BLOCKS_MAX];    // q second Linear [128][128]
    const float* wk1[PACK_BLOCKS_MAX];    // k second Linear [128][128]
    const float* wv1[PACK_BLOCKS_MAX];    // v second Linear ([128][128] x2h, [16][128] h2x)
    float* att[PACK_BLOCKS_MAX];
    // the five 128-column groups of the assembled node projection (PDk | PDv | PSk | PSv | q hidden): base pointer of each group's
    // source matrix -- the CENTRED first Linears inside `att` for the first four, wq0 for the fifth.  A table indexed by the group,
    // not an if-chain over pointers: hipcc 7.2 selected the wrong base for the fifth group in the wave-per-column scale kernel.
    const float* nsrc[PACK_BLOCKS_MAX][5];
    unsigned char x2h[PACK_BLOCKS_MAX];
    int n;
};
// stage 1: centred first Linears of k and v (A_WAKC / A_BAKC / A_WAVC / A_BAVC) of every block
hipError_t launch_pack_stage1(const PackBlocks& pb, hipStream_t s);
// stage 2 (after stage 1 on the same stream): column scales, split-f16 node tables, Wbk fragments, bn2, rbf scales, rbf fragment
// tables (k, v, and the edge-major v table of x2h blocks), dWt, the swizzled Wbv image of x2h blocks
hipError_t launch_pack_stage2(const PackBlocks& pb, hipStream_t s);
// second-generation graph kernels (graph_mfma.hip)
hipError_t launch_pack_gate_img(const float* w1, const float* b1, const float* g, const float* be, const float* w2,
                                float* img, hipStream_t s);
hipError_t launch_knn_reg(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int32_t* nbr,
                          int32_t* deg, hipStream_t s, const int* rows = nullptr, const int* n_rows = nullptr);
// graph-cached calls: the listed centres' lists from (pocket list U the graph's ligand atoms) by rank counting (graph_mfma.hip)
// `s_ew` / `e_w` / `newmask` (optional): kept pocket entries carry their cached gate value to their new rank, newmask[i] = the ranks
// that hold new entries -- what launch_gate_mfma(..., newmask) then evaluates instead of all 32 slots
hipError_t launch_knn_merge(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, const uint8_t* lig,
                            const int32_t* s_nbr, const int32_t* s_deg, int32_t* nbr, int32_t* deg, hipStream_t s, const int* rows,
                            const int* n_rows, const float* s_ew = nullptr, float* e_w = nullptr, unsigned* newmask = nullptr);
// graph-cached calls: the listed centres' merged neighbour lists and their gate values in one launch -- kept pocket entries carry
// their cached gate value to their new rank, the gate MLP runs on the entries that are new (graph_mfma.hip, knn_merge_gate_kernel)
hipError_t launch_knn_merge_gate(const float* packed, const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes,
                                 const uint8_t* lig, const int32_t* s_nbr, const int32_t* s_deg, const float* s_ew, int32_t* nbr,
                                 int32_t* deg, float* e_w, hipStream_t s, const int* rows, const int* n_rows);
hipError_t launch_gate_mfma(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                            float* e_w, hipStream_t s, const int* rows = nullptr, const int* n_rows = nullptr,
                            const unsigned* newmask = nullptr);
// head of a graph-cached forward call in one launch: proximity flags -> `dirty`, their compaction -> `list` / `count` (zero on
// entry), the pocket's graph -> nbr / deg / ew, the cached features of layers 0 / 1 -> out1 / out2 (graph_mfma.hip)
hipError_t launch_graph_cache_begin(const float* x, const int32_t* graph_ptr, int n_graphs, const uint8_t* lig, const float* r32sq,
                                    int n, uint8_t* dirty, int* list, int* count, const int32_t* s_nbr, const int32_t* s_deg,
                                    const float* s_ew, int32_t* nbr, int32_t* deg, float* ew, const float* h1, const float* h2,
                                    float* out1, float* out2, hipStream_t s);
// MFMA node kernels (node_mfma.hip): P = h Wn + bn, q = MLP tail, Qt = folded query
hipError_t launch_pack_node_tables(const PackBlocks& pb, hipStream_t s);     // node_mfma.hip part of stage 2
// counter_zeroed: the caller has already set *count to zero on this stream (cbgx_unitransformer_forward zeroes all its list counters
// with ONE memset instead of one per list: ~15 fills of 4.6 us per denoising step)
hipError_t launch_build_active(const uint8_t* flag, int n, int* list, int* count, hipStream_t s, bool counter_zeroed = false);
hipError_t launch_mark_seed(const uint8_t* a, const uint8_t* b, int n, uint8_t* m, hipStream_t s);
hipError_t launch_mark_from_nbr(const uint8_t* flag, const int32_t* nbr, const int32_t* deg, int n, uint8_t* out,
                                hipStream_t s);
hipError_t launch_mark_nbr(const int* list, const int* count, int n_upper, const int32_t* nbr, const int32_t* deg,
                           uint8_t* m, hipStream_t s);
// large_lists: the destination list (when given) is expected to hold a sizeable part of the nodes (x2h blocks of the cached / pruned
// layers) rather than the few movable atoms of an h2x block: throughput launches instead of one workgroup per column chunk
hipError_t launch_node_mfma(const float* att, const float* h, const uint8_t* lig, int n_nodes, float* P, float* qbuf,
                            float* Qt, const int* act, const int* act_count, const int* src, const int* src_count,
                            hipStream_t s, bool large_lists = false, const int* fold = nullptr, const int* fold_count = nullptr);
// the fused node stage (node_stage_kernel, inputs of <= NODE_STAGE_MAX_ROWS rows) as a list of jobs on the same input features:
// blockIdx.y = job.  `rows` NULL: all n_nodes rows.  `chunk_mask`: 64-column chunks of the projection to produce (CHUNKS_*);
// `proj_only`: phase 1 only (the PS columns of a block's source rows).
struct NodeStageJob {
    const float* att;
    float *P, *q, *Qt;
    const int *rows, *n_rows;
    unsigned chunk_mask;
    int proj_only;
    // optional: the folded query Qt is produced only for rows with fold_flag[row] != 0 -- the x2h edge stage reads it for
    // general-role destinations only (the node or a neighbour is a ligand atom: ~28 % of a pocket), protein-only ones fold in
    // registers (edge_mfma.hip) -- 8 KB of stores per row saved for the others
    const uint8_t* fold_flag;
};
constexpr int NS_JOBS_MAX = 4;
struct NodeStageJobs { NodeStageJob j[NS_JOBS_MAX]; int n; };
constexpr int NODE_STAGE_MAX_ROWS = 8192;   // up to here the latency-built fused kernel replaces the three-kernel chain
// the jobs of one attention block's node stage appended to `jobs` (same selection as launch_node_mfma's fused path): with a
// destination list two jobs (own columns on `act`, PS columns on `src` -- all rows when src is NULL), without one job on all rows
bool add_node_stage_jobs(NodeStageJobs& jobs, const float* att, float* P, float* qbuf, float* Qt, const int* act,
                         const int* act_count, const int* src, const int* src_count, const uint8_t* fold_flag = nullptr);
hipError_t launch_node_stage_jobs(const NodeStageJobs& jobs, const float* h, const uint8_t* lig, int n_nodes, hipStream_t s);
// all node lists of a forward call in four launches (node_mfma.hip): three level kernels over flags + one multi-job compaction
struct GraphFlags {
    const uint8_t *gen, *lig;
    const uint8_t* D1;            // input of the cached levels (may alias d1)
    uint8_t *d1, *a1, *a2, *a3, *D2, *S1, *S2;
};
constexpr int LIST_JOBS_MAX = 20;      // (a cached + pruned + dual call uses 16)
struct ListJobs {
    const uint8_t* flag[LIST_JOBS_MAX];      // member if flag != 0 (NULL: every node) ...
    const uint8_t* flag2[LIST_JOBS_MAX];     // ... and (flag2 != 0) == want2 when flag2 is given
    int want2[LIST_JOBS_MAX];
    int* list[LIST_JOBS_MAX];
    int* count[LIST_JOBS_MAX];
    int n_jobs;
};
hipError_t launch_list_level(const GraphFlags& f, const int32_t* nbr, const int32_t* deg, int n, int level, bool cached, bool prune,
                             hipStream_t s);
// Small inputs (n <= GRAPH_LISTS_MAX_NODES): the three level kernels and the compaction above as ONE launch, one workgroup per graph
// with the graph's flags in LDS (flags only ever propagate along edges, and edges stay inside a graph).  The jobs name their flags by
// id instead of by pointer; of the flag arrays in global memory only the three inputs are read and only d1 is written.
enum GraphFlagId { GF_GEN = 0, GF_LIG, GF_D1IN, GF_d1, GF_a1, GF_a2, GF_a3, GF_D2, GF_S1, GF_S2, GF_COUNT, GF_ALL = -1 };
constexpr int GRAPH_LISTS_MAX_NODES = 8192;
struct GraphListJobs {
    signed char flag[LIST_JOBS_MAX];         // GraphFlagId (GF_ALL: every node) ...
    signed char flag2[LIST_JOBS_MAX];        // ... and (flag2 != 0) == want2 unless GF_ALL
    signed char want2[LIST_JOBS_MAX];
    int* list[LIST_JOBS_MAX];
    int* count[LIST_JOBS_MAX];
    int n_jobs;
};
// `d1_out` [n]: the d1 flags (the node or one of its neighbours is a ligand atom) are also written to global memory
hipError_t launch_graph_lists(const uint8_t* gen, const uint8_t* lig, const uint8_t* d1_in, const int32_t* nbr, const int32_t* deg,
                              const int32_t* graph_ptr, int n_graphs, const GraphListJobs& jobs, bool cached, bool prune,
                              uint8_t* d1_out, hipStream_t s);
hipError_t launch_build_lists(const ListJobs& jobs, int n, hipStream_t s);
hipError_t launch_split_list(const int* list, const int* count, int n, const uint8_t* flag, int* out1, int* cnt1, int* out0,
                             int* cnt0, hipStream_t s, bool counters_zeroed = false);
// MFMA edge kernel (edge_mfma.hip)
int set_edge_workgroup_limit(int n);
hipError_t launch_edge_mfma(bool x2h, const float* att, const float* x, const float* h, const float* P,
                            const float* Qt, const int32_t* nbr, const int32_t* deg, const uint8_t* lig,
                            const uint8_t* gen, const float* e_w, int n_nodes, float* out, float* dx_out,
                            const int* act, const int* act_count, hipStream_t s);
hipError_t launch_edge_x2h_dual(const float* att, const float* x, const float* h, const float* P, const float* Qt,
                                const float* qbuf, const int32_t* nbr, const int32_t* deg, const uint8_t* lig,
                                const uint8_t* gen, const float* e_w, int n_nodes, float* out, const int* list_pp,
                                const int* count_pp, const int* list_gen, const int* count_gen, bool full_layer, hipStream_t s);
// TargetDiff step prologue / epilogue (step.hip)
hipError_t launch_step_prologue(const float* x_lig, const float* c_lig, const int32_t* lig_rows, int n_lig, int C,
                                const float* emb_w, const float* emb_b, const float* ind_w, const float* ind_b, float* x,
                                float* h, hipStream_t s, const int32_t* t_ptr = nullptr);
hipError_t launch_step_epilogue(const float* x_den, const float* logits, const int32_t* lig_rows, const float* x_lig,
                                const float* c_lig, const uint8_t* gen_lig, int n_lig, int C, int t,
                                const float* const* tabs, float log_c, const float* eps, const float* u, float* x_next,
                                float* c_next, int32_t* v_next, hipStream_t s, int32_t* t_ptr = nullptr);
// epilogue of step t + prologue of step t - 1 (the composed rows x[lig_rows], h[lig_rows] of the next denoiser call) in one launch
hipError_t launch_step_boundary(const float* x_den, const float* logits, const int32_t* lig_rows, const float* x_lig,
                                const float* c_lig, const uint8_t* gen_lig, int n_lig, int C, int t, const float* const* tabs,
                                float log_c, const float* eps, const float* u, float* x_next, float* c_next, const float* emb_w,
                                const float* emb_b, const float* ind_w, const float* ind_b, float* x, float* h, hipStream_t s);
hipError_t launch_diffbp_epilogue(const float* x_den, const float* x_com, const float* x_in, const float* logits,
                                  const int32_t* lig_rows, const int32_t* lig_ptr, const float* x_lig, const float* c_lig,
                                  const uint8_t* gen_lig, int n_graphs, int C, int t, int T, const float* acp_tab,
                                  const float* beta_tab, int absorbing, const float* eps, const float* u, float* x_next,
                                  float* c_next, hipStream_t s);
hipError_t launch_diffsbdd_step(const float* x_den, const float* logits, const int32_t* graph_ptr, const int32_t* lig_rows,
                                const int32_t* lig_ptr, const uint8_t* lig_flag, const float* x_lig, const float* c_lig,
                                int n_graphs, int C, float inv_alpha, float coef, float sigma, int do_x, int do_c,
                                const float* eps_x, const float* eps_c, const float* emb_w, const float* emb_b,
                                const float* ind_w, const float* ind_b, float* x_next, float* c_next, float* x, float* h,
                                float* shift_out, float* frame, hipStream_t s);
constexpr int PACK_MAX = 64;
struct PackPiece {
    const float* src; float* dst; int src_ld, src_off, transpose, dst_ld, rows, cols;
};
struct PackBatch {
    PackPiece p[PACK_MAX]; int n;
};
hipError_t launch_pack_copy_multi(const PackBatch& b, hipStream_t s);
hipError_t launch_pack_copy(const float* src, int src_ld, int src_off, int transpose, float* dst, int dst_ld,
                            int rows, int cols, hipStream_t s);

}  // namespace cbgx

// libcbgx C ABI, training half (include/cbgx.h "training" section): taped forward, backward of single
// attention blocks (what the parity tests call) and of the whole denoiser.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>

#include <cstdlib>

#include "../../include/cbgx.h"
#include "kernels.h"
#include "layout.h"
#include "train.h"

using namespace cbgx;

namespace cbgx { int set_error(int code, const char* fmt, ...); }

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return set_error(CBGX_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define RC_TRY(expr)         \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

static inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// Neighbour-row gradients of the x2h edge backward: 0 = fp32 atomics on dP (default: the faster training step), 1 = edge rows + a gather
// in a fixed order (train_scatter.hip): the kernel is 10 % faster (750 vs 832 us per 16.5 k-node launch) and dL/dh becomes reproducible
// bit for bit, but the gather reads N x 32 KB (105 us, HBM-bound) and the step is 2 - 3 % slower (profiles/ab_train_r06[jkmn].log).
// Same mathematics, different summation order; chosen per call by the environment (CBGX_BX_EDGE_ROWS=1), this macro is the default.
#ifndef CBGX_BX_EDGE_ROWS
#define CBGX_BX_EDGE_ROWS 0
#endif
static inline bool edge_rows_mode() {
    const char* e = getenv("CBGX_BX_EDGE_ROWS");
    return e ? atoi(e) != 0 : CBGX_BX_EDGE_ROWS != 0;
}
// schedule-only knobs of the training path, read at every call (a getenv next to a 15 ms step) so that one process can compare them
// (tests/test_gpu_training.py, scripts/ab_train_base.sh)
static inline bool env_on(const char* name) { const char* e = getenv(name); return !e || atoi(e) != 0; }
constexpr int EDGE_GRID = 256;    // persistent workgroups of the edge backward (one per CU: ~97 KB LDS each)
#ifndef CBGX_NODE_GRID
#define CBGX_NODE_GRID 256        // A/B knob (scripts/build_variant.py)
#endif
constexpr int NODE_GRID = CBGX_NODE_GRID;    // persistent workgroups of the node-level reductions (one NS_SIZE slab each)
constexpr int FOLD = 8;           // slab reductions are two-level: n_slabs -> FOLD (slab_fold_kernel) -> 1 (reduce_store)
constexpr int GATE_GRID = 1024;  // gate weight-gradient kernel: 160-thread workgroups, four per CU keep every SIMD busy
constexpr int MAX_SPLITS = 128;   // node groups of the weight-gradient products (one [128 x 640] partial slab each)
constexpr int QLN_SLOTS = 64;     // query-LayerNorm affine gradient accumulators (atomics): one zeroed slot per attention block of a backward

// ---- tape: what the taped forward keeps for the backward ------------------------------------------------
struct Tape {
    int32_t* nbr;
    int32_t* deg;
    float* e_w;
    float* xs;   // [(L+1)][N][3]   xs[l] = coordinates entering layer l
    float* hs;   // [(L+1)][N][128] hs[l] = features entering layer l (hs[l+1] = x2h output = h2x input)
    // node stage of every attention block as the forward computed it (block 2l = x2h of layer l, 2l+1 = h2x): the
    // projection P [N][640] and the folded query Qt [N][16][128].  177 KB per node and layer -- 3.2 GB for a 32-graph
    // batch, a cheap trade on a 288 GB part for not running three node kernels per block again in the backward.
    float* P;    // [2L][N][640]
    float* Qt;   // [2L][N][16][128]
    size_t total;
};

static Tape carve_tape(void* base, int n, int L) {
    Tape t;
    size_t off = 0;
    char* b = (char*)base;
    auto take = [&](size_t bytes) { char* p = b + off; off += align_up(bytes); return p; };
    const size_t N = (size_t)(n > 0 ? n : 1);
    t.nbr = (int32_t*)take(N * KNN * 4);
    t.deg = (int32_t*)take(N * 4);
    t.e_w = (float*)take(N * KNN * 4);
    t.xs = (float*)take((size_t)(L + 1) * N * 3 * 4);
    t.hs = (float*)take((size_t)(L + 1) * N * H * 4);
    t.P = (float*)take((size_t)2 * L * N * PROW * 4);
    t.Qt = (float*)take((size_t)2 * L * N * HEADS * H * 4);
    t.total = off;
    return t;
}

// ---- backward workspace -----------------------------------------------------------------------------------
struct TrainWs {
    float *P, *Qt, *Gt, *T, *S, *gb, *sw, *qs, *dqb, *zb, *dP, *gh, *gx[2], *de_w, *tmp, *partial, *folded, *qln, *nk;
    // an attention block's three slab sets live side by side (edge slabs in `partial`), so that ONE fold launch and ONE
    // reduce-and-store launch per block serve all of them
    float *partial_node, *partial_wgrad, *folded_node, *folded_wgrad;
    int *act, *act_count;
    uint8_t* mask;        // receptive field of the loss, walked backwards (see cbgx_unitransformer_backward)
    int *rf_list[2], *rf_count;
    // round 6, x2h edge backward without neighbour-row atomics (train_scatter.hip, CBGX_BX_EDGE_ROWS=1): one row per edge, and the
    // incoming-edge lists of every source node (built once per backward call)
    float *gate_partial, *gate_folded;     // the gate backward's own slabs: it runs before the auxiliary stream is joined (round 6)
    float* dE;            // [N][32][256]
    int *rin_cnt, *rin_ptr, *rin_tmp, *rin_edge;
    int* lig_list;        // rows with lig_flag (count: rf_count + 32): the classifier head's backward walks it when the loss reads ligand rows only
    // second set of the per-block buffers (round 5): the weight-gradient reductions of a block run on an auxiliary stream while the
    // caller's stream is already in the next block, so consecutive blocks alternate between two sets (attention_block_backward)
    float *T2, *S2, *sw2, *qs2, *dqb2, *zb2, *dP2, *partial2, *folded2, *partial_node2, *partial_wgrad2, *folded_node2, *folded_wgrad2;
    size_t partial_floats;
    size_t total;
};

static size_t partial_floats_needed() {
    size_t a = (size_t)EDGE_GRID * PB_SIZE;
    size_t b = (size_t)NODE_GRID * NS_SIZE;
    size_t c = (size_t)MAX_SPLITS * H * PROW;
    size_t d = (size_t)GATE_GRID * GB_SIZE;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return m > d ? m : d;
}

static TrainWs carve_train(void* base, int n) {
    TrainWs w;
    size_t off = 0;
    char* b = (char*)base;
    auto take = [&](size_t bytes) { char* p = b + off; off += align_up(bytes); return p; };
    const size_t N = (size_t)(n > 0 ? n : 1);
    w.P = (float*)take(N * PROW * 4);
    w.Qt = (float*)take(N * HEADS * H * 4);
    w.Gt = (float*)take(N * HEADS * H * 4);
    w.T = (float*)take(N * HEADS * H * 4);
    w.S = (float*)take(N * HEADS * H * 4);
    w.gb = (float*)take(N * HEADS * 4);
    w.sw = (float*)take(N * HEADS * 4);
    w.qs = (float*)take(N * H * 4);
    w.dqb = (float*)take(N * H * 4);
    w.zb = (float*)take(N * H * 4);
    w.dP = (float*)take(N * PROW * 4 + 256);      // + the x2h edge backward's work counters: zeroed by the same fill
    w.gh = (float*)take(N * H * 4);
    w.gx[0] = (float*)take(N * 3 * 4);
    w.gx[1] = (float*)take(N * 3 * 4);
    w.de_w = (float*)take(N * KNN * 4);
    w.tmp = (float*)take(N * H * 4);
    w.act = (int*)take(N * 4);
    w.act_count = (int*)take(256);
    w.mask = (uint8_t*)take(N);
    w.rf_list[0] = (int*)take(N * 4);
    w.rf_list[1] = (int*)take(N * 4);
    w.rf_count = (int*)take(256);
    w.lig_list = (int*)take(N * 4);
    w.partial_floats = partial_floats_needed();
    w.partial = (float*)take(w.partial_floats * 4);
    w.folded = (float*)take((size_t)FOLD * H * PROW * 4);
    w.qln = (float*)take((size_t)QLN_SLOTS * 2 * H * 4);
    w.partial_node = (float*)take((size_t)NODE_GRID * NS_SIZE * 4);
    w.partial_wgrad = (float*)take((size_t)MAX_SPLITS * H * PROW * 4);
    w.folded_node = (float*)take((size_t)FOLD * NS_SIZE * 4);
    w.folded_wgrad = (float*)take((size_t)FOLD * H * PROW * 4);
    w.nk = (float*)take(BX_NK_FLOATS * 4);     // x2h edge backward: the key path of every wave in flight, parked between two phases
    w.gate_partial = (float*)take((size_t)GATE_GRID * GB_SIZE * 4);
    w.gate_folded = (float*)take((size_t)FOLD * GB_SIZE * 4);
    w.dE = (float*)take(N * KNN * 2 * H * 4);
    w.rin_cnt = (int*)take(N * 4);
    w.rin_ptr = (int*)take((N + 1) * 4);
    w.rin_tmp = (int*)take(N * KNN * 4);
    w.rin_edge = (int*)take(N * KNN * 4);
    w.T2 = (float*)take(N * HEADS * H * 4);
    w.S2 = (float*)take(N * HEADS * H * 4);
    w.sw2 = (float*)take(N * HEADS * 4);
    w.qs2 = (float*)take(N * H * 4);
    w.dqb2 = (float*)take(N * H * 4);
    w.zb2 = (float*)take(N * H * 4);
    w.dP2 = (float*)take(N * PROW * 4 + 256);
    w.partial2 = (float*)take(w.partial_floats * 4);
    w.folded2 = (float*)take((size_t)FOLD * H * PROW * 4);
    w.partial_node2 = (float*)take((size_t)NODE_GRID * NS_SIZE * 4);
    w.partial_wgrad2 = (float*)take((size_t)MAX_SPLITS * H * PROW * 4);
    w.folded_node2 = (float*)take((size_t)FOLD * NS_SIZE * 4);
    w.folded_wgrad2 = (float*)take((size_t)FOLD * H * PROW * 4);
    w.total = off;
    return w;
}

static inline int edge_grid(int n) { return n < EDGE_GRID ? (n > 0 ? n : 1) : EDGE_GRID; }
static inline int node_grid(int n) { return n < NODE_GRID ? (n > 0 ? n : 1) : NODE_GRID; }
// node groups of wgrad_mfma_kernel: at least 128 nodes (8 passes) each
#ifndef CBGX_WGRAD_GROUPS_MAX
#define CBGX_WGRAD_GROUPS_MAX 51      // x 5 column blocks = 255 workgroups, one round on 256 CUs; 16.7 MB of partial slabs instead of 42
                                      // (A/B, scripts/build_variant.py: 128 -> 1944, 102 -> 1965, 64 -> 1957, 51 -> 1968 graph-steps/s)
#endif
static inline int wgrad_groups(int n) {
    int g = (n + 127) / 128;
    return g < 1 ? 1 : (g > CBGX_WGRAD_GROUPS_MAX ? CBGX_WGRAD_GROUPS_MAX : g);
}
static inline int splits_for(int n) {
    int s = (n + 511) / 512;
    return s < 1 ? 1 : (s > MAX_SPLITS ? MAX_SPLITS : s);
}

#define RS(src, nsl, stride, ld, rows, cols, dst, dld, tr) \
    HIP_TRY(launch_reduce_store(src, nsl, stride, ld, rows, cols, dst, dld, tr, s))

// two-level reduction of `n_slabs` partial slabs of `size` floats: returns the FOLD-slab buffer (stride = size) through
// *out / *n_out, or the input itself when it is already small
static int fold_slabs(const float* src, int n_slabs, size_t stride, int size, TrainWs& w, const float** out, int* n_out,
                      size_t* stride_out, hipStream_t s) {
    if (n_slabs <= FOLD) { *out = src; *n_out = n_slabs; *stride_out = stride; return CBGX_OK; }
    HIP_TRY(launch_slab_fold(src, n_slabs, stride, size, FOLD, w.folded, s));
    *out = w.folded; *n_out = FOLD; *stride_out = (size_t)size;
    return CBGX_OK;
}
#define FOLDED(src, nsl, stride, size) \
    const float* fz; int fn; size_t fs; RC_TRY(fold_slabs(src, nsl, stride, size, w, &fz, &fn, &fs, s))

// Backward of one attention block.
//   x2h: g_out = dL/dh_out [N,128];   h2x: g_out = dL/dx_out [N,3] (only gen rows matter; rows = gen list)
// Effects: gh [N,128] += dL/dh_in contributions (x2h: gh must already hold g_out -- the residual path -- and is
// updated in place; h2x: gh += ...), dx [N,3] += coordinate gradients (atomics), de_w += gate gradients,
// grads[18] (k6 v6 q6 tensors, reference layouts) overwritten.
// Round 5: `ov` (optional) -- the weight-gradient part of the block (the node-level outer products and column sums, the dense
// projection's weight gradient, the slab folds and the reduce-and-store of all 18 tensors: ~135 us of small kernels per block,
// 2 ms of a 16 ms training step, none of it needed by the next block) runs on an auxiliary stream behind a fork event recorded
// after the query backward, while the caller's stream goes on with dgrad and the next block.  Consecutive blocks alternate between
// two sets of the buffers those kernels read (T, S, sw, qs, dqb, zb, dP and the slab sets); before a block touches its set it waits
// for the auxiliary work of the block that used the set before it (two blocks earlier -- which also covers g_out: the x2h block two
// blocks later is the first to overwrite the gradient buffer an x2h block's outer products read).
struct BlockOverlap {
    AuxLane* aux;       // auxiliary stream + events of the caller's stream (api.hip); NULL: everything on the caller's stream
    int next_set;       // set of the next block
    bool used[2];       // done[set] has been recorded
    bool joined = false;
    // Every exit path joins the auxiliary stream: an RC_TRY / HIP_TRY return from inside the layer loop must not leave it writing
    // gradients and slab buffers that the caller -- or the next call on this workspace, which starts with used = {false, false} and
    // would wait on nothing -- reuses.  The success path joins with stream-side waits (cbgx_unitransformer_backward, below) and sets
    // `joined`; a failure path blocks the host here until the auxiliary stream has drained.
    ~BlockOverlap() {
        if (aux && !joined && (used[0] || used[1])) (void)hipStreamSynchronize(aux->s);
    }
};

static int attention_block_backward(bool x2h, const float* att, const float* x, const float* h_in, const float* g_out,
                                    const int32_t* nbr, const int32_t* deg, const uint8_t* lig, const float* e_w,
                                    const int* rows, const int* n_rows, int n, TrainWs& w_all, float* gh, float* dx,
                                    float* de_w, float* const* grads, hipStream_t s, const float* P_saved = nullptr,
                                    const float* Qt_saved = nullptr, float* qln = nullptr, const float* gh_src = nullptr,
                                    const int* dp_rows = nullptr, const int* dp_n_rows = nullptr, BlockOverlap* ov = nullptr,
                                    bool rin_ready = false, const float* dx_init = nullptr) {
    // `dx_init` (h2x blocks of a layer loop): dx = dx_init (the identity path of x_{l+1} = x_l + ...) before the block adds to it
    TrainWs w = w_all;      // this block's view of the workspace: the per-block buffers of its set
    const int set = (ov && ov->aux) ? ov->next_set : 0;
    if (set) {
        w.T = w_all.T2; w.S = w_all.S2; w.sw = w_all.sw2; w.qs = w_all.qs2; w.dqb = w_all.dqb2; w.zb = w_all.zb2; w.dP = w_all.dP2;
        w.partial = w_all.partial2; w.folded = w_all.folded2; w.partial_node = w_all.partial_node2;
        w.partial_wgrad = w_all.partial_wgrad2; w.folded_node = w_all.folded_node2; w.folded_wgrad = w_all.folded_wgrad2;
    }
    if (ov && ov->aux) {
        if (ov->used[set]) HIP_TRY(hipStreamWaitEvent(s, ov->aux->done[set], 0));    // the set's previous user has finished with it
        ov->next_set = set ^ 1;
    }
    // `dp_rows` / `dp_n_rows` (h2x blocks): a device-side list that contains every row of dP this block can touch (the listed nodes
    // and their neighbours); the dense products and column sums over dP then walk the list instead of all N rows
    // `qln` [2][128]: zeroed accumulator of the query LayerNorm's affine gradients (nullptr: w.qln, zeroed here);
    // `gh_src`: gh = gh_src + (this block's contribution) instead of gh += (the caller then needs no snapshot of g_out == gh_src)
    // x2h blocks run the one-wave-per-node backward (8 nodes in flight per workgroup); h2x blocks and, in libcbgx_xcheck.so,
    // cbgx_debug_set_edge_kernel(2) the second-generation workgroup-per-node kernel
    const bool gen3 = x2h && g_edge_impl == 0;
    // (gen3: 8 nodes in flight per workgroup; a grid that is a multiple of 8 switches on its XCD-aware node partition)
    int eg = gen3 ? edge_grid((n + 7) / 8) : edge_grid(n);
    // with the weight-gradient kernels on the auxiliary stream (BlockOverlap), the x2h edge backward may leave a few compute units to
    // them: it takes a CU whole, so on 256 workgroups nothing runs next to it (CBGX_BX_GRID: its largest grid; an experiment knob)
    static const int bx_grid = [] { const char* e = getenv("CBGX_BX_GRID"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= EDGE_GRID ? v : EDGE_GRID; }();
    if (gen3 && ov && ov->aux && eg > bx_grid) eg = bx_grid;
    if (gen3 && eg >= 64) eg &= ~7;
    const int ng = node_grid(n);
    // recompute the node stage of the forward: the MFMA node kernels (centred projection; own columns and query fold only
    // for the listed rows) with the MFMA edge backward, the first-generation ones with the VALU cross-check kernel
    // (skipped when the taped forward left its own P / Qt: P_saved, Qt_saved)
    const bool mfma = g_edge_impl != 1;
    const bool saved = P_saved && Qt_saved;
    const float* Pn = saved ? P_saved : w.P;
    const float* Qn = saved ? Qt_saved : w.Qt;
    if (!saved) {
#ifdef CBGX_XCHECK
        if (!mfma) {
            HIP_TRY(launch_node_gemm_v1(h_in, H, att + A_WN, att + A_BN, w.P, PROW, n, PROW, 0, s, nullptr, nullptr));
            HIP_TRY(launch_node_query_v1(att, w.P, w.Qt, n, s));
        } else
#endif
            HIP_TRY(launch_node_mfma(att, h_in, lig, n, w.P, w.qs, w.Qt, rows, n_rows, nullptr, nullptr, s, x2h));
    }
    if (x2h) HIP_TRY(launch_fold_grad(att, g_out, n, w.Gt, w.gb, s, mfma ? rows : nullptr, mfma ? n_rows : nullptr));      // (a listed block reads listed rows' folds only)
    // Edge rows (CBGX_BX_EDGE_ROWS=1, round 6): a full launch of the one-wave-per-node kernel writes d pre of every edge to the edge's own
    // row of w.dE and launch_edge_rows_reduce gathers the rows of every source node in a fixed order -- no atomics on dP.  Every column
    // of dP then has exactly one writer (PD: the edge kernel's plain stores, PS: the gather, q hidden: the query backward), so only the
    // work counters behind it are zeroed.  `rin_ready`: the caller has built the incoming-edge lists (w_all.rin_ptr / rin_edge).
    const bool er = gen3 && !rows && rin_ready;
    if (er) HIP_TRY(hipMemsetAsync(w.dP + (size_t)n * PROW, 0, 256, s));
    else if (!x2h && dp_rows && mfma && ov && ov->aux && env_on("CBGX_TRAIN_ZERO_ROWS"))
        // h2x block in the layer loop: every reader of dP walks `dp_rows`, so only those rows are zeroed (a fill of all N rows was 42 MB
        // per block; +0.4 % on the training line, profiles/ab_train_r06r.log)
    {
        HIP_TRY(launch_zero_rows(w.dP, PROW, dp_rows, dp_n_rows, n, s, dx_init ? dx : nullptr, dx_init, n * 3));
        dx_init = nullptr;
    } else HIP_TRY(hipMemsetAsync(w.dP, 0, (size_t)n * PROW * sizeof(float) + 256, s));     // (and the work counters behind it)
    if (dx_init) HIP_TRY(hipMemcpyAsync(dx, dx_init, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!qln) {
        qln = w.qln;
        HIP_TRY(hipMemsetAsync(qln, 0, 2 * H * sizeof(float), s));
    }
    // libcbgx_xcheck.so only: cbgx_debug_set_edge_kernel(1) selects the first-generation (VALU) backward kernels as an
    // on-device cross-check
#ifdef CBGX_XCHECK
    if (!mfma)
        HIP_TRY(launch_edge_backward(x2h, att, x, Pn, Qn, w.Gt, w.gb, g_out, nbr, deg, lig, e_w, rows, n_rows, n, w.T,
                                     w.S, w.sw, w.dP, dx, de_w, w.partial, eg, s));
    else
#endif
    if (gen3) {
        HIP_TRY(launch_edge_backward_x2h(att, x, Pn, Qn, w.Gt, w.gb, nbr, deg, lig, e_w, rows, n_rows, n, w.T, w.S, w.sw,
                                         w.dP, dx, de_w, w.partial, w.nk, reinterpret_cast<int*>(w.dP + (size_t)n * PROW), eg, s,
                                         er ? w.dE : nullptr));
        if (er) HIP_TRY(launch_edge_rows_reduce(w.dE, w.rin_ptr, w.rin_edge, n, w.dP, s));
    } else
        HIP_TRY(launch_edge_backward_mfma(x2h, att, x, Pn, Qn, w.Gt, w.gb, g_out, nbr, deg, lig, e_w, rows, n_rows, n,
                                          w.T, w.S, w.sw, w.dP, dx, de_w, w.partial, eg, s, 1));
    float *k0w = grads[0], *k0b = grads[1], *kg = grads[2], *kb = grads[3], *k1w = grads[4], *k1b = grads[5];
    float *v0w = grads[6], *v0b = grads[7], *vg = grads[8], *vb = grads[9], *v1w = grads[10], *v1b = grads[11];
    float *q0w = grads[12], *q0b = grads[13], *qg = grads[14], *qb = grads[15], *q1w = grads[16], *q1b = grads[17];
    // The block's three slab sets (edge kernel, node-level reductions, dense projection) are reduced at its end: one first-level
    // fold launch for the sets with more than FOLD slabs, one reduce-and-store launch for every gradient tensor of the block
    // (six launches and three 512-byte fills per block before; the sums are the same, slab by slab).
    RsBatch rb;
    rb.n = 0;
    FoldBatch fb;
    fb.n = 0;
    bool batch_overflow = false;      // (a stack struct passed by value to a kernel must not be overrun by one more piece)
    auto piece = [&](const float* src, int nsl, size_t stride, int ld, int rws, int cls, float* dst, int dld, int tr) {
        if (rb.n >= RS_MAX) { batch_overflow = true; return; }
        rb.p[rb.n++] = RsPiece{src, dst, stride, nsl, ld, rws, cls, dld, tr};
    };
    const float* fz; int fn; size_t fs;
    auto folded = [&](const float* src, int nsl, size_t stride, int size, float* dst) {
        if (nsl <= FOLD) { fz = src; fn = nsl; fs = stride; return; }
        if (fb.n >= FOLD_JOBS_MAX) { batch_overflow = true; fz = src; fn = nsl; fs = stride; return; }
        fb.j[fb.n++] = FoldJob{src, dst, stride, nsl, size};
        fz = dst; fn = FOLD; fs = (size_t)size;
    };
    {   // edge-indexed weight gradients: type / rbf columns of the first Linears, LayerNorm affine
        folded(w.partial, eg, PB_SIZE, PB_SIZE, w.folded);
        piece(fz + PB_WT, fn, fs, 2 * H, NT, H, k0w, KV_IN, 1);
        piece(fz + PB_WT + H, fn, fs, 2 * H, NT, H, v0w, KV_IN, 1);
        piece(fz + PB_WR, fn, fs, 2 * H, NT * G, H, k0w + NT, KV_IN, 1);
        piece(fz + PB_WR + H, fn, fs, 2 * H, NT * G, H, v0w + NT, KV_IN, 1);
        piece(fz + PB_LNG, fn, fs, H, 1, H, kg, H, 0);
        piece(fz + PB_LNG + H, fn, fs, H, 1, H, vg, H, 0);
        piece(fz + PB_LNB, fn, fs, H, 1, H, kb, H, 0);
        piece(fz + PB_LNB + H, fn, fs, H, 1, H, vb, H, 0);
        if (!x2h) {
            piece(fz + PB_WBV16, fn, fs, H, HEADS, H, v1w, H, 0);
            piece(fz + PB_BBV16, fn, fs, HEADS, 1, HEADS, v1b, HEADS, 0);
        }
    }
    // query MLP backward (fills dP[:, 512:640]); its LayerNorm affine gradients are accumulated into qln by atomics
    {
        const int tiles = (n + 15) / 16, qgrid = tiles < 1024 ? tiles : 1024;   // 25 KB of LDS, four waves per SIMD: four workgroups per CU
#ifdef CBGX_XCHECK
        if (!mfma) HIP_TRY(launch_q_backward(att, Pn, w.T, rows, n_rows, n, w.qs, w.dqb, w.zb, w.dP, qln, qgrid, s));
        else
#endif
            HIP_TRY(launch_q_backward_mfma(att, Pn, w.T, rows, n_rows, n, w.qs, w.dqb, w.zb, w.dP, qln, qgrid, s));
    }
    // ---- from here to the reduce-and-store: weight gradients only.  With `ov` they go to the auxiliary stream (`s` is shadowed) and
    // dgrad -- what the next block waits for -- is issued on the caller's stream first.
    hipStream_t s_main = s;
    const bool deferred = ov && ov->aux;
    if (deferred) {
        HIP_TRY(hipEventRecord(ov->aux->fork, s_main));
        HIP_TRY(hipStreamWaitEvent(ov->aux->s, ov->aux->fork, 0));
        if (mfma)
            HIP_TRY(launch_dgrad_mfma(w.dP, PROW, att + A_WN, PROW, gh, H, n, PROW, 1, s_main, gh_src, gh_src ? nullptr : dp_rows,
                                      gh_src ? nullptr : dp_n_rows));
        else {
            if (gh_src && gh_src != gh) HIP_TRY(hipMemcpyAsync(gh, gh_src, (size_t)n * H * 4, hipMemcpyDeviceToDevice, s_main));
            HIP_TRY(launch_sgemm(false, true, w.dP, PROW, att + A_WN, PROW, gh, H, n, H, PROW, 1, 0, 1, s_main));
        }
        s = ov->aux->s;
    }
    // node-level reductions, all into one slab per workgroup (NS_* layout), folded and scattered once:
    //   second Linears: dWbk = sum_i (q_i / sqrt 8) (x) T_i ;  x2h: dWbv = sum_i G_i (x) S_i ;  dWq1 = sum_i dq_i (x) z_i
    //   biases: second v / q Linears, and the first Linears = column sums of dP (k | v | - | - | q hidden)
#ifdef CBGX_XCHECK
    const auto outer = mfma ? launch_outer_accum_mfma : launch_outer_accum;
#else
    const auto outer = launch_outer_accum_mfma;
#endif
    float* pn = w.partial_node;
    HIP_TRY(outer(true, w.qs, w.T, rows, n_rows, n, pn + NS_WBK, NS_SIZE, ng, s));
    if (x2h) {
        HIP_TRY(outer(true, g_out, w.S, rows, n_rows, n, pn + NS_WBV, NS_SIZE, ng, s));
        HIP_TRY(launch_colsum(g_out, H, H, w.sw, rows, n_rows, n, pn + NS_V1B, NS_SIZE, ng, s));
    }
    HIP_TRY(outer(false, w.dqb, w.zb, rows, n_rows, n, pn + NS_WQ1, NS_SIZE, ng, s));
    HIP_TRY(launch_colsum(w.dqb, H, H, nullptr, rows, n_rows, n, pn + NS_Q1B, NS_SIZE, ng, s));
    HIP_TRY(launch_colsum(w.dP, PROW, PROW, nullptr, dp_rows, dp_n_rows, n, pn + NS_DP, NS_SIZE, ng, s));
    {
        folded(pn, ng, NS_SIZE, NS_SIZE, w.folded_node);
        piece(fz + NS_WBK, fn, fs, H, H, H, k1w, H, 0);
        piece(fz + NS_WQ1, fn, fs, H, H, H, q1w, H, 0);
        piece(fz + NS_Q1B, fn, fs, H, 1, H, q1b, H, 0);
        piece(fz + NS_DP, fn, fs, PROW, 1, H, k0b, H, 0);
        piece(fz + NS_DP + H, fn, fs, PROW, 1, H, v0b, H, 0);
        piece(fz + NS_DP + 4 * H, fn, fs, PROW, 1, H, q0b, H, 0);
        if (x2h) {
            piece(fz + NS_WBV, fn, fs, H, H, H, v1w, H, 0);
            piece(fz + NS_V1B, fn, fs, H, 1, H, v1b, H, 0);
        }
        piece(qln, 1, 0, H, 1, H, qg, H, 0);
        piece(qln + H, 1, 0, H, 1, H, qb, H, 0);
        piece(qln, 0, 0, H, 1, H, k1b, H, 0);      // a sum over zero slabs: the key bias cancels in the softmax, its gradient is 0
    }
    // dense projection: dWn[k][n] = sum_i h_in[i][k] dP[i][n]  -> h_dst / h_src / q columns of the first Linears
    const int sp = mfma ? wgrad_groups(n) : (splits_for(n) > 32 ? 32 : splits_for(n));
    if (mfma)
        HIP_TRY(launch_wgrad_mfma(h_in, H, w.dP, PROW, n, PROW / H, w.partial_wgrad, PROW, (size_t)H * PROW, sp, s, dp_rows,
                                  dp_n_rows));
    else
        HIP_TRY(launch_sgemm(true, false, h_in, H, w.dP, PROW, w.partial_wgrad, PROW, H, PROW, n, sp, (size_t)H * PROW, 0, s));
    {
        folded(w.partial_wgrad, sp, (size_t)H * PROW, H * PROW, w.folded_wgrad);
        piece(fz + 0 * H, fn, fs, PROW, H, H, k0w + NT + NT * G, KV_IN, 1);
        piece(fz + 1 * H, fn, fs, PROW, H, H, v0w + NT + NT * G, KV_IN, 1);
        piece(fz + 2 * H, fn, fs, PROW, H, H, k0w + NT + NT * G + H, KV_IN, 1);
        piece(fz + 3 * H, fn, fs, PROW, H, H, v0w + NT + NT * G + H, KV_IN, 1);
        piece(fz + 4 * H, fn, fs, PROW, H, H, q0w, H, 1);
    }
    if (batch_overflow) return cbgx::set_error(CBGX_E_INVALID, "backward: reduce batch overflow (RS_MAX / FOLD_JOBS_MAX too small)");
    HIP_TRY(launch_slab_fold_multi(fb, FOLD, s));
    HIP_TRY(launch_reduce_store_multi(rb, s));
    if (deferred) {
        HIP_TRY(hipEventRecord(ov->aux->done[set], s));
        ov->used[set] = true;
        return CBGX_OK;       // (dgrad was issued above, on the caller's stream)
    }
    // dL/dh_in = (gh_src or gh itself) + dP Wn^T
    if (mfma)
        HIP_TRY(launch_dgrad_mfma(w.dP, PROW, att + A_WN, PROW, gh, H, n, PROW, 1, s, gh_src, gh_src ? nullptr : dp_rows,
                                  gh_src ? nullptr : dp_n_rows));
    else {
        if (gh_src && gh_src != gh) HIP_TRY(hipMemcpyAsync(gh, gh_src, (size_t)n * H * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(launch_sgemm(false, true, w.dP, PROW, att + A_WN, PROW, gh, H, n, H, PROW, 1, 0, 1, s));
    }
    return CBGX_OK;
}

// backward of the distance gate: weight gradients from the accumulated dL/de_w, on the workspace's gate slabs (no buffer of a block set)
static int gate_backward(const float* packed, const float* xs, const int32_t* nbr, const int32_t* deg, int n, TrainWs& w_all,
                         float* const* grads, hipStream_t s, const int* rows = nullptr, const int* n_rows = nullptr) {
    TrainWs w = w_all;
    w.folded = w_all.gate_folded;
    HIP_TRY(launch_gate_backward_mfma(packed, xs, nbr, deg, n, w.de_w, w.gate_partial, GATE_GRID, s, rows, n_rows));
    FOLDED(w.gate_partial, GATE_GRID, GB_SIZE, GB_SIZE);
    RS(fz + GB_W1, fn, fs, G, GH, G, grads[0], G, 0);
    RS(fz + GB_B1, fn, fs, GH, 1, GH, grads[1], GH, 0);
    RS(fz + GB_LNG, fn, fs, GH, 1, GH, grads[2], GH, 0);
    RS(fz + GB_LNB, fn, fs, GH, 1, GH, grads[3], GH, 0);
    RS(fz + GB_W2, fn, fs, GH, 1, GH, grads[4], GH, 0);
    RS(fz + GB_B2, fn, fs, 1, 1, 1, grads[5], 1, 0);
    return CBGX_OK;
}

static int check_grads(float* const* g, int count, const char* who) {
    if (!g) return set_error(CBGX_E_INVALID, "%s: grads is NULL", who);
    for (int i = 0; i < count; ++i)
        if (!g[i]) return set_error(CBGX_E_INVALID, "%s: grads[%d] is NULL", who, i);
    return CBGX_OK;
}

extern "C" {

size_t cbgx_train_tape_bytes(int n_nodes, int num_layers) {
    if (n_nodes < 0 || num_layers < 1) return 0;
    return carve_tape(nullptr, n_nodes, num_layers).total;
}

size_t cbgx_train_workspace_bytes(int n_nodes) {
    if (n_nodes < 0) return 0;
    return carve_train(nullptr, n_nodes).total;
}

int cbgx_unitransformer_forward_train(const float* packed, int num_layers, int num_classes, const float* x,
                                      const float* h, const int32_t* graph_ptr, const uint8_t* lig_flag,
                                      const uint8_t* gen_flag, int n_nodes, int n_graphs, float* x_out, float* h_out,
                                      float* logits, void* tape, size_t tape_bytes, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return cbgx_unitransformer_forward_train_ex(packed, num_layers, num_classes, x, h, graph_ptr, lig_flag, gen_flag, n_nodes, n_graphs,
                                                x_out, h_out, logits, 0u, tape, tape_bytes, workspace, workspace_bytes, stream);
}

int cbgx_unitransformer_forward_train_ex(const float* packed, int num_layers, int num_classes, const float* x,
                                         const float* h, const int32_t* graph_ptr, const uint8_t* lig_flag,
                                         const uint8_t* gen_flag, int n_nodes, int n_graphs, float* x_out, float* h_out,
                                         float* logits, unsigned flags, void* tape, size_t tape_bytes, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (flags & ~CBGX_FWD_H_ON_SOURCES) return set_error(CBGX_E_INVALID, "forward_train: unknown flags 0x%x", flags);
    if (n_nodes < 0 || n_graphs < 0 || num_layers < 1) return set_error(CBGX_E_INVALID, "forward_train: bad sizes");
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !x || !h || !graph_ptr || !lig_flag || !gen_flag || !x_out || !tape || !workspace)
        return set_error(CBGX_E_INVALID, "forward_train: NULL pointer");
    Tape tp = carve_tape(tape, n_nodes, num_layers);
    if (tape_bytes < tp.total) return set_error(CBGX_E_WORKSPACE, "forward_train: tape %zu < %zu", tape_bytes, tp.total);
    TrainWs w = carve_train(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return set_error(CBGX_E_WORKSPACE, "forward_train: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    const size_t nx = (size_t)n_nodes * 3, nh = (size_t)n_nodes * H;
    HIP_TRY(hipMemcpyAsync(tp.xs, x, nx * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(tp.hs, h, nh * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(launch_knn(x, graph_ptr, n_graphs, n_nodes, tp.nbr, tp.deg, s));
    HIP_TRY(launch_gate(packed, x, tp.nbr, tp.deg, n_nodes, tp.e_w, s));
    HIP_TRY(launch_build_active(gen_flag, n_nodes, w.act, w.act_count, s));
    // Round 6: the node stage of the x2h block of layer l + 1 (projection, query MLP, fold: ~57 us at 16.5 k nodes) reads h_{l+1} only,
    // like the h2x block of layer l (~60 us) -- it runs on the caller's auxiliary stream next to that block (what the sampling forward
    // has done since round 3); every block owns its P / Qt on the tape, the two node stages use different query scratch rows.
    // +0.7 % on the training line (profiles/ab_train_r06r.log).  CBGX_TRAIN_OVERLAP=0 / CBGX_TRAIN_FWD_OVERLAP=0, the per-kernel
    // profile and the VALU cross-check kernels keep one stream.
    AuxLane* aux = (env_on("CBGX_TRAIN_OVERLAP") && env_on("CBGX_TRAIN_FWD_OVERLAP") && g_edge_impl == 0 && !profile_is_on() &&
                    num_layers > 1) ? aux_for(s) : nullptr;
    struct Joiner {     // a failure between fork and join must not leave the auxiliary stream writing the tape
        AuxLane* a; bool out = false;
        ~Joiner() { if (a && out) (void)hipStreamSynchronize(a->s); }
    } jn{aux};
    // Round 6, h_out == NULL (the caller does not read h': TargetDiff / DiffSBDD training, whose backward then comes with
    // grad_h_out == NULL and prunes the same way): the receptive-field pruning of the sampling forward.  The outputs that remain are
    // x_out and the logits of ligand rows, so the last x2h block only has to produce h' on A1 = gen | lig | nbr(gen), the one before it
    // on A2 = A1 | nbr(A1) (its sources: A3 = A2 | nbr(A2)); h2x blocks need the neighbour projection on nbr(gen) only (subset of A1).
    // Rows outside those sets are zero on the tape (the pruned backward's dense products over h walk all rows with zero weights).
    // (CBGX_FWD_H_ON_SOURCES: h_out is wanted, but on A1 only -- DiffBP's centre-of-mass head reads it on the movable atoms and their
    // neighbours; the backward then prunes around the support of its dL/dh_out, which lies inside A1, unless CBGX_TRAIN_PRUNE_GH=0)
    const bool fprune = (!h_out || ((flags & CBGX_FWD_H_ON_SOURCES) && env_on("CBGX_TRAIN_PRUNE_GH"))) && num_layers >= 3 &&
                        g_edge_impl != 1 && env_on("CBGX_TRAIN_FWD_PRUNE") && (!h_out || !logits || num_classes <= 128);
    const int* A[3] = {w.rf_list[0], w.rf_list[1], w.lig_list};
    const int* An[3] = {w.rf_count, w.rf_count + 16, w.rf_count + 32};
    if (fprune) {
        HIP_TRY(launch_mark_seed(gen_flag, lig_flag, n_nodes, w.mask, s));
        HIP_TRY(launch_mark_nbr(w.act, w.act_count, n_nodes, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n_nodes, w.rf_list[0], w.rf_count, s));
        HIP_TRY(launch_mark_nbr(w.rf_list[0], w.rf_count, n_nodes, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n_nodes, w.rf_list[1], w.rf_count + 16, s));
        HIP_TRY(launch_mark_nbr(w.rf_list[1], w.rf_count + 16, n_nodes, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n_nodes, w.lig_list, w.rf_count + 32, s));
    }
    // (destination list, its count, source list, its count) of the x2h block of layer l
    auto x2h_lists = [&](int l, const int*& d, const int*& dn, const int*& sr, const int*& sn) {
        d = dn = sr = sn = nullptr;
        const int k = num_layers - 1 - l;      // 0 for the last layer
        if (fprune && k < 2) { d = A[k]; dn = An[k]; sr = A[k + 1]; sn = An[k + 1]; }
    };
    for (int l = 0; l < num_layers; ++l) {
        const float* xc = tp.xs + (size_t)l * nx;
        const float* hc = tp.hs + (size_t)l * nh;
        float* xn = tp.xs + (size_t)(l + 1) * nx;
        float* hn = tp.hs + (size_t)(l + 1) * nh;
        float* Px = tp.P + (size_t)(2 * l) * n_nodes * PROW;
        float* Qx = tp.Qt + (size_t)(2 * l) * n_nodes * HEADS * H;
        const int *d, *dn, *sr, *sn;
        x2h_lists(l, d, dn, sr, sn);
        if (d) HIP_TRY(hipMemsetAsync(hn, 0, nh * 4, s));      // rows the listed launch does not write
        if (!aux) {
            HIP_TRY(launch_attention(true, packed + x2h_off(l), xc, hc, tp.nbr, tp.deg, lig_flag, gen_flag, tp.e_w, n_nodes,
                                     Px, Qx, w.qs, hn, nullptr, d, dn, sr, sn, s));
        } else {
            if (l == 0) HIP_TRY(launch_node_mfma(packed + x2h_off(0), hc, lig_flag, n_nodes, Px, w.qs2, Qx, d, dn, sr, sn, s, true));
            else { HIP_TRY(hipStreamWaitEvent(s, aux->join, 0)); jn.out = false; }      // this layer's node stage (auxiliary stream)
            HIP_TRY(launch_edge_mfma(true, packed + x2h_off(l), xc, hc, Px, Qx, tp.nbr, tp.deg, lig_flag, gen_flag, tp.e_w, n_nodes,
                                     hn, nullptr, d, dn, s));
            if (l + 1 < num_layers) {
                const int *d2, *d2n, *s2, *s2n;
                x2h_lists(l + 1, d2, d2n, s2, s2n);
                HIP_TRY(hipEventRecord(aux->fork, s));
                HIP_TRY(hipStreamWaitEvent(aux->s, aux->fork, 0));
                jn.out = true;
                HIP_TRY(launch_node_mfma(packed + x2h_off(l + 1), hn, lig_flag, n_nodes, tp.P + (size_t)(2 * l + 2) * n_nodes * PROW,
                                         w.qs2, tp.Qt + (size_t)(2 * l + 2) * n_nodes * HEADS * H, d2, d2n, s2, s2n, aux->s, true));
                HIP_TRY(hipEventRecord(aux->join, aux->s));
            }
        }
        HIP_TRY(launch_attention(false, packed + h2x_off(l), xc, hn, tp.nbr, tp.deg, lig_flag, gen_flag, tp.e_w, n_nodes,
                                 Px + (size_t)n_nodes * PROW, Qx + (size_t)n_nodes * HEADS * H, w.qs, xn, nullptr, w.act,
                                 w.act_count, fprune ? A[0] : nullptr, fprune ? An[0] : nullptr, s));
    }
    const float* hl = tp.hs + (size_t)num_layers * nh;
    HIP_TRY(hipMemcpyAsync(x_out, tp.xs + (size_t)num_layers * nx, nx * 4, hipMemcpyDeviceToDevice, s));
    if (h_out) HIP_TRY(hipMemcpyAsync(h_out, hl, nh * 4, hipMemcpyDeviceToDevice, s));
    if (logits) {
        if (num_classes < 1) return set_error(CBGX_E_INVALID, "forward_train: num_classes=%d", num_classes);
        const float* c = packed + cls_off(num_layers);
        // (pruned: the logits of the A1 rows -- the ligand rows are among them; other rows of `logits` are not written)
        HIP_TRY(launch_node_gemm(hl, H, c + C_W0T, c + C_B0, w.P, H, n_nodes, H, 1, s, fprune ? A[0] : nullptr, fprune ? An[0] : nullptr));
        HIP_TRY(launch_node_gemm(w.P, H, c + C_W1T, c + cls_b1(num_classes), logits, num_classes, n_nodes, num_classes,
                                 0, s, fprune ? A[0] : nullptr, fprune ? An[0] : nullptr));
    }
    return CBGX_OK;
}

int cbgx_x2h_attention_backward(const float* packed, int layer, const float* x, const float* h, const int32_t* nbr,
                                const int32_t* deg, const uint8_t* lig_flag, const float* e_w, int n_nodes,
                                const float* grad_h_out, float* grad_h, float* grad_x, float* grad_e_w,
                                float* const* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_nodes <= 0 || layer < 0) return set_error(CBGX_E_INVALID, "x2h_backward: bad sizes");
    if (!packed || !x || !h || !nbr || !deg || !lig_flag || !e_w || !grad_h_out || !grad_h || !grad_x || !grad_e_w ||
        !workspace)
        return set_error(CBGX_E_INVALID, "x2h_backward: NULL pointer");
    RC_TRY(check_grads(grads, 18, "x2h_backward"));
    TrainWs w = carve_train(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return set_error(CBGX_E_WORKSPACE, "x2h_backward: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(grad_h, grad_h_out, (size_t)n_nodes * H * 4, hipMemcpyDeviceToDevice, s));   // residual
    HIP_TRY(hipMemsetAsync(grad_x, 0, (size_t)n_nodes * 3 * 4, s));
    HIP_TRY(hipMemsetAsync(grad_e_w, 0, (size_t)n_nodes * KNN * 4, s));
    const bool rin = edge_rows_mode() && g_edge_impl == 0;
    if (rin) HIP_TRY(launch_rin_build(nbr, deg, n_nodes, w.rin_cnt, w.rin_ptr, w.rin_tmp, w.rin_edge, s));
    return attention_block_backward(true, packed + x2h_off(layer), x, h, grad_h_out, nbr, deg, lig_flag, e_w, nullptr,
                                    nullptr, n_nodes, w, grad_h, grad_x, grad_e_w, grads, s, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, rin);
}

int cbgx_h2x_attention_backward(const float* packed, int layer, const float* x, const float* h, const int32_t* nbr,
                                const int32_t* deg, const uint8_t* lig_flag, const uint8_t* gen_flag, const float* e_w,
                                int n_nodes, const float* grad_x_out, float* grad_h, float* grad_x, float* grad_e_w,
                                float* const* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_nodes <= 0 || layer < 0) return set_error(CBGX_E_INVALID, "h2x_backward: bad sizes");
    if (!packed || !x || !h || !nbr || !deg || !lig_flag || !gen_flag || !e_w || !grad_x_out || !grad_h || !grad_x ||
        !grad_e_w || !workspace)
        return set_error(CBGX_E_INVALID, "h2x_backward: NULL pointer");
    RC_TRY(check_grads(grads, 18, "h2x_backward"));
    TrainWs w = carve_train(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return set_error(CBGX_E_WORKSPACE, "h2x_backward: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_build_active(gen_flag, n_nodes, w.act, w.act_count, s));
    HIP_TRY(hipMemsetAsync(grad_h, 0, (size_t)n_nodes * H * 4, s));
    HIP_TRY(hipMemcpyAsync(grad_x, grad_x_out, (size_t)n_nodes * 3 * 4, hipMemcpyDeviceToDevice, s));   // x_out = x + ...
    HIP_TRY(hipMemsetAsync(grad_e_w, 0, (size_t)n_nodes * KNN * 4, s));
    return attention_block_backward(false, packed + h2x_off(layer), x, h, grad_x_out, nbr, deg, lig_flag, e_w, w.act,
                                    w.act_count, n_nodes, w, grad_h, grad_x, grad_e_w, grads, s);
}

int cbgx_unitransformer_backward(const float* packed, int num_layers, int num_classes, const void* tape,
                                 size_t tape_bytes, const uint8_t* lig_flag, const uint8_t* gen_flag, int n_nodes,
                                 const float* grad_x_out, const float* grad_h_out, const float* grad_logits,
                                 float* const* grads, int num_grads, float* grad_h_in, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (n_nodes <= 0 || num_layers < 1 || num_classes < 1) return set_error(CBGX_E_INVALID, "backward: bad sizes");
    if (!packed || !tape || !lig_flag || !gen_flag || !workspace) return set_error(CBGX_E_INVALID, "backward: NULL pointer");
    if (num_grads != 6 + 36 * num_layers + 4)
        return set_error(CBGX_E_INVALID, "backward: expected %d gradient tensors, got %d", 6 + 36 * num_layers + 4, num_grads);
    RC_TRY(check_grads(grads, num_grads, "backward"));
    Tape tp = carve_tape((void*)tape, n_nodes, num_layers);
    if (tape_bytes < tp.total) return set_error(CBGX_E_WORKSPACE, "backward: tape %zu < %zu", tape_bytes, tp.total);
    TrainWs w = carve_train(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return set_error(CBGX_E_WORKSPACE, "backward: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    const int n = n_nodes, L = num_layers, C = num_classes;
    const size_t nx = (size_t)n * 3, nh = (size_t)n * H;
    const int ng = node_grid(n), sp = splits_for(n);

    // dL/dh_L: the caller's gradient plus the classifier head
    if (grad_h_out) HIP_TRY(hipMemcpyAsync(w.gh, grad_h_out, nh * 4, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemsetAsync(w.gh, 0, nh * 4, s));
    float* const* cg = grads + 6 + 36 * L;
    // (round 6: with a caller gradient on h_out the same listed head runs on the SUPPORT of grad_logits -- the rows with a non-zero
    // entry, marked on the device; DiffBP's losses read the logits of ligand rows only as well.  CBGX_TRAIN_PRUNE_GH=0: the dense head)
    const bool head_listed = grad_logits && g_edge_impl != 1 && C <= 128 && (grad_h_out == nullptr || env_on("CBGX_TRAIN_PRUNE_GH"));
    if (head_listed) {
        // The caller's promise behind grad_h_out == NULL (include/cbgx.h): its loss reads the logits on lig_flag rows only, so
        // grad_logits is zero elsewhere and the head's backward walks the ligand rows (a twentieth of the nodes) instead of all
        // of them: listed node GEMMs for the recompute, listed weight- / input-gradient products, nothing read outside the list.
        const float* c = packed + cls_off(L);
        const float* hl = tp.hs + (size_t)L * nh;
        float* pre = w.qs;    // [N,128] scratch: listed rows only
        float* act = w.zb;
        float* dact = w.dqb;
        const int* ll = w.lig_list;
        const int* lc = w.rf_count + 32;
        if (grad_h_out) {       // no promise about grad_logits: its support
            HIP_TRY(launch_mark_nonzero_rows(grad_logits, n, w.mask, s, C, 1));
            HIP_TRY(launch_build_active(w.mask, n, w.lig_list, w.rf_count + 32, s));
        } else
            HIP_TRY(launch_build_active(lig_flag, n, w.lig_list, w.rf_count + 32, s));
        HIP_TRY(launch_node_gemm(hl, H, c + C_W0T, c + C_B0, pre, H, n, H, 0, s, ll, lc));
        HIP_TRY(launch_node_gemm(hl, H, c + C_W0T, c + C_B0, act, H, n, H, 1, s, ll, lc));
        // classifier.2: dW1[c][k] = sum_i dlogits[i][c] act[i][k];  db1 = colsum(dlogits)
        HIP_TRY(launch_cls_w1_grad_rows(grad_logits, C, act, ll, lc, cg[2], s));
        HIP_TRY(launch_colsum(grad_logits, C, C, nullptr, ll, lc, n, w.partial, C, ng, s));
        { FOLDED(w.partial, ng, C, C); RS(fz, fn, fs, C, 1, C, cg[3], C, 0); }
        // d(act) = dlogits W1 (all rows: a [N x C] x [C x 128] product, zero outside the list);  d(pre) = d(act) sigmoid(pre) on the list
        HIP_TRY(launch_sgemm(false, true, grad_logits, C, c + C_W1T, C, dact, H, n, H, C, 1, 0, 0, s));
        HIP_TRY(launch_ssp_backward_rows(pre, dact, ll, lc, n, w.tmp, s));
        // classifier.0: dW0[n][k] = sum_i dpre[i][n] h[i][k];  db0 = colsum(dpre);  dh += dpre W0
        const int wg = wgrad_groups(n);
        HIP_TRY(launch_wgrad_mfma(w.tmp, H, hl, H, n, 1, w.partial, H, (size_t)H * H, wg, s, ll, lc));
        { FOLDED(w.partial, wg, (size_t)H * H, H * H); RS(fz, fn, fs, H, H, H, cg[0], H, 0); }
        HIP_TRY(launch_colsum(w.tmp, H, H, nullptr, ll, lc, n, w.partial, H, ng, s));
        { FOLDED(w.partial, ng, H, H); RS(fz, fn, fs, H, 1, H, cg[1], H, 0); }
        HIP_TRY(launch_dgrad_mfma(w.tmp, H, c + C_W0T, H, w.gh, H, n, H, 1, s, nullptr, ll, lc));
    } else if (grad_logits) {
        const float* c = packed + cls_off(L);
        const float* hl = tp.hs + (size_t)L * nh;
        float* pre = w.qs;    // [N,128] scratch
        float* act = w.zb;
        float* dact = w.dqb;
        HIP_TRY(launch_node_gemm(hl, H, c + C_W0T, c + C_B0, pre, H, n, H, 0, s));
        HIP_TRY(launch_node_gemm(hl, H, c + C_W0T, c + C_B0, act, H, n, H, 1, s));
        // classifier.2: dW1[c][k] = sum_i dlogits[i][c] act[i][k];  db1 = colsum(dlogits)
        HIP_TRY(launch_sgemm(true, false, grad_logits, C, act, H, w.partial, H, C, H, n, sp, (size_t)C * H, 0, s));
        { FOLDED(w.partial, sp, (size_t)C * H, C * H); RS(fz, fn, fs, H, C, H, cg[2], H, 0); }
        HIP_TRY(launch_colsum(grad_logits, C, C, nullptr, nullptr, nullptr, n, w.partial, C, ng, s));
        { FOLDED(w.partial, ng, C, C); RS(fz, fn, fs, C, 1, C, cg[3], C, 0); }
        // d(act) = dlogits W1  (C_W1T is [128][C]);  d(pre) = d(act) sigmoid(pre)
        HIP_TRY(launch_sgemm(false, true, grad_logits, C, c + C_W1T, C, dact, H, n, H, C, 1, 0, 0, s));
        HIP_TRY(launch_ssp_backward(pre, dact, (long)nh, w.tmp, s));
        // classifier.0: dW0[n][k] = sum_i dpre[i][n] h[i][k];  db0 = colsum(dpre);  dh += dpre W0
        if (g_edge_impl != 1) {
            const int wg = wgrad_groups(n);
            HIP_TRY(launch_wgrad_mfma(w.tmp, H, hl, H, n, 1, w.partial, H, (size_t)H * H, wg, s));
            FOLDED(w.partial, wg, (size_t)H * H, H * H); RS(fz, fn, fs, H, H, H, cg[0], H, 0);
        } else {
            HIP_TRY(launch_sgemm(true, false, w.tmp, H, hl, H, w.partial, H, H, H, n, sp, (size_t)H * H, 0, s));
            FOLDED(w.partial, sp, (size_t)H * H, H * H); RS(fz, fn, fs, H, H, H, cg[0], H, 0);
        }
        HIP_TRY(launch_colsum(w.tmp, H, H, nullptr, nullptr, nullptr, n, w.partial, H, ng, s));
        { FOLDED(w.partial, ng, H, H); RS(fz, fn, fs, H, 1, H, cg[1], H, 0); }
        if (g_edge_impl != 1)
            HIP_TRY(launch_dgrad_mfma(w.tmp, H, c + C_W0T, H, w.gh, H, n, H, 1, s));
        else
            HIP_TRY(launch_sgemm(false, true, w.tmp, H, c + C_W0T, H, w.gh, H, n, H, H, 1, 0, 1, s));
    } else {
        HIP_TRY(hipMemsetAsync(cg[0], 0, (size_t)H * H * 4, s));
        HIP_TRY(hipMemsetAsync(cg[1], 0, H * 4, s));
        HIP_TRY(hipMemsetAsync(cg[2], 0, (size_t)C * H * 4, s));
        HIP_TRY(hipMemsetAsync(cg[3], 0, C * 4, s));
    }
    int cur = 0;
    if (grad_x_out) HIP_TRY(hipMemcpyAsync(w.gx[cur], grad_x_out, nx * 4, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemsetAsync(w.gx[cur], 0, nx * 4, s));
    HIP_TRY(hipMemsetAsync(w.de_w, 0, (size_t)n * KNN * 4, s));
    HIP_TRY(launch_build_active(gen_flag, n, w.act, w.act_count, s));
    // Receptive-field pruning, the mirror image of the forward's: without a caller gradient on h_out, dL/dh_L is non-zero
    // on ligand rows only (classifier), the last h2x block adds rows of A1 = gen | lig | nbr(gen), and the x2h block of a
    // layer spreads its input gradient one hop further (A2 = A1 | nbr(A1)).  Rows outside carry an exactly zero gradient
    // and contribute exactly zero to everything, so the last two x2h blocks are walked on A1 / A2 only.
    // Round 6: with a caller gradient on h_out the same pruning holds around ITS support (the rows with a non-zero entry, marked on
    // the device): DiffBP's centre-of-mass head reads h_out on the movable atoms and their neighbours only, and until now its
    // training ran all nine x2h blocks -- and the dense products of all nine h2x blocks -- on every row (CBGX_TRAIN_PRUNE_GH=0: as before).
    const bool prune = (grad_h_out == nullptr || (env_on("CBGX_TRAIN_PRUNE_GH") && (!grad_logits || C <= 128))) && L >= 3;
    if (prune) {
        HIP_TRY(launch_mark_seed(gen_flag, lig_flag, n, w.mask, s));
        if (grad_h_out) {       // the seed: every row where dL/dh_L can be non-zero (no promise about the caller's two gradients)
            HIP_TRY(launch_mark_nonzero_rows(grad_h_out, n, w.mask, s));
            if (grad_logits && C <= 128) HIP_TRY(launch_mark_nonzero_rows(grad_logits, n, w.mask, s, C, 0));
        }
        HIP_TRY(launch_mark_nbr(w.act, w.act_count, n, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n, w.rf_list[0], w.rf_count, s));
        HIP_TRY(launch_mark_nbr(w.rf_list[0], w.rf_count, n, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n, w.rf_list[1], w.rf_count + 16, s));
    }

    // one zeroed accumulator per attention block for the query LayerNorm's affine gradients (atomics): a single fill here
    // instead of one per block
    const bool qln_slots = 2 * L <= QLN_SLOTS;
    if (qln_slots) HIP_TRY(hipMemsetAsync(w.qln, 0, (size_t)2 * L * 2 * H * sizeof(float), s));
    float* gh_cur = w.gh;       // dL/dh of the layer boundary being crossed; the x2h blocks write the other buffer (no snapshot copy)
    float* gh_oth = w.tmp;
    // weight-gradient work of every block on the caller's auxiliary stream (attention_block_backward, BlockOverlap): off while the
    // per-kernel profile runs (its sections are timed on one stream), with the VALU cross-check kernels, or by CBGX_TRAIN_OVERLAP=0
    const bool overlap_env = env_on("CBGX_TRAIN_OVERLAP");
    // (and only with one query-LayerNorm accumulator slot per block: the shared fallback slot would be refilled by the next block
    // while the auxiliary stream still reads it)
    BlockOverlap ov{(overlap_env && g_edge_impl != 1 && !profile_is_on() && qln_slots) ? aux_for(s) : nullptr, 0, {false, false}};
    // edge-row mode: incoming-edge lists of every source node, once for all layers
    const bool rin = edge_rows_mode() && g_edge_impl == 0 && (!prune || L > 2);
    if (rin) HIP_TRY(launch_rin_build(tp.nbr, tp.deg, n, w.rin_cnt, w.rin_ptr, w.rin_tmp, w.rin_edge, s));
    for (int l = L - 1; l >= 0; --l) {
        const float* xl = tp.xs + (size_t)l * nx;
        const float* h_in = tp.hs + (size_t)l * nh;
        const float* h_mid = tp.hs + (size_t)(l + 1) * nh;
        float* const* g = grads + 6 + 36 * l;
        // x_{l+1} = x_l + gen * H2X(x_l, h_mid): identity path first, then the block's own contributions
        const int nxt = cur ^ 1;
        const float* Px = tp.P + (size_t)(2 * l) * n * PROW;
        const float* Qx = tp.Qt + (size_t)(2 * l) * n * HEADS * H;
        RC_TRY(attention_block_backward(false, packed + h2x_off(l), xl, h_mid, w.gx[cur], tp.nbr, tp.deg, lig_flag, tp.e_w,
                                        w.act, w.act_count, n, w, gh_cur, w.gx[nxt], w.de_w, g + 18, s,
                                        Px + (size_t)n * PROW, Qx + (size_t)n * HEADS * H,
                                        qln_slots ? w.qln + (size_t)(2 * l + 1) * 2 * H : nullptr, nullptr,
                                        // dP of an h2x block is non-zero on gen | nbr(gen) only, a subset of the receptive-field list A1
                                        prune ? w.rf_list[0] : nullptr, prune ? w.rf_count : nullptr, &ov, false, w.gx[cur]));
        // h_mid = h_in + X2H(x_l, h_in): gh_cur holds dL/dh_mid, which is also the residual part of dL/dh_in.  The block reads it
        // (fold, outer products, bias sums) and writes dL/dh_in = gh_cur + dP Wn^T into the OTHER buffer.
        const int k = L - 1 - l;      // 0 for the last layer
        const int* rows = (prune && k < 2) ? w.rf_list[k] : nullptr;
        const int* n_rows = (prune && k < 2) ? w.rf_count + 16 * k : nullptr;
        RC_TRY(attention_block_backward(true, packed + x2h_off(l), xl, h_in, gh_cur, tp.nbr, tp.deg, lig_flag, tp.e_w,
                                        rows, n_rows, n, w, gh_oth, w.gx[nxt], w.de_w, g, s, Px, Qx,
                                        qln_slots ? w.qln + (size_t)(2 * l) * 2 * H : nullptr, gh_cur, nullptr, nullptr, &ov, rin));
        { float* t = gh_cur; gh_cur = gh_oth; gh_oth = t; }
        cur = nxt;
    }
    if (grad_h_in) HIP_TRY(hipMemcpyAsync(grad_h_in, gh_cur, nh * 4, hipMemcpyDeviceToDevice, s));
    // distance gate (computed once from the input coordinates, used by all 2L blocks): on its own slab buffers, BEFORE the auxiliary
    // stream is joined -- the last blocks' weight-gradient kernels (~150 us) run beside it instead of ahead of it
    RC_TRY(gate_backward(packed, tp.xs, tp.nbr, tp.deg, n, w, grads, s));
    // the auxiliary stream's last weight gradients must be in place when this call's work on `s` is: join both sets
    if (ov.aux)
        for (int k = 0; k < 2; ++k)
            if (ov.used[k]) HIP_TRY(hipStreamWaitEvent(s, ov.aux->done[k], 0));
    ov.joined = true;
    return CBGX_OK;
}

// ---- a stack of H2X blocks on its own graph (DiffBP's CoMPredictor, diffbp.py:30-101): taped forward + backward -------
size_t cbgx_h2x_stack_tape_bytes(int n_nodes, int num_layers) {
    if (n_nodes < 0 || num_layers < 1) return 0;
    const size_t N = (size_t)(n_nodes > 0 ? n_nodes : 1);
    return align_up(N * KNN * 4) + align_up(N * 4) + align_up(N * KNN * 4) + align_up((size_t)(num_layers + 1) * N * 3 * 4);
}

struct StackTape { int32_t* nbr; int32_t* deg; float* e_w; float* xs; };
static StackTape carve_stack_tape(void* base, int n) {
    StackTape t;
    char* b = (char*)base;
    size_t off = 0;
    const size_t N = (size_t)(n > 0 ? n : 1);
    t.nbr = (int32_t*)(b + off); off += align_up(N * KNN * 4);
    t.deg = (int32_t*)(b + off); off += align_up(N * 4);
    t.e_w = (float*)(b + off); off += align_up(N * KNN * 4);
    t.xs = (float*)(b + off);
    return t;
}

int cbgx_h2x_stack_forward_train(const float* packed, int num_layers, const float* x, const float* h,
                                 const int32_t* graph_ptr, const uint8_t* lig_flag, const uint8_t* gen_flag, int n_nodes,
                                 int n_graphs, float* x_out, void* tape, size_t tape_bytes, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (n_nodes <= 0 || n_graphs < 1 || num_layers < 1) return set_error(CBGX_E_INVALID, "h2x_stack_forward_train: bad sizes");
    if (!packed || !x || !h || !graph_ptr || !lig_flag || !gen_flag || !x_out || !tape || !workspace)
        return set_error(CBGX_E_INVALID, "h2x_stack_forward_train: NULL pointer");
    if (tape_bytes < cbgx_h2x_stack_tape_bytes(n_nodes, num_layers))
        return set_error(CBGX_E_WORKSPACE, "h2x_stack_forward_train: tape too small");
    TrainWs w = carve_train(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return set_error(CBGX_E_WORKSPACE, "h2x_stack_forward_train: workspace %zu < %zu", workspace_bytes, w.total);
    StackTape tp = carve_stack_tape(tape, n_nodes);
    hipStream_t s = (hipStream_t)stream;
    const size_t nx = (size_t)n_nodes * 3;
    HIP_TRY(hipMemcpyAsync(tp.xs, x, nx * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(launch_build_active(gen_flag, n_nodes, w.act, w.act_count, s));
    // As cbgx_h2x_stack_forward (api.hip): an h2x block only reads the neighbour lists and gate values of the rows it moves and the
    // projections of their in-neighbours, so the kNN search, the gate MLP and the neighbour projection run on those lists (round 6;
    // the backward walks the same lists -- CBGX_TRAIN_STACK_LISTS=0 switches both back to all rows)
    const bool lists = g_edge_impl == 0 && env_on("CBGX_TRAIN_STACK_LISTS");
    const int *src = nullptr, *src_n = nullptr;
    if (lists) {
        HIP_TRY(launch_knn_reg(x, graph_ptr, n_graphs, n_nodes, tp.nbr, tp.deg, s, w.act, w.act_count));
        HIP_TRY(launch_gate_mfma(packed, x, tp.nbr, tp.deg, n_nodes, tp.e_w, s, w.act, w.act_count));
        HIP_TRY(launch_mark_seed(gen_flag, gen_flag, n_nodes, w.mask, s));
        HIP_TRY(launch_mark_nbr(w.act, w.act_count, n_nodes, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n_nodes, w.rf_list[0], w.rf_count, s));
        src = w.rf_list[0]; src_n = w.rf_count;
    } else {
        HIP_TRY(launch_knn(x, graph_ptr, n_graphs, n_nodes, tp.nbr, tp.deg, s));
        HIP_TRY(launch_gate(packed, x, tp.nbr, tp.deg, n_nodes, tp.e_w, s));
    }
    for (int l = 0; l < num_layers; ++l)
        HIP_TRY(launch_attention(false, packed + GATE_SIZE + (size_t)l * ATT_SIZE, tp.xs + (size_t)l * nx, h, tp.nbr, tp.deg,
                                 lig_flag, gen_flag, tp.e_w, n_nodes, w.P, w.Qt, w.qs, tp.xs + (size_t)(l + 1) * nx, nullptr,
                                 w.act, w.act_count, src, src_n, s));
    HIP_TRY(hipMemcpyAsync(x_out, tp.xs + (size_t)num_layers * nx, nx * 4, hipMemcpyDeviceToDevice, s));
    return CBGX_OK;
}

int cbgx_h2x_stack_backward(const float* packed, int num_layers, const void* tape, size_t tape_bytes, const float* h,
                            const uint8_t* lig_flag, const uint8_t* gen_flag, int n_nodes, const float* grad_x_out,
                            float* const* grads, int num_grads, float* grad_h, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (n_nodes <= 0 || num_layers < 1) return set_error(CBGX_E_INVALID, "h2x_stack_backward: bad sizes");
    if (!packed || !tape || !h || !lig_flag || !gen_flag || !grad_x_out || !grad_h || !workspace)
        return set_error(CBGX_E_INVALID, "h2x_stack_backward: NULL pointer");
    if (num_grads != 6 + 18 * num_layers)
        return set_error(CBGX_E_INVALID, "h2x_stack_backward: expected %d gradient tensors, got %d", 6 + 18 * num_layers, num_grads);
    RC_TRY(check_grads(grads, num_grads, "h2x_stack_backward"));
    if (tape_bytes < cbgx_h2x_stack_tape_bytes(n_nodes, num_layers))
        return set_error(CBGX_E_WORKSPACE, "h2x_stack_backward: tape too small");
    TrainWs w = carve_train(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return set_error(CBGX_E_WORKSPACE, "h2x_stack_backward: workspace %zu < %zu", workspace_bytes, w.total);
    StackTape tp = carve_stack_tape((void*)tape, n_nodes);
    hipStream_t s = (hipStream_t)stream;
    const int n = n_nodes;
    const size_t nx = (size_t)n * 3, nh = (size_t)n * H;
    HIP_TRY(hipMemsetAsync(w.gh, 0, nh * 4, s));
    int cur = 0;
    HIP_TRY(hipMemcpyAsync(w.gx[cur], grad_x_out, nx * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(w.de_w, 0, (size_t)n * KNN * 4, s));
    HIP_TRY(launch_build_active(gen_flag, n, w.act, w.act_count, s));
    // Round 6, as the denoiser's layer loop: the projection gradient of an h2x block is non-zero on gen | nbr(gen) only (this stack's
    // own graph), so its fill and every dense product over it walk that list; the weight-gradient kernels of a block run on the caller's
    // auxiliary stream on alternating buffer sets, one query-LayerNorm accumulator slot per block (CBGX_TRAIN_OVERLAP=0: one stream)
    const bool lists = g_edge_impl != 1 && env_on("CBGX_TRAIN_STACK_LISTS");
    if (lists) {
        HIP_TRY(launch_mark_seed(gen_flag, gen_flag, n, w.mask, s));
        HIP_TRY(launch_mark_nbr(w.act, w.act_count, n, tp.nbr, tp.deg, w.mask, s));
        HIP_TRY(launch_build_active(w.mask, n, w.rf_list[0], w.rf_count, s));
    }
    const bool qln_slots = num_layers <= QLN_SLOTS;
    if (qln_slots) HIP_TRY(hipMemsetAsync(w.qln, 0, (size_t)num_layers * 2 * H * sizeof(float), s));
    BlockOverlap ov{(lists && env_on("CBGX_TRAIN_OVERLAP") && !profile_is_on() && qln_slots) ? aux_for(s) : nullptr, 0, {false, false}};
    for (int l = num_layers - 1; l >= 0; --l) {
        const int nxt = cur ^ 1;
        // x_out = x + gen * delta: the identity part of the coordinate gradient is the block's `dx_init`
        RC_TRY(attention_block_backward(false, packed + GATE_SIZE + (size_t)l * ATT_SIZE, tp.xs + (size_t)l * nx, h, w.gx[cur],
                                        tp.nbr, tp.deg, lig_flag, tp.e_w, w.act, w.act_count, n, w, w.gh, w.gx[nxt], w.de_w,
                                        grads + 6 + 18 * l, s, nullptr, nullptr,
                                        qln_slots ? w.qln + (size_t)l * 2 * H : nullptr, nullptr,
                                        lists ? w.rf_list[0] : nullptr, lists ? w.rf_count : nullptr, &ov, false, w.gx[cur]));
        cur = nxt;
    }
    HIP_TRY(hipMemcpyAsync(grad_h, w.gh, nh * 4, hipMemcpyDeviceToDevice, s));
    // (the stack's blocks add to de_w on the movable rows only: the gate backward walks that list -- 16.5 k rows -> ~900)
    RC_TRY(gate_backward(packed, tp.xs, tp.nbr, tp.deg, n, w, grads, s, lists ? w.act : nullptr, lists ? w.act_count : nullptr));
    if (ov.aux)
        for (int k = 0; k < 2; ++k)
            if (ov.used[k]) HIP_TRY(hipStreamWaitEvent(s, ov.aux->done[k], 0));
    ov.joined = true;
    return CBGX_OK;
}

// ---- TargetDiff's training arithmetic around the denoiser (train_loss.hip) ---------------------------------------------------------
constexpr int LOSS_MAX_GRAPHS = 4096;     // per-graph sums of the loss kernel live in LDS (3 floats per graph)

int cbgx_targetdiff_train_noise(const float* x0, const int64_t* v0, const int64_t* t, const int64_t* batch, const uint8_t* gen,
                                int n_lig, int num_classes, const float* alphas_cumprod, const float* log_alphas_cumprod,
                                const float* log_one_minus_alphas_cumprod, const float* eps, const float* u, float* x_t,
                                float* c_t, int64_t* v_t, void* stream) {
    if (n_lig == 0) return CBGX_OK;
    if (n_lig < 0 || num_classes < 1 || num_classes > 32)
        return set_error(CBGX_E_INVALID, "train_noise: bad sizes (n_lig=%d C=%d)", n_lig, num_classes);
    if (!x0 || !v0 || !t || !batch || !gen || !alphas_cumprod || !log_alphas_cumprod || !log_one_minus_alphas_cumprod || !eps ||
        !u || !x_t || !c_t || !v_t)
        return set_error(CBGX_E_INVALID, "train_noise: NULL pointer");
    HIP_TRY(launch_train_noise(x0, v0, t, batch, gen, n_lig, num_classes, alphas_cumprod, log_alphas_cumprod,
                               log_one_minus_alphas_cumprod, (float)log((double)num_classes), eps, u, x_t, c_t, v_t,
                               (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_targetdiff_loss(const float* x_out, const float* logits, const int64_t* lig_rows, const float* x0, const int64_t* v0,
                         const int64_t* v_t, const int64_t* t, const int64_t* batch, const uint8_t* gen, int n_lig, int n_graphs,
                         int num_classes, const float* const* tables, float* losses, float* x_pred, float* c_pred,
                         float* grad_pos, float* grad_logit, void* stream) {
    if (n_lig <= 0 || n_graphs <= 0 || n_graphs > LOSS_MAX_GRAPHS || num_classes < 1 || num_classes > 32)
        return set_error(CBGX_E_INVALID, "targetdiff_loss: bad sizes (n_lig=%d B=%d (max %d) C=%d)", n_lig, n_graphs,
                         LOSS_MAX_GRAPHS, num_classes);
    if (!x_out || !logits || !lig_rows || !x0 || !v0 || !v_t || !t || !batch || !gen || !tables || !losses || !grad_pos ||
        !grad_logit)
        return set_error(CBGX_E_INVALID, "targetdiff_loss: NULL pointer");
    for (int i = 0; i < 4; ++i)
        if (!tables[i]) return set_error(CBGX_E_INVALID, "targetdiff_loss: table %d is NULL", i);
    HIP_TRY(launch_train_loss(x_out, logits, lig_rows, x0, v0, v_t, t, batch, gen, n_lig, n_graphs, num_classes, tables,
                              (float)log((double)num_classes), losses, x_pred, c_pred, grad_pos, grad_logit,
                              (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_diffbp_loss(const float* x_out, const float* x_in, const float* x_stack, const float* logits, const int64_t* sort_idx,
                     const int32_t* graph_ptr, const uint8_t* lig_flag, const float* pos_noise, const float* com_noise,
                     const int64_t* v0, const uint8_t* type_flag, const uint8_t* gen, const int64_t* t, int n_protein, int n_lig,
                     int n_graphs, int num_classes, const float* alphas_cumprod, const float* betas, float rho, float gamma,
                     float* losses, float* scal, float* gstats, float* a_pos, float* a_int, float* b_com, float* b_int, float* z_atom,
                     int32_t* bad, void* stream) {
    if (n_lig <= 0 || n_graphs <= 0 || n_protein < 0 || num_classes < 1 || num_classes > 32)
        return set_error(CBGX_E_INVALID, "diffbp_loss: bad sizes (n_lig=%d B=%d C=%d)", n_lig, n_graphs, num_classes);
    if (!x_out || !x_in || !x_stack || !logits || !sort_idx || !graph_ptr || !lig_flag || !pos_noise || !com_noise || !v0 || !type_flag ||
        !gen || !t || !alphas_cumprod || !betas || !losses || !scal || !gstats || !a_pos || !a_int || !b_com || !b_int || !z_atom || !bad)
        return set_error(CBGX_E_INVALID, "diffbp_loss: NULL pointer");
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int32_t), (hipStream_t)stream));
    HIP_TRY(launch_diffbp_loss(x_out, x_in, x_stack, logits, sort_idx, graph_ptr, lig_flag, pos_noise, com_noise, v0, type_flag, gen, t,
                               n_protein, n_lig, n_graphs, num_classes, alphas_cumprod, betas, rho, gamma, gstats, losses, scal, a_pos,
                               a_int, b_com, b_int, z_atom, bad, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_diffsbdd_train_noise(const float* x0, const float* x_protein, const int64_t* v0, const float* eps_x, const float* eps_c,
                              const uint8_t* gen, const int64_t* t, const int64_t* sort_idx, const int32_t* graph_ptr, int n_protein,
                              int n_lig, int n_graphs, int num_classes, const float* alpha_table, const float* sigma_table,
                              int num_timesteps, float* x_t, float* x_protein_t, float* c_t, float* gdata, void* stream) {
    if (n_lig <= 0 || n_graphs <= 0 || n_protein < 0 || num_classes < 1 || num_classes > 32 || num_timesteps < 1)
        return set_error(CBGX_E_INVALID, "diffsbdd_train_noise: bad sizes (n_lig=%d B=%d C=%d T=%d)", n_lig, n_graphs, num_classes,
                         num_timesteps);
    if (!x0 || (n_protein && (!x_protein || !x_protein_t)) || !v0 || !eps_x || !eps_c || !gen || !t || !sort_idx || !graph_ptr ||
        !alpha_table || !sigma_table || !x_t || !c_t || !gdata)
        return set_error(CBGX_E_INVALID, "diffsbdd_train_noise: NULL pointer");
    HIP_TRY(launch_diffsbdd_noise(x0, x_protein, v0, eps_x, eps_c, gen, t, sort_idx, graph_ptr, n_protein, n_graphs, num_classes,
                                  alpha_table, sigma_table, num_timesteps, x_t, x_protein_t, c_t, gdata, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_diffsbdd_loss(const float* x_out, const float* logits, const float* eps_x, const float* eps_c, const int64_t* t,
                       const int64_t* sort_idx, const int32_t* graph_ptr, int n_protein, int n_lig, int n_graphs, int num_classes,
                       const float* gdata, float* glosses, float* losses, float* x_pred, float* c_pred, float* grad_pos,
                       float* grad_logit, void* stream) {
    if (n_lig <= 0 || n_graphs <= 0 || n_protein < 0 || num_classes < 1 || num_classes > 32)
        return set_error(CBGX_E_INVALID, "diffsbdd_loss: bad sizes (n_lig=%d B=%d C=%d)", n_lig, n_graphs, num_classes);
    if (!x_out || !logits || !eps_x || !eps_c || !t || !sort_idx || !graph_ptr || !gdata || !glosses || !losses || !x_pred || !c_pred ||
        !grad_pos || !grad_logit)
        return set_error(CBGX_E_INVALID, "diffsbdd_loss: NULL pointer");
    HIP_TRY(launch_diffsbdd_loss(x_out, logits, eps_x, eps_c, t, sort_idx, graph_ptr, n_protein, n_graphs, num_classes, gdata, glosses,
                                 losses, x_pred, c_pred, grad_pos, grad_logit, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_compose_plan(const int64_t* batch_protein, const int64_t* batch_ligand, int n_protein, int n_ligand, int n_graphs,
                      int32_t* scratch, int64_t* sort_idx, int64_t* batch_idx, uint8_t* lig_flag, int64_t* lig_rows, int32_t* graph_ptr,
                      void* stream) {
    if (n_protein < 0 || n_ligand < 0 || n_graphs < 1 || n_graphs > (1 << 20))
        return set_error(CBGX_E_INVALID, "compose_plan: bad sizes (N_protein=%d N_ligand=%d B=%d)", n_protein, n_ligand, n_graphs);
    if (!scratch || !graph_ptr || (n_protein && !batch_protein) || (n_ligand && (!batch_ligand || !lig_rows)) ||
        ((n_protein + n_ligand) && (!sort_idx || !batch_idx || !lig_flag)))
        return set_error(CBGX_E_INVALID, "compose_plan: NULL pointer");
    HIP_TRY(launch_compose_plan(batch_protein, batch_ligand, n_protein, n_ligand, n_graphs, scratch, sort_idx, batch_idx, lig_flag,
                                lig_rows, graph_ptr, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_embed_compose(const float* x_protein, const float* x_ligand, const float* protein_feat, const int64_t* protein_aa,
                       const float* ligand_feat, const int64_t* sort_idx, const uint8_t* gen_protein, const uint8_t* gen_ligand,
                       int n_protein, int n_ligand, int feat_dim, int num_aa, int lig_dim, const float* const* params, float* x, float* h,
                       float* ext, uint8_t* gen_flag, void* stream) {
    if (n_protein < 0 || n_ligand < 0 || feat_dim < 1 || num_aa < 1 || lig_dim < 1 || feat_dim + num_aa + lig_dim + 2 > EMB_MAX_J)
        return set_error(CBGX_E_INVALID, "embed_compose: bad sizes (N_protein=%d N_ligand=%d F=%d A=%d C=%d; F + A + C + 2 <= %d)",
                         n_protein, n_ligand, feat_dim, num_aa, lig_dim, EMB_MAX_J);
    if (n_protein + n_ligand == 0) return CBGX_OK;
    if (!params || !sort_idx || !x || !h || !ext || (n_protein && (!x_protein || !protein_feat || !protein_aa)) ||
        (n_ligand && (!x_ligand || !ligand_feat)) || (gen_flag && n_ligand && !gen_ligand))
        return set_error(CBGX_E_INVALID, "embed_compose: NULL pointer");
    for (int i = 0; i < 8; ++i)
        if (!params[i]) return set_error(CBGX_E_INVALID, "embed_compose: parameter %d is NULL", i);
    const EmbedParams p{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7]};
    HIP_TRY(launch_embed_compose(x_protein, x_ligand, protein_feat, protein_aa, ligand_feat, sort_idx, gen_protein, gen_ligand, n_protein,
                                 n_ligand, feat_dim, num_aa, lig_dim, p, x, h, ext, gen_flag, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_embed_compose_backward(const float* grad_h, const float* ext, int n_nodes, int feat_dim, int num_aa, int lig_dim, float* partial,
                                int groups, float* grad_out, void* stream) {
    if (n_nodes < 0 || feat_dim < 1 || num_aa < 1 || lig_dim < 1 || feat_dim + num_aa + lig_dim + 2 > EMB_MAX_J || groups < 1 ||
        groups > 256)
        return set_error(CBGX_E_INVALID, "embed_compose_backward: bad sizes (N=%d F=%d A=%d C=%d groups=%d)", n_nodes, feat_dim, num_aa,
                         lig_dim, groups);
    if (!grad_out || (n_nodes && (!grad_h || !ext || !partial))) return set_error(CBGX_E_INVALID, "embed_compose_backward: NULL pointer");
    if (n_nodes == 0) {
        HIP_TRY(hipMemsetAsync(grad_out, 0, sizeof(float) * H * (feat_dim + num_aa + lig_dim + 2), (hipStream_t)stream));
        return CBGX_OK;
    }
    HIP_TRY(launch_embed_compose_backward(grad_h, ext, n_nodes, feat_dim, num_aa, lig_dim, partial, groups, grad_out,
                                          (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_targetdiff_loss_backward(const float* grad_pos, const float* grad_logit, const int64_t* sort_idx, int n_protein,
                                  int n_nodes, int num_classes, const float* g_loss_pos, const float* g_loss_atom,
                                  float* grad_x_out, float* grad_logits, void* stream) {
    if (n_nodes == 0) return CBGX_OK;
    if (n_nodes < 0 || n_protein < 0 || n_protein > n_nodes || num_classes < 1 || num_classes > 32)
        return set_error(CBGX_E_INVALID, "targetdiff_loss_backward: bad sizes (N=%d N_protein=%d C=%d)", n_nodes, n_protein,
                         num_classes);
    if (!grad_pos || !grad_logit || !sort_idx || !grad_x_out || !grad_logits)
        return set_error(CBGX_E_INVALID, "targetdiff_loss_backward: NULL pointer");
    HIP_TRY(launch_train_loss_bwd(grad_pos, grad_logit, sort_idx, n_protein, n_nodes, num_classes, g_loss_pos, g_loss_atom,
                                  grad_x_out, grad_logits, (hipStream_t)stream));
    return CBGX_OK;
}

}  // extern "C"

// libcbgx C ABI (include/cbgx.h): argument checking, workspace carving, kernel sequencing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "../../include/cbgx.h"
#ifdef CBGX_XCHECK
#include "../../include/cbgx_xcheck.h"
#endif
#include "kernels.h"
#include "layout.h"

using namespace cbgx;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace cbgx {
// same thread-local message for the entry points that live in other translation units (api_train.hip)
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace cbgx

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(CBGX_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
    int32_t* nbr;
    int32_t* deg;
    float* e_w;
    float* P;
    float* Qt;
    float* q;
    float* P2;       // second / third node-stage buffer sets: the x2h node stage of layer l+1 runs on an auxiliary
    float* Qt2;      // stream while the h2x block of layer l runs on the caller's stream
    float* q2;
    float* P3;
    float* Qt3;
    float* q3;
    int* act;        // indices of gen_flag nodes (h2x work list)
    int* act_count;
    int* rf_list[3]; // receptive-field pruning: A1, A2, A3
    int* rf_count;   // [3], 64 B apart
    int* fw_list[4];     // static-context cache: D1, S1 = D1 | nbr(D1), D2, S2
    int* fw_count;       // [4], 64 B apart
    // per-node flags behind the lists (node_mfma.hip, list_level_kernel), zeroed with the counters by ONE fill per forward call:
    // receptive-field sets a1 a2 a3; "differs from the ligand-free pocket" D1 (proximity flags of the graph cache) D2 S1 S2; d1flag
    uint8_t *fa1, *fa2, *fa3, *fD1, *fD2, *fS1, *fS2;
    float* hbuf[2];
    float* xbuf[2];
    // x2h layers run as ONE launch over two work lists (edge_mfma.hip, edge_x2h_dual_kernel): destinations with a ligand atom among
    // themselves and their neighbours (d1flag; general role, folded query from Qt) and protein-only ones (query folded in registers)
    uint8_t* d1flag;     // the node or one of its neighbours is a ligand atom: general role of the x2h edge stage
    int* sp_list[4][2];  // [all nodes | cached layer 1 (D2) | pruned A1 | pruned A2][1 = general, 0 = protein-only]
    int* sp_count;       // per set one 128-byte region: general count at +0, protein-only count at +64 bytes
    int* zero_count;     // an always-empty list's count
    unsigned* newmask;   // graph-cached calls: per listed centre, the ranks of its merged neighbour list that hold new entries
    void* counters;      // flags + act_count, rf_count, fw_count, sp_count, zero_count are carved from ONE block: one fill per call
    size_t counters_bytes;
    size_t total;
};

static Workspace carve(void* base, int n) {
    Workspace w;
    size_t off = 0;
    char* b = (char*)base;
    auto take = [&](size_t bytes) { char* p = b + off; off += align_up(bytes); return p; };
    size_t N = (size_t)(n > 0 ? n : 1);
    w.nbr = (int32_t*)take(N * KNN * 4);
    w.deg = (int32_t*)take(N * 4);
    w.e_w = (float*)take(N * KNN * 4);
    w.P = (float*)take(N * PROW * 4);
    w.Qt = (float*)take(N * HEADS * H * 4);
    w.q = (float*)take(N * H * 4);
    w.P2 = (float*)take(N * PROW * 4);
    w.Qt2 = (float*)take(N * HEADS * H * 4);
    w.q2 = (float*)take(N * H * 4);
    w.P3 = (float*)take(N * PROW * 4);
    w.Qt3 = (float*)take(N * HEADS * H * 4);
    w.q3 = (float*)take(N * H * 4);
    w.act = (int*)take(N * 4);
    w.act_count = nullptr;       // (carved from the counter block below)
    for (int k = 0; k < 3; ++k) w.rf_list[k] = (int*)take(N * 4);
    w.rf_count = nullptr;
    for (int k = 0; k < 4; ++k) w.fw_list[k] = (int*)take(N * 4);
    w.fw_count = nullptr;
    w.hbuf[0] = (float*)take(N * H * 4);
    w.hbuf[1] = (float*)take(N * H * 4);
    w.xbuf[0] = (float*)take(N * 3 * 4);
    w.xbuf[1] = (float*)take(N * 3 * 4);
    for (int k = 0; k < 4; ++k) { w.sp_list[k][1] = (int*)take(N * 4); w.sp_list[k][0] = (int*)take(N * 4); }
    w.newmask = (unsigned*)take(N * 4);
    {
        // every device-side list count of a forward call, 64 bytes apart (a counter word is hammered by returning atomics)
        const size_t fl = align_up(N);
        w.counters_bytes = 8 * fl + 256 + 256 + 256 + 4 * 128 + 256;
        char* c = take(w.counters_bytes);
        w.counters = c;
        w.d1flag = (uint8_t*)c; w.fa1 = w.d1flag + fl; w.fa2 = w.fa1 + fl; w.fa3 = w.fa2 + fl;
        w.fD1 = w.fa3 + fl; w.fD2 = w.fD1 + fl; w.fS1 = w.fD2 + fl; w.fS2 = w.fS1 + fl;
        c += 8 * fl;
        w.act_count = (int*)c;
        w.rf_count = (int*)(c + 256);
        w.fw_count = (int*)(c + 512);
        w.sp_count = (int*)(c + 768);
        w.zero_count = (int*)(c + 768 + 4 * 128);
    }
    w.total = off;
    return w;
}

// Auxiliary streams: one per (host thread, caller stream), created on first use, with the two events that fork it from and join it
// back into that caller stream.  A caller that keeps two forward calls in flight on two of its streams (two resident batches: the
// HBM-bound node kernels of one run under the matrix-bound edge kernel of the other) must not have them share an auxiliary stream --
// the node stages of the second call would queue behind all nine of the first.  A small table keyed by the caller's stream handle;
// when it is full the entries change owner round-robin (aux_for).  Nothing here is shared between host threads.
constexpr int MAX_AUX = 8;    // batches a caller can keep in flight without aux streams changing owner (bench --streams 8, sample_many)
struct AuxTable {
    int dev = -1;
    int n = 0;
    int next_victim = 0;
    AuxStream e[MAX_AUX];
};
static thread_local AuxTable g_aux_table;
namespace cbgx {
AuxStream* aux_for(hipStream_t caller) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    AuxTable& t = g_aux_table;
    if (t.n && t.dev != dev) return nullptr;   // one device per host thread (one process per GPU); otherwise stay serial
    for (int k = 0; k < t.n; ++k)
        if (t.e[k].owner == caller) return &t.e[k];
    if (t.n == MAX_AUX) {
        // table full: the entries change owner round-robin.  Stream and events are reused, not destroyed -- work the previous
        // owner's calls queued on the stream simply stays ahead of the new owner's (stream order), waits already enqueued on the
        // events keep referring to the records they were enqueued after, and nothing here blocks the host, so this is also
        // legal while the caller's stream is being captured into a graph.
        AuxStream& e = t.e[t.next_victim];
        t.next_victim = (t.next_victim + 1) % MAX_AUX;
        e.owner = caller;
        return &e;
    }
    AuxStream a;
    a.owner = caller;
    // CBGX_AUX_PRIORITY=low (opt-in, schedule only): the auxiliary stream at the device's least priority, so that the caller's node-level
    // kernels get the CUs first where both queues have workgroups ready (measured: profiles/ab_train_r06r2.log)
    static const bool low = [] { const char* e = getenv("CBGX_AUX_PRIORITY"); return e && e[0] == 'l'; }();
    int least = 0, greatest = 0;
    if (low && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        if (hipStreamCreateWithPriority(&a.s, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a.done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a.done[1], hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(a.s);
        return nullptr;
    }
    t.dev = dev;
    t.e[t.n] = a;
    return &t.e[t.n++];
}
}  // namespace cbgx

extern "C" {

int cbgx_abi_version(void) { return CBGX_ABI_VERSION; }
const char* cbgx_last_error(void) { return g_err; }

#ifdef CBGX_XCHECK
// test-only library (include/cbgx_xcheck.h): route the stages through the first-generation VALU kernels
int cbgx_debug_set_edge_kernel(int impl) {
    if (impl < 0 || impl > 2) return fail(CBGX_E_INVALID, "debug_set_edge_kernel: impl must be 0 (current), 1 (valu) or 2 (mfma, second-generation x2h backward)");
    int old = g_edge_impl;
    g_edge_impl = impl;
    return old;
}
#endif

int cbgx_set_edge_workgroups(int n) { return set_edge_workgroup_limit(n); }

size_t cbgx_packed_weights_floats(int num_layers, int num_classes) {
    if (num_layers < 0 || num_classes < 1) return 0;
    return packed_floats(num_layers, num_classes);
}

// strided / transposed copies are collected and issued as one launch per batch (they only read the caller's tensors and
// are only consumed by later forward calls, so deferring them to the next flush on the same stream is safe)
static thread_local PackBatch g_pack;
static int flush_pack(hipStream_t s) {
    if (g_pack.n) {
        HIP_TRY(launch_pack_copy_multi(g_pack, s));
        g_pack.n = 0;
    }
    return CBGX_OK;
}
static int queue_pack(const float* src, int ld, int off, int tr, float* dst, int dld, int rows, int cols, hipStream_t s) {
    if (g_pack.n == PACK_MAX) { int rc = flush_pack(s); if (rc) return rc; }
    g_pack.p[g_pack.n++] = PackPiece{src, dst, ld, off, tr, dld, rows, cols};
    return CBGX_OK;
}
#define CP(src, ld, off, tr, dst, dld, rows, cols)                                 \
    do {                                                                           \
        int _rc = queue_pack(src, ld, off, tr, dst, dld, rows, cols, s);           \
        if (_rc) return _rc;                                                       \
    } while (0)

// gate section (GATE_* offsets) from dist_emb.1.net.{0.weight,0.bias,1.weight,1.bias,3.weight,3.bias}
static int pack_gate_section(const float* const* t, float* packed, hipStream_t s) {
    // gate MLP: net.0 [160,20], net.1 LN(160), net.3 [1,160]
    CP(t[0], G, 0, 0, packed + GATE_W1, G, GH, G);
    CP(t[1], GH, 0, 0, packed + GATE_B1, GH, 1, GH);
    CP(t[2], GH, 0, 0, packed + GATE_LNG, GH, 1, GH);
    CP(t[3], GH, 0, 0, packed + GATE_LNB, GH, 1, GH);
    CP(t[4], GH, 0, 0, packed + GATE_W2, GH, 1, GH);
    CP(t[5], 1, 0, 0, packed + GATE_B2, 1, 1, 1);
    HIP_TRY(launch_pack_gate_img(t[0], t[1], t[2], t[3], t[4], packed + GATE_IMG, s));
    return flush_pack(s);
}

// ---- attention blocks (ATT layout) from k(6) v(6) q(6) MLP tensors each; blk 0 = x2h, 1 = h2x.  Packed for ALL blocks of a
// model phase by phase (the weights are re-packed every training step): copies of the reference tensors, stage 1 (centring),
// copies of the centred matrices, stage 2 (every fragment table) -- ~20 launches instead of 12 per block.
struct BlockRef { const float* const* p; int blk; float* a; };

// copies that read the caller's tensors only
static int queue_block_copies(const BlockRef& r, hipStream_t s) {
    const float* const* p = r.p;
    float* a = r.a;
    const int blk = r.blk;
    const float *wk0 = p[0], *bk0 = p[1], *gk = p[2], *bek = p[3], *wk1 = p[4];
    const float *wv0 = p[6], *bv0 = p[7], *gv = p[8], *bev = p[9], *wv1 = p[10], *bv1 = p[11];
    const float *wq0 = p[12], *bq0 = p[13], *gq = p[14], *beq = p[15], *wq1 = p[16], *bq1 = p[17];
    // node projection [k][c]: PDk | PDv | PSk | PSv | q hidden
    CP(wk0, KV_IN, NT + NT * G, 1, a + A_WN + 0 * H, PROW, H, H);
    CP(wv0, KV_IN, NT + NT * G, 1, a + A_WN + 1 * H, PROW, H, H);
    CP(wk0, KV_IN, NT + NT * G + H, 1, a + A_WN + 2 * H, PROW, H, H);
    CP(wv0, KV_IN, NT + NT * G + H, 1, a + A_WN + 3 * H, PROW, H, H);
    CP(wq0, H, 0, 1, a + A_WN + 4 * H, PROW, H, H);
    CP(bk0, H, 0, 0, a + A_BN + 0 * H, H, 1, H);
    CP(bv0, H, 0, 0, a + A_BN + 1 * H, H, 1, H);
    CP(bq0, H, 0, 0, a + A_BN + 4 * H, H, 1, H);
    // edge-type one-hot columns and rbf columns of the first Linear
    CP(wk0, KV_IN, 0, 1, a + A_WT, 2 * H, NT, H);
    CP(wv0, KV_IN, 0, 1, a + A_WT + H, 2 * H, NT, H);
    CP(wk0, KV_IN, NT, 1, a + A_WR, 2 * H, NT * G, H);
    CP(wv0, KV_IN, NT, 1, a + A_WR + H, 2 * H, NT * G, H);
    CP(gk, H, 0, 0, a + A_LNK_G, H, 1, H);
    CP(bek, H, 0, 0, a + A_LNK_B, H, 1, H);
    CP(gv, H, 0, 0, a + A_LNV_G, H, 1, H);
    CP(bev, H, 0, 0, a + A_LNV_B, H, 1, H);
    CP(gq, H, 0, 0, a + A_LNQ_G, H, 1, H);
    CP(beq, H, 0, 0, a + A_LNQ_B, H, 1, H);
    CP(wq1, H, 0, 1, a + A_WQ1T, H, H, H);
    CP(bq1, H, 0, 0, a + A_BQ1, H, 1, H);
    CP(wk1, H, 0, 0, a + A_WBK, H, H, H);
    CP(wk1, H, 0, 1, a + A_WBKT, H, H, H);
    CP(wq1, H, 0, 0, a + A_WQ1O, H, H, H);
    for (int ty = 0; ty < NT; ++ty) {   // WRT[ty][c][g] = W_a[c][4 + 20 ty + g]  (g 20..31 stay zero)
        CP(wk0, KV_IN, NT + G * ty, 0, a + A_WRT + (size_t)ty * 2 * H * 32, 32, H, G);
        CP(wv0, KV_IN, NT + G * ty, 0, a + A_WRT + ((size_t)ty * 2 * H + H) * 32, 32, H, G);
    }
    // LDS image of the MFMA edge kernel: LayerNorm affines; second v Linear
    float* img = a + A_IMG;
    CP(gk, H, 0, 0, img + IMG_LN + 0 * H, H, 1, H);
    CP(bek, H, 0, 0, img + IMG_LN + 1 * H, H, 1, H);
    CP(gv, H, 0, 0, img + IMG_LN + 2 * H, H, 1, H);
    CP(bev, H, 0, 0, img + IMG_LN + 3 * H, H, 1, H);
    if (blk == 0) {
        CP(wv1, H, 0, 1, a + A_WBV, H, H, H);   // [m][n]
        CP(bv1, H, 0, 0, a + A_BBV, H, 1, H);
    } else {
        CP(wv1, H, 0, 0, a + A_WBV, H, HEADS, H);  // [head][m]
        CP(bv1, HEADS, 0, 0, a + A_BBV, HEADS, 1, HEADS);
    }
    return CBGX_OK;
}

// copies that read the centred first Linears stage 1 has just produced on this stream
static int queue_centred_copies(const BlockRef& r, hipStream_t s) {
    float* a = r.a;
    CP(a + A_WAKC, KV_IN, NT, 1, a + A_WRC, 2 * H, NT * G, H);
    CP(a + A_WAVC, KV_IN, NT, 1, a + A_WRC + H, 2 * H, NT * G, H);
    return CBGX_OK;
}

static int pack_attention_blocks(const BlockRef* refs, int n_blocks, hipStream_t s) {
    for (int k0 = 0; k0 < n_blocks; k0 += PACK_BLOCKS_MAX) {
        const int nb = n_blocks - k0 < PACK_BLOCKS_MAX ? n_blocks - k0 : PACK_BLOCKS_MAX;
        PackBlocks pb;
        memset(&pb, 0, sizeof(pb));
        pb.n = nb;
        for (int k = 0; k < nb; ++k) {
            const BlockRef& r = refs[k0 + k];
            pb.wk0[k] = r.p[0]; pb.bk0[k] = r.p[1]; pb.wk1[k] = r.p[4];
            pb.wv0[k] = r.p[6]; pb.bv0[k] = r.p[7]; pb.wv1[k] = r.p[10];
            pb.wq0[k] = r.p[12]; pb.bq0[k] = r.p[13]; pb.wq1[k] = r.p[16];
            pb.att[k] = r.a;
            pb.nsrc[k][0] = r.a + A_WAKC; pb.nsrc[k][1] = r.a + A_WAVC; pb.nsrc[k][2] = r.a + A_WAKC; pb.nsrc[k][3] = r.a + A_WAVC;
            pb.nsrc[k][4] = r.p[12];
            pb.x2h[k] = r.blk == 0;
            int rc = queue_block_copies(r, s);
            if (rc) return rc;
        }
        { int rc = flush_pack(s); if (rc) return rc; }
        HIP_TRY(launch_pack_stage1(pb, s));
        for (int k = 0; k < nb; ++k) { int rc = queue_centred_copies(refs[k0 + k], s); if (rc) return rc; }
        { int rc = flush_pack(s); if (rc) return rc; }
        HIP_TRY(launch_pack_stage2(pb, s));
    }
    return CBGX_OK;
}

int cbgx_pack_weights(const float* const* t, int num_tensors, int L, int C, float* packed, void* stream) {
    if (!t || !packed) return fail(CBGX_E_INVALID, "pack_weights: NULL pointer");
    if (L < 1 || C < 1) return fail(CBGX_E_INVALID, "pack_weights: num_layers=%d num_classes=%d", L, C);
    if (num_tensors != 6 + 36 * L + 4)
        return fail(CBGX_E_INVALID, "pack_weights: expected %d tensors, got %d", 6 + 36 * L + 4, num_tensors);
    for (int i = 0; i < num_tensors; ++i)
        if (!t[i]) return fail(CBGX_E_INVALID, "pack_weights: tensor %d is NULL", i);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(packed, 0, packed_floats(L, C) * sizeof(float), s));
    { int rc = pack_gate_section(t, packed, s); if (rc) return rc; }
    {
        std::vector<BlockRef> refs;
        for (int l = 0; l < L; ++l)
            for (int blk = 0; blk < 2; ++blk)
                refs.push_back(BlockRef{t + 6 + 36 * l + 18 * blk /* k(6) v(6) q(6) */, blk, packed + (blk == 0 ? x2h_off(l) : h2x_off(l))});
        int rc = pack_attention_blocks(refs.data(), (int)refs.size(), s);
        if (rc) return rc;
    }
    const float* const* c = t + 6 + 36 * L;
    float* cp = packed + cls_off(L);
    CP(c[0], H, 0, 1, cp + C_W0T, H, H, H);
    CP(c[1], H, 0, 0, cp + C_B0, H, 1, H);
    CP(c[2], H, 0, 1, cp + C_W1T, C, H, C);
    CP(c[3], C, 0, 0, cp + cls_b1(C), C, 1, C);
    return flush_pack(s);
}

// ---- a stack of H2X blocks on its own kNN graph + gate (DiffBP's CoMPredictor) ---------------------
size_t cbgx_packed_h2x_stack_floats(int num_layers) {
    if (num_layers < 1) return 0;
    return GATE_SIZE + (size_t)num_layers * ATT_SIZE;
}

int cbgx_pack_h2x_stack(const float* const* t, int num_tensors, int L, float* packed, void* stream) {
    if (!t || !packed) return fail(CBGX_E_INVALID, "pack_h2x_stack: NULL pointer");
    if (L < 1 || num_tensors != 6 + 18 * L)
        return fail(CBGX_E_INVALID, "pack_h2x_stack: expected %d tensors for %d layers, got %d", 6 + 18 * L, L, num_tensors);
    for (int i = 0; i < num_tensors; ++i)
        if (!t[i]) return fail(CBGX_E_INVALID, "pack_h2x_stack: tensor %d is NULL", i);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(packed, 0, cbgx_packed_h2x_stack_floats(L) * sizeof(float), s));
    { int rc = pack_gate_section(t, packed, s); if (rc) return rc; }
    std::vector<BlockRef> refs;
    for (int l = 0; l < L; ++l) refs.push_back(BlockRef{t + 6 + 18 * l, 1, packed + GATE_SIZE + (size_t)l * ATT_SIZE});
    { int rc = pack_attention_blocks(refs.data(), (int)refs.size(), s); if (rc) return rc; }
    return flush_pack(s);
}

int cbgx_h2x_stack_forward(const float* packed, int num_layers, const float* x, const float* h,
                           const int32_t* graph_ptr, const uint8_t* lig_flag, const uint8_t* gen_flag, int n_nodes,
                           int n_graphs, float* x_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_nodes < 0 || n_graphs < 0 || num_layers < 1) return fail(CBGX_E_INVALID, "h2x_stack: bad sizes");
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !x || !h || !graph_ptr || !lig_flag || !gen_flag || !x_out || !workspace)
        return fail(CBGX_E_INVALID, "h2x_stack: NULL pointer");
    Workspace w = carve(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return fail(CBGX_E_WORKSPACE, "h2x_stack: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_build_active(gen_flag, n_nodes, w.act, w.act_count, s));
    if (g_edge_impl == 1) {
        HIP_TRY(launch_knn(x, graph_ptr, n_graphs, n_nodes, w.nbr, w.deg, s));
        HIP_TRY(launch_gate(packed, x, w.nbr, w.deg, n_nodes, w.e_w, s));
    } else {
        // An H2X block only ever reads the neighbour list and the gate values of the nodes it moves (edge kernel and source marking
        // both run over the gen_flag list), and the stack's graph is built once from the input coordinates (diffbp.py:84-93): the
        // kNN search and the gate MLP run on the listed rows only -- the ligand atoms, ~5 % of a pocket -- instead of on every node
        // (DiffBP paid 415 + 276 us per step for them at 200 graphs, against 163 + 112 us for the denoiser's cached graph).
        HIP_TRY(launch_knn_reg(x, graph_ptr, n_graphs, n_nodes, w.nbr, w.deg, s, w.act, w.act_count));
        HIP_TRY(launch_gate_mfma(packed, x, w.nbr, w.deg, n_nodes, w.e_w, s, w.act, w.act_count));
    }
    // The source rows of the stack: the in-neighbours of the movable rows (and those rows themselves).  Only their PS columns are
    // ever gathered, so only they are projected -- and only their rows of h are read, which is what lets the denoiser in front
    // (cbgx_unitransformer_forward_cached, CBGX_FWD_H_ON_SOURCES) skip every other row of its last layers.
    const int *src = nullptr, *src_n = nullptr;
    if (g_edge_impl != 1) {
        HIP_TRY(launch_mark_seed(gen_flag, gen_flag, n_nodes, w.fa1, s));
        HIP_TRY(launch_mark_nbr(w.act, w.act_count, n_nodes, w.nbr, w.deg, w.fa1, s));
        HIP_TRY(launch_build_active(w.fa1, n_nodes, w.rf_list[0], w.rf_count, s));
        src = w.rf_list[0]; src_n = w.rf_count;
    }
    const float* xc = x;
    for (int l = 0; l < num_layers; ++l) {
        float* xn = (l == num_layers - 1) ? x_out : w.xbuf[l & 1];
        HIP_TRY(launch_attention(false, packed + GATE_SIZE + (size_t)l * ATT_SIZE, xc, h, w.nbr, w.deg, lig_flag, gen_flag,
                                 w.e_w, n_nodes, w.P, w.Qt, w.q, xn, nullptr, w.act, w.act_count, src, src_n, s));
        xc = xn;
    }
    return CBGX_OK;
}

size_t cbgx_workspace_bytes(int n_nodes, int n_graphs) {
    (void)n_graphs;
    return carve(nullptr, n_nodes).total;
}

int cbgx_knn_graph(const float* x, const int32_t* graph_ptr, int n_graphs, int n_nodes, int k, int32_t* nbr,
                   int32_t* deg, void* stream) {
    if (k != KNN) return fail(CBGX_E_INVALID, "knn_graph: only k=%d is supported (got %d)", KNN, k);
    if (n_nodes < 0 || n_graphs < 0) return fail(CBGX_E_INVALID, "knn_graph: negative size");
    if (n_nodes == 0) return CBGX_OK;
    if (!x || !graph_ptr || !nbr || !deg || n_graphs < 1) return fail(CBGX_E_INVALID, "knn_graph: NULL pointer");
    HIP_TRY(launch_knn(x, graph_ptr, n_graphs, n_nodes, nbr, deg, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_edge_gate(const float* packed, const float* x, const int32_t* nbr, const int32_t* deg, int n_nodes,
                   float* e_w, void* stream) {
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !x || !nbr || !deg || !e_w || n_nodes < 0) return fail(CBGX_E_INVALID, "edge_gate: bad argument");
    HIP_TRY(launch_gate(packed, x, nbr, deg, n_nodes, e_w, (hipStream_t)stream));
    return CBGX_OK;
}

// (general, protein-only) list pair `set` of the workspace from a destination list (NULL = all nodes) and the d1 flags
static int split_by_d1(const Workspace& w, int set, const int* list, const int* count, int n_nodes, hipStream_t s,
                       bool counters_zeroed = false) {
    HIP_TRY(launch_split_list(list, count, n_nodes, w.d1flag, w.sp_list[set][1], w.sp_count + 32 * set, w.sp_list[set][0],
                              w.sp_count + 32 * set + 16, s, counters_zeroed));
    return CBGX_OK;
}

int cbgx_x2h_attention(const float* packed, int layer, const float* x, const float* h, const int32_t* nbr,
                       const int32_t* deg, const uint8_t* lig_flag, const float* e_w, int n_nodes, float* h_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !x || !h || !nbr || !deg || !lig_flag || !e_w || !h_out || !workspace || layer < 0 || n_nodes < 0)
        return fail(CBGX_E_INVALID, "x2h_attention: bad argument");
    Workspace w = carve(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return fail(CBGX_E_WORKSPACE, "x2h_attention: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    if (g_edge_impl == 1) {
        HIP_TRY(launch_attention(true, packed + x2h_off(layer), x, h, nbr, deg, lig_flag, nullptr, e_w, n_nodes, w.P,
                                 w.Qt, w.q, h_out, nullptr, nullptr, nullptr, nullptr, nullptr, s));
        return CBGX_OK;
    }
    // what a layer of cbgx_unitransformer_forward runs: protein-only destinations fold their query in registers, Qt is
    // produced for the others only, one two-role edge launch
    HIP_TRY(launch_mark_from_nbr(lig_flag, nbr, deg, n_nodes, w.d1flag, s));
    { int rc = split_by_d1(w, 0, nullptr, nullptr, n_nodes, s); if (rc) return rc; }
    const float* att = packed + x2h_off(layer);
    HIP_TRY(launch_node_mfma(att, h, lig_flag, n_nodes, w.P, w.q, w.Qt, nullptr, nullptr, nullptr, nullptr, s, true,
                             w.sp_list[0][1], w.sp_count));
    HIP_TRY(launch_edge_x2h_dual(att, x, h, w.P, w.Qt, w.q, nbr, deg, lig_flag, nullptr, e_w, n_nodes, h_out, w.sp_list[0][0],
                                 w.sp_count + 16, w.sp_list[0][1], w.sp_count, true, s));
    return CBGX_OK;
}

int cbgx_h2x_attention(const float* packed, int layer, const float* x, const float* h, const int32_t* nbr,
                       const int32_t* deg, const uint8_t* lig_flag, const uint8_t* gen_flag, const float* e_w,
                       int n_nodes, float* x_out, float* delta_x, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !x || !h || !nbr || !deg || !lig_flag || !gen_flag || !e_w || !x_out || !workspace || layer < 0 ||
        n_nodes < 0)
        return fail(CBGX_E_INVALID, "h2x_attention: bad argument");
    Workspace w = carve(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return fail(CBGX_E_WORKSPACE, "h2x_attention: workspace %zu < %zu", workspace_bytes, w.total);
    HIP_TRY(launch_build_active(gen_flag, n_nodes, w.act, w.act_count, (hipStream_t)stream));
    HIP_TRY(launch_attention(false, packed + h2x_off(layer), x, h, nbr, deg, lig_flag, gen_flag, e_w, n_nodes, w.P,
                             w.Qt, w.q, x_out, delta_x, w.act, w.act_count, nullptr, nullptr, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_classifier(const float* packed, int num_layers, int num_classes, const float* h, int n_nodes, float* logits,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !h || !logits || !workspace || num_layers < 0 || num_classes < 1 || n_nodes < 0)
        return fail(CBGX_E_INVALID, "classifier: bad argument");
    Workspace w = carve(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return fail(CBGX_E_WORKSPACE, "classifier: workspace %zu < %zu", workspace_bytes, w.total);
    const float* c = packed + cls_off(num_layers);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_node_gemm(h, H, c + C_W0T, c + C_B0, w.P, H, n_nodes, H, 1, s));
    HIP_TRY(launch_node_gemm(w.P, H, c + C_W1T, c + cls_b1(num_classes), logits, num_classes, n_nodes, num_classes, 0, s));
    return CBGX_OK;
}

static int forward_impl(const float* packed, int num_layers, int num_classes, const float* x, const float* h,
                        const int32_t* graph_ptr, const uint8_t* lig_flag, const uint8_t* gen_flag, int n_nodes,
                        int n_graphs, float* x_out, float* h_out, float* logits, const float* static_h1,
                        const float* static_h2, const int32_t* static_nbr, const int32_t* static_deg,
                        const float* static_ew, const float* static_r32sq, unsigned flags, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (n_nodes < 0 || n_graphs < 0 || num_layers < 1) return fail(CBGX_E_INVALID, "forward: bad sizes");
    if (flags & ~CBGX_FWD_H_ON_SOURCES) return fail(CBGX_E_INVALID, "forward: unknown flags 0x%x", flags);
    if (n_nodes == 0) return CBGX_OK;
    if (!packed || !x || !h || !graph_ptr || !lig_flag || !gen_flag || !x_out || !workspace)
        return fail(CBGX_E_INVALID, "forward: NULL pointer");
    if (logits && num_classes < 1) return fail(CBGX_E_INVALID, "forward: num_classes=%d", num_classes);
    Workspace w = carve(workspace, n_nodes);
    if (workspace_bytes < w.total)
        return fail(CBGX_E_WORKSPACE, "forward: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(w.counters, 0, w.counters_bytes, s));      // every list count of this call: one fill instead of ~15
    const bool cached = static_h1 && static_h2 && num_layers >= 4;
    // with the graph part of the cache, only the nodes that have a ligand atom within reach get a fresh neighbour list
    // and gate: everything else about the pocket's own graph was computed once (same order, same bits)
    const bool graph_cached = cached && static_nbr && static_deg && static_ew && static_r32sq && g_edge_impl != 1;
    // H2X only ever moves gen_flag nodes (x_out = x + dx * gen_flag): they are listed once (`act`), every h2x block runs on the list.
    // Receptive-field pruning (only when the caller does not ask for h_out): the outputs that remain are x_out and the
    // logits of ligand rows, so the last x2h blocks only have to produce features that can still reach them:
    //   A1 = gen | lig | nbr(gen)      destinations of the last x2h (classifier rows, the last h2x's own + source rows)
    //   A2 = A1 | nbr(A1)              its sources = destinations of the x2h before it;   A3 = A2 | nbr(A2) its sources
    // Rows outside these sets are simply not written in the last two feature buffers (and never read).  A1 is built in every
    // call: it is also the set of possible *sources* of an H2X block, so the h2x node projection PS is produced for those rows only.
    // (CBGX_FWD_H_ON_SOURCES: h_out is wanted on A1 only -- the destinations of the last x2h block -- so the same pruning holds)
    const bool prune = (h_out == nullptr || (flags & CBGX_FWD_H_ON_SOURCES)) && num_layers >= 3;
    // Static-context cache (optional): static_h1 / static_h2 [N,128] hold the features that leave layer 0 / layer 1 in
    // the ligand-free pocket (rows of ligand atoms unused).  A protein node with no ligand atom among its neighbours sees
    // exactly that pocket in layer 0, so its output is the cached row; the set that differs grows by one hop per layer:
    //   D1 = lig | {i : nbr(i) has a ligand atom},   D2 = D1 | {i : nbr(i) meets D1};   sources S_k = D_k | nbr(D_k).
    // Layers 0 and 1 then run on D1 / D2 only, every other row of their output is a copy of the cache.
    // x2h layers run their (general, protein-only) list pairs: d1flag = the node or one of its neighbours is a ligand atom, from
    // the neighbour lists this call works with.  Set 0 all nodes, 1 the cached layer 1 (D2), 2 / 3 the pruned layers (A1 / A2).
    const bool dual = g_edge_impl != 1;
    GraphFlags gf{gen_flag, lig_flag, graph_cached ? w.fD1 : w.d1flag, w.d1flag, w.fa1, w.fa2, w.fa3, w.fD2, w.fS1, w.fS2};
    if (graph_cached) {
        // D1 flags + list, the pocket's own graph and the cached features of layers 0 / 1 (into hbuf[0] / hbuf[1]: num_layers >= 4,
        // so neither is the caller's h_out, and nothing else touches them before their layer): one launch
        HIP_TRY(launch_graph_cache_begin(x, graph_ptr, n_graphs, lig_flag, static_r32sq, n_nodes, w.fD1, w.fw_list[0], w.fw_count,
                                         static_nbr, static_deg, static_ew, w.nbr, w.deg, w.e_w, static_h1, static_h2, w.hbuf[0],
                                         w.hbuf[1], s));
        // the D1 centres' merged neighbour lists and gate values (kept pocket entries carry their cached value to their new
        // rank; the gate MLP runs on the ligand atoms that entered the list): one launch.  CBGX_MERGE_GATE=0: the two kernels.
        // Small inputs: ONE launch (knn_merge_gate_kernel: a persistent kernel at one wave per SIMD -- at 173 k nodes it measured
        // 612 us against 153 + 175 for the two kernels, profiles/ab_fwd_r05f.log, so large inputs keep two launches); large inputs:
        // the merge marks the new ranks and the gate kernel evaluates only those.  CBGX_MERGE_GATE=0: the round-4 pair (all slots).
        static const int merge_gate = [] { const char* e = getenv("CBGX_MERGE_GATE"); return e ? atoi(e) : 1; }();
        if (merge_gate && n_nodes <= GRAPH_LISTS_MAX_NODES) {
            HIP_TRY(launch_knn_merge_gate(packed, x, graph_ptr, n_graphs, n_nodes, lig_flag, static_nbr, static_deg, static_ew, w.nbr,
                                          w.deg, w.e_w, s, w.fw_list[0], w.fw_count));
        } else if (merge_gate) {
            HIP_TRY(launch_knn_merge(x, graph_ptr, n_graphs, n_nodes, lig_flag, static_nbr, static_deg, w.nbr, w.deg, s, w.fw_list[0],
                                     w.fw_count, static_ew, w.e_w, w.newmask));
            HIP_TRY(launch_gate_mfma(packed, x, w.nbr, w.deg, n_nodes, w.e_w, s, w.fw_list[0], w.fw_count, w.newmask));
        } else {
            HIP_TRY(launch_knn_merge(x, graph_ptr, n_graphs, n_nodes, lig_flag, static_nbr, static_deg, w.nbr, w.deg, s, w.fw_list[0],
                                     w.fw_count));
            HIP_TRY(launch_gate_mfma(packed, x, w.nbr, w.deg, n_nodes, w.e_w, s, w.fw_list[0], w.fw_count));
        }
    } else {
        HIP_TRY(launch_knn(x, graph_ptr, n_graphs, n_nodes, w.nbr, w.deg, s));
        HIP_TRY(launch_gate(packed, x, w.nbr, w.deg, n_nodes, w.e_w, s));
    }
    // every list of the call.  Large inputs: three level kernels over the flags (a level reads what the previous one completed) and one
    // compaction; inputs of <= GRAPH_LISTS_MAX_NODES nodes: one launch, a workgroup per graph with the graph's flags in LDS (round 5:
    // the four launches were 20 us of a 600 us one-graph step)
    {
        const bool per_graph = n_nodes <= GRAPH_LISTS_MAX_NODES && g_edge_impl != 1;
        const uint8_t* flag_ptr[GF_COUNT] = {gen_flag, lig_flag, gf.D1, w.d1flag, w.fa1, w.fa2, w.fa3, w.fD2, w.fS1, w.fS2};
        ListJobs jobs;
        GraphListJobs gjobs;
        memset(&jobs, 0, sizeof(jobs));
        memset(&gjobs, 0, sizeof(gjobs));
        bool jobs_overflow = false;
        auto add = [&](int f, int f2, int want2, int* list, int* count) {
            if (jobs.n_jobs >= LIST_JOBS_MAX) { jobs_overflow = true; return; }
            const int k = jobs.n_jobs++;
            jobs.flag[k] = f == GF_ALL ? nullptr : flag_ptr[f]; jobs.flag2[k] = f2 == GF_ALL ? nullptr : flag_ptr[f2];
            jobs.want2[k] = want2; jobs.list[k] = list; jobs.count[k] = count;
            gjobs.flag[k] = (signed char)f; gjobs.flag2[k] = (signed char)f2; gjobs.want2[k] = (signed char)want2;
            gjobs.list[k] = list; gjobs.count[k] = count;
            gjobs.n_jobs = jobs.n_jobs;
        };
        add(GF_GEN, GF_ALL, 0, w.act, w.act_count);
        add(GF_a1, GF_ALL, 0, w.rf_list[0], w.rf_count);
        if (prune) {
            add(GF_a2, GF_ALL, 0, w.rf_list[1], w.rf_count + 16);
            add(GF_a3, GF_ALL, 0, w.rf_list[2], w.rf_count + 32);
        }
        if (cached) {
            if (!graph_cached) add(GF_d1, GF_ALL, 0, w.fw_list[0], w.fw_count);
            add(GF_D2, GF_ALL, 0, w.fw_list[2], w.fw_count + 32);
            add(GF_S1, GF_ALL, 0, w.fw_list[1], w.fw_count + 16);
            add(GF_S2, GF_ALL, 0, w.fw_list[3], w.fw_count + 48);
        }
        if (dual) {
            auto pair = [&](int set, int f) {
                add(f, GF_d1, 1, w.sp_list[set][1], w.sp_count + 32 * set);
                add(f, GF_d1, 0, w.sp_list[set][0], w.sp_count + 32 * set + 16);
            };
            pair(0, GF_ALL);
            if (cached) pair(1, GF_D2);
            if (prune) { pair(2, GF_a1); pair(3, GF_a2); }
        }
        if (jobs_overflow) return fail(CBGX_E_INVALID, "forward: more than %d node lists (LIST_JOBS_MAX)", LIST_JOBS_MAX);
        if (per_graph) {
            HIP_TRY(launch_graph_lists(gen_flag, lig_flag, graph_cached ? w.fD1 : nullptr, w.nbr, w.deg, graph_ptr, n_graphs, gjobs,
                                       cached, prune, w.d1flag, s));
        } else {
            HIP_TRY(launch_list_level(gf, w.nbr, w.deg, n_nodes, 0, cached, prune, s));
            if (cached || prune) {
                HIP_TRY(launch_list_level(gf, w.nbr, w.deg, n_nodes, 1, cached, prune, s));
                HIP_TRY(launch_list_level(gf, w.nbr, w.deg, n_nodes, 2, cached, prune, s));
            }
            HIP_TRY(launch_build_lists(jobs, n_nodes, s));
        }
    }
    struct X2HLists { const int *gen, *gen_n, *pp, *pp_n; bool full; };
    auto x2h_lists = [&](int l) {
        X2HLists r{w.sp_list[0][1], w.sp_count, w.sp_list[0][0], w.sp_count + 16, true};
        auto set = [&](int k) { r = X2HLists{w.sp_list[k][1], w.sp_count + 32 * k, w.sp_list[k][0], w.sp_count + 32 * k + 16, false}; };
        if (cached && l == 0) r = X2HLists{w.fw_list[0], w.fw_count, w.sp_list[0][0], w.zero_count, false};   // D1: general role only
        if (cached && l == 1) set(1);
        if (prune && l >= num_layers - 2) set(2 + (num_layers - 1 - l));
        return r;
    };
    // Two-stream schedule (MFMA kernels, profiling off): the node stage of x2h(l+1) only needs h_{l+1}, which exists as
    // soon as the x2h edge kernel of layer l has run, while the h2x block of layer l (which only moves coordinates) is
    // still to come -- so it runs on an auxiliary stream next to that h2x block.  Three node-stage buffer sets: x2h
    // alternates between two, h2x has its own.
    static const bool overlap_env = [] { const char* e = getenv("CBGX_OVERLAP"); return !e || atoi(e) != 0; }();
    AuxStream* aux = (overlap_env && g_edge_impl != 1 && !profile_is_on() && num_layers > 1) ? aux_for(s) : nullptr;
    const bool overlap = aux != nullptr;
    auto layer_lists = [&](int l, const int*& dst, const int*& dst_n, const int*& src, const int*& src_n) {
        dst = dst_n = src = src_n = nullptr;
        if (cached && l < 2) {
            dst = w.fw_list[2 * l]; dst_n = w.fw_count + 32 * l;
            src = w.fw_list[2 * l + 1]; src_n = w.fw_count + 32 * l + 16;
        }
        if (prune && l >= num_layers - 2) {
            const int k = num_layers - 1 - l;   // 0 for the last layer, 1 for the one before
            dst = w.rf_list[k]; dst_n = w.rf_count + 16 * k;
            src = w.rf_list[k + 1]; src_n = w.rf_count + 16 * (k + 1);
        }
    };
    float* Pset[2] = {w.P, w.P2};
    float* Qtset[2] = {w.Qt, w.Qt2};
    float* qset[2] = {w.q, w.q2};
    const float* xc = x;
    const float* hc = h;
    // Small inputs (round 5): ONE stream and ONE node-stage launch per layer.  The h2x block of layer l and the x2h block of layer
    // l + 1 both read h_{l+1} and nothing else that is new, so their node stages are jobs of the same node_stage_kernel launch and a
    // layer is three dependent launches -- x2h edge, node stages, h2x edge -- with no event between them.  The two-stream schedule
    // below hides the second node stage behind the h2x block instead, at the price of a fork and a join event per layer, ~7 us
    // each on the caller's queue: 80 us per layer at one graph, of which 14 are the events and 24 + 25 the two edge launches
    // (profiles/step_timeline_r05a_p1s1_ov1.json).  CBGX_FUSE_ROWS: largest input that takes this schedule (0 = never).
    static const int fuse_rows = [] { const char* e = getenv("CBGX_FUSE_ROWS"); return e ? atoi(e) : NODE_STAGE_MAX_ROWS; }();
    if (dual && num_layers > 1 && n_nodes <= fuse_rows && n_nodes <= NODE_STAGE_MAX_ROWS) {
        {
            const int *dst, *dst_n, *src, *src_n;
            layer_lists(0, dst, dst_n, src, src_n);
            NodeStageJobs jobs;
            jobs.n = 0;
            // (the general role of a cached layer 0 is the whole D1 list: folded rows for all of it; every other x2h layer's
            // general role is the d1-flagged part of its destinations)
            add_node_stage_jobs(jobs, packed + x2h_off(0), Pset[0], qset[0], Qtset[0], dst, dst_n, src, src_n,
                                cached ? nullptr : w.d1flag);
            HIP_TRY(launch_node_stage_jobs(jobs, h, lig_flag, n_nodes, s));
        }
        for (int l = 0; l < num_layers; ++l) {
            float* hn = (l == num_layers - 1 && h_out) ? h_out : w.hbuf[l & 1];
            float* xn = (l == num_layers - 1) ? x_out : w.xbuf[l & 1];
            if (cached && l < 2 && !graph_cached)     // (graph-cached calls: restore_graph_kernel has already placed both)
                HIP_TRY(hipMemcpyAsync(hn, l == 0 ? static_h1 : static_h2, (size_t)n_nodes * H * sizeof(float),
                                       hipMemcpyDeviceToDevice, s));
            const X2HLists xl = x2h_lists(l);
            const int set = l & 1;
            HIP_TRY(launch_edge_x2h_dual(packed + x2h_off(l), xc, hc, Pset[set], Qtset[set], qset[set], w.nbr, w.deg, lig_flag,
                                         gen_flag, w.e_w, n_nodes, hn, xl.pp, xl.pp_n, xl.gen, xl.gen_n, xl.full, s));
            NodeStageJobs jobs;
            jobs.n = 0;
            add_node_stage_jobs(jobs, packed + h2x_off(l), w.P3, w.q3, w.Qt3, w.act, w.act_count, w.rf_list[0], w.rf_count);
            if (l + 1 < num_layers) {
                const int *d2, *d2n, *s2, *s2n;
                layer_lists(l + 1, d2, d2n, s2, s2n);
                add_node_stage_jobs(jobs, packed + x2h_off(l + 1), Pset[set ^ 1], qset[set ^ 1], Qtset[set ^ 1], d2, d2n, s2, s2n,
                                    w.d1flag);
            }
            HIP_TRY(launch_node_stage_jobs(jobs, hn, lig_flag, n_nodes, s));
            HIP_TRY(launch_edge_mfma(false, packed + h2x_off(l), xc, hn, w.P3, w.Qt3, w.nbr, w.deg, lig_flag, gen_flag, w.e_w,
                                     n_nodes, xn, nullptr, w.act, w.act_count, s));
            xc = xn;
            hc = hn;
        }
    } else {
    if (overlap) {
        const int *dst, *dst_n, *src, *src_n;
        layer_lists(0, dst, dst_n, src, src_n);
        const X2HLists xl = x2h_lists(0);
        HIP_TRY(launch_node_mfma(packed + x2h_off(0), h, lig_flag, n_nodes, Pset[0], qset[0], Qtset[0], dst, dst_n, src,
                                 src_n, s, true, xl.gen, xl.gen_n));
    }
    for (int l = 0; l < num_layers; ++l) {
        float* hn = (l == num_layers - 1 && h_out) ? h_out : w.hbuf[l & 1];
        float* xn = (l == num_layers - 1) ? x_out : w.xbuf[l & 1];
        const int *dst, *dst_n, *src, *src_n;
        layer_lists(l, dst, dst_n, src, src_n);
        if (cached && l < 2 && !graph_cached)
            HIP_TRY(hipMemcpyAsync(hn, l == 0 ? static_h1 : static_h2, (size_t)n_nodes * H * sizeof(float),
                                   hipMemcpyDeviceToDevice, s));
        const X2HLists xl = x2h_lists(l);
        if (!overlap) {
            if (dual) {
                HIP_TRY(launch_node_mfma(packed + x2h_off(l), hc, lig_flag, n_nodes, w.P, w.q, w.Qt, dst, dst_n, src, src_n, s, true,
                                         xl.gen, xl.gen_n));
                HIP_TRY(launch_edge_x2h_dual(packed + x2h_off(l), xc, hc, w.P, w.Qt, w.q, w.nbr, w.deg, lig_flag, gen_flag, w.e_w,
                                             n_nodes, hn, xl.pp, xl.pp_n, xl.gen, xl.gen_n, xl.full, s));
            } else {
                HIP_TRY(launch_attention(true, packed + x2h_off(l), xc, hc, w.nbr, w.deg, lig_flag, gen_flag, w.e_w, n_nodes,
                                         w.P, w.Qt, w.q, hn, nullptr, dst, dst_n, src, src_n, s));
            }
            HIP_TRY(launch_attention(false, packed + h2x_off(l), xc, hn, w.nbr, w.deg, lig_flag, gen_flag, w.e_w, n_nodes,
                                     w.P, w.Qt, w.q, xn, nullptr, w.act, w.act_count, w.rf_list[0], w.rf_count, s));
        } else {
            const int set = l & 1;
            if (l > 0) HIP_TRY(hipStreamWaitEvent(s, aux->join, 0));       // node stage of this layer (aux stream) done
            HIP_TRY(launch_edge_x2h_dual(packed + x2h_off(l), xc, hc, Pset[set], Qtset[set], qset[set], w.nbr, w.deg, lig_flag,
                                         gen_flag, w.e_w, n_nodes, hn, xl.pp, xl.pp_n, xl.gen, xl.gen_n, xl.full, s));
            if (l + 1 < num_layers) {
                const int *d2, *d2n, *s2, *s2n;
                layer_lists(l + 1, d2, d2n, s2, s2n);
                HIP_TRY(hipEventRecord(aux->fork, s));
                HIP_TRY(hipStreamWaitEvent(aux->s, aux->fork, 0));
                const X2HLists xn = x2h_lists(l + 1);
                HIP_TRY(launch_node_mfma(packed + x2h_off(l + 1), hn, lig_flag, n_nodes, Pset[set ^ 1], qset[set ^ 1],
                                         Qtset[set ^ 1], d2, d2n, s2, s2n, aux->s, true, xn.gen, xn.gen_n));
                HIP_TRY(hipEventRecord(aux->join, aux->s));
            }
            HIP_TRY(launch_attention(false, packed + h2x_off(l), xc, hn, w.nbr, w.deg, lig_flag, gen_flag, w.e_w, n_nodes,
                                     w.P3, w.Qt3, w.q3, xn, nullptr, w.act, w.act_count, w.rf_list[0], w.rf_count, s));
        }
        xc = xn;
        hc = hn;
    }
    }   // two-stream / serial schedules
    if (logits) {
        const float* c = packed + cls_off(num_layers);
        // pruned mode: logits are only defined on ligand rows, which are a subset of A1 (rf_list[0])
        const int* rows = prune ? w.rf_list[0] : nullptr;
        const int* n_rows = prune ? w.rf_count : nullptr;
        HIP_TRY(launch_node_gemm(hc, H, c + C_W0T, c + C_B0, w.P, H, n_nodes, H, 1, s, rows, n_rows));
        HIP_TRY(launch_node_gemm(w.P, H, c + C_W1T, c + cls_b1(num_classes), logits, num_classes, n_nodes,
                                 num_classes, 0, s, rows, n_rows));
    }
    return CBGX_OK;
}

int cbgx_unitransformer_forward(const float* packed, int num_layers, int num_classes, const float* x, const float* h,
                                const int32_t* graph_ptr, const uint8_t* lig_flag, const uint8_t* gen_flag,
                                int n_nodes, int n_graphs, float* x_out, float* h_out, float* logits, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return forward_impl(packed, num_layers, num_classes, x, h, graph_ptr, lig_flag, gen_flag, n_nodes, n_graphs, x_out,
                        h_out, logits, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, workspace, workspace_bytes,
                        stream);
}

int cbgx_unitransformer_forward_cached(const float* packed, int num_layers, int num_classes, const float* x,
                                       const float* h, const int32_t* graph_ptr, const uint8_t* lig_flag,
                                       const uint8_t* gen_flag, int n_nodes, int n_graphs, const float* static_h1,
                                       const float* static_h2, const int32_t* static_nbr, const int32_t* static_deg,
                                       const float* static_ew, const float* static_r32sq, float* x_out, float* h_out,
                                       float* logits, unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
    if (!static_h1 || !static_h2) return fail(CBGX_E_INVALID, "forward_cached: NULL static context");
    return forward_impl(packed, num_layers, num_classes, x, h, graph_ptr, lig_flag, gen_flag, n_nodes, n_graphs, x_out,
                        h_out, logits, static_h1, static_h2, static_nbr, static_deg, static_ew, static_r32sq, flags, workspace,
                        workspace_bytes, stream);
}

int cbgx_targetdiff_prologue(const float* x_lig, const float* c_lig, const int32_t* lig_rows, int n_lig, int num_classes,
                             const float* lig_emb_w, const float* lig_emb_b, const float* ind_w, const float* ind_b,
                             float* x, float* h, void* stream) {
    if (n_lig == 0) return CBGX_OK;
    if (n_lig < 0 || num_classes < 1 || num_classes > 32) return fail(CBGX_E_INVALID, "prologue: bad sizes");
    if (!x_lig || !c_lig || !lig_rows || !lig_emb_w || !lig_emb_b || !ind_w || !ind_b || !x || !h)
        return fail(CBGX_E_INVALID, "prologue: NULL pointer");
    HIP_TRY(launch_step_prologue(x_lig, c_lig, lig_rows, n_lig, num_classes, lig_emb_w, lig_emb_b, ind_w, ind_b, x, h,
                                 (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_targetdiff_epilogue(const float* x_den, const float* logits, const int32_t* lig_rows, const float* x_lig,
                             const float* c_lig, const uint8_t* gen_lig, int n_lig, int num_classes, int t,
                             int num_timesteps, const float* const* tables, const float* eps, const float* u,
                             float* x_next, float* c_next, int32_t* v_next, void* stream) {
    if (n_lig == 0) return CBGX_OK;
    if (n_lig < 0 || num_classes < 1 || num_classes > 32 || t < 0 || t >= num_timesteps)
        return fail(CBGX_E_INVALID, "epilogue: bad sizes (n_lig=%d C=%d t=%d T=%d)", n_lig, num_classes, t, num_timesteps);
    if (!x_den || !logits || !lig_rows || !x_lig || !c_lig || !gen_lig || !tables || !eps || !u || !x_next || !c_next)
        return fail(CBGX_E_INVALID, "epilogue: NULL pointer");
    for (int i = 0; i < 7; ++i)
        if (!tables[i]) return fail(CBGX_E_INVALID, "epilogue: table %d is NULL", i);
    HIP_TRY(launch_step_epilogue(x_den, logits, lig_rows, x_lig, c_lig, gen_lig, n_lig, num_classes, t, tables,
                                 (float)log((double)num_classes), eps, u, x_next, c_next, v_next, (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_targetdiff_step_boundary(const float* x_den, const float* logits, const int32_t* lig_rows, const float* x_lig,
                                  const float* c_lig, const uint8_t* gen_lig, int n_lig, int num_classes, int t,
                                  int num_timesteps, const float* const* tables, const float* eps, const float* u,
                                  float* x_next, float* c_next, const float* lig_emb_w, const float* lig_emb_b,
                                  const float* ind_w, const float* ind_b, float* x, float* h, void* stream) {
    if (n_lig == 0) return CBGX_OK;
    if (n_lig < 0 || num_classes < 1 || num_classes > 32 || t < 0 || t >= num_timesteps)
        return fail(CBGX_E_INVALID, "step_boundary: bad sizes (n_lig=%d C=%d t=%d T=%d)", n_lig, num_classes, t, num_timesteps);
    if (!x_den || !logits || !lig_rows || !x_lig || !c_lig || !gen_lig || !tables || !eps || !u || !x_next || !c_next ||
        !lig_emb_w || !lig_emb_b || !ind_w || !ind_b || !x || !h)
        return fail(CBGX_E_INVALID, "step_boundary: NULL pointer");
    for (int i = 0; i < 7; ++i)
        if (!tables[i]) return fail(CBGX_E_INVALID, "step_boundary: table %d is NULL", i);
    HIP_TRY(launch_step_boundary(x_den, logits, lig_rows, x_lig, c_lig, gen_lig, n_lig, num_classes, t, tables,
                                 (float)log((double)num_classes), eps, u, x_next, c_next, lig_emb_w, lig_emb_b, ind_w, ind_b, x, h,
                                 (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_diffbp_epilogue(const float* x_den, const float* x_com, const float* x_in, const float* logits,
                         const int32_t* lig_rows, const int32_t* lig_ptr, const float* x_lig, const float* c_lig,
                         const uint8_t* gen_lig, int n_lig, int n_graphs, int num_classes, int t, int num_timesteps,
                         const float* alphas_cumprod, const float* betas, int absorbing_state, const float* eps,
                         const float* u, float* x_next, float* c_next, void* stream) {
    if (n_lig == 0 || n_graphs == 0) return CBGX_OK;
    if (n_lig < 0 || n_graphs < 0 || num_classes < 1 || num_classes > 32 || t < 0 || t >= num_timesteps ||
        absorbing_state < 0 || absorbing_state >= num_classes)
        return fail(CBGX_E_INVALID, "diffbp_epilogue: bad sizes (n_lig=%d B=%d C=%d t=%d T=%d)", n_lig, n_graphs, num_classes, t,
                    num_timesteps);
    if (!x_den || !x_com || !x_in || !logits || !lig_rows || !lig_ptr || !x_lig || !c_lig || !gen_lig || !alphas_cumprod ||
        !betas || !eps || !u || !x_next || !c_next)
        return fail(CBGX_E_INVALID, "diffbp_epilogue: NULL pointer");
    HIP_TRY(launch_diffbp_epilogue(x_den, x_com, x_in, logits, lig_rows, lig_ptr, x_lig, c_lig, gen_lig, n_graphs, num_classes, t,
                                   num_timesteps, alphas_cumprod, betas, absorbing_state, eps, u, x_next, c_next,
                                   (hipStream_t)stream));
    return CBGX_OK;
}

int cbgx_diffsbdd_step(const float* x_den, const float* logits, const int32_t* graph_ptr, const int32_t* lig_rows,
                       const int32_t* lig_ptr, const uint8_t* lig_flag, const float* x_lig, const float* c_lig, int n_lig,
                       int n_graphs, int num_classes, float inv_alpha, float coef, float sigma, int update_positions,
                       int update_types, const float* eps_x, const float* eps_c, const float* lig_emb_w,
                       const float* lig_emb_b, const float* ind_w, const float* ind_b, float* x_next, float* c_next,
                       float* x, float* h, float* shift, float* frame_shift, void* stream) {
    if (n_lig == 0 || n_graphs == 0) return CBGX_OK;
    if (n_lig < 0 || n_graphs < 0 || num_classes < 1 || num_classes > 32)
        return fail(CBGX_E_INVALID, "diffsbdd_step: bad sizes (n_lig=%d B=%d C=%d)", n_lig, n_graphs, num_classes);
    if (!x_den || !logits || !graph_ptr || !lig_rows || !lig_ptr || !lig_flag || !x_lig || !c_lig || !lig_emb_w || !lig_emb_b ||
        !ind_w || !ind_b || !x_next || !c_next || !x || !h || (update_positions && !eps_x) || (update_types && !eps_c))
        return fail(CBGX_E_INVALID, "diffsbdd_step: NULL pointer");
    HIP_TRY(launch_diffsbdd_step(x_den, logits, graph_ptr, lig_rows, lig_ptr, lig_flag, x_lig, c_lig, n_graphs, num_classes,
                                 inv_alpha, coef, sigma, update_positions, update_types, eps_x, eps_c, lig_emb_w, lig_emb_b, ind_w,
                                 ind_b, x_next, c_next, x, h, shift, frame_shift, (hipStream_t)stream));
    return CBGX_OK;
}

// ---- trajectory-resident variants (one captured hipGraph can then be replayed for every step) ---------------------
int cbgx_targetdiff_prologue_traj(const float* traj_x, const float* traj_c, const int32_t* t_dev, const int32_t* lig_rows,
                                  int n_lig, int num_classes, const float* lig_emb_w, const float* lig_emb_b,
                                  const float* ind_w, const float* ind_b, float* x, float* h, void* stream) {
    if (n_lig == 0) return CBGX_OK;
    if (n_lig < 0 || num_classes < 1 || num_classes > 32) return fail(CBGX_E_INVALID, "prologue_traj: bad sizes");
    if (!traj_x || !traj_c || !t_dev || !lig_rows || !lig_emb_w || !lig_emb_b || !ind_w || !ind_b || !x || !h)
        return fail(CBGX_E_INVALID, "prologue_traj: NULL pointer");
    HIP_TRY(launch_step_prologue(traj_x, traj_c, lig_rows, n_lig, num_classes, lig_emb_w, lig_emb_b, ind_w, ind_b, x, h,
                                 (hipStream_t)stream, t_dev));
    return CBGX_OK;
}

int cbgx_targetdiff_epilogue_traj(const float* x_den, const float* logits, const int32_t* lig_rows, float* traj_x,
                                  float* traj_c, const uint8_t* gen_lig, int n_lig, int num_classes, int32_t* t_dev,
                                  const float* const* tables, const float* eps, const float* u, void* stream) {
    if (n_lig == 0) return CBGX_OK;
    if (n_lig < 0 || num_classes < 1 || num_classes > 32) return fail(CBGX_E_INVALID, "epilogue_traj: bad sizes");
    if (!x_den || !logits || !lig_rows || !traj_x || !traj_c || !gen_lig || !t_dev || !tables || !eps || !u)
        return fail(CBGX_E_INVALID, "epilogue_traj: NULL pointer");
    for (int i = 0; i < 7; ++i)
        if (!tables[i]) return fail(CBGX_E_INVALID, "epilogue_traj: table %d is NULL", i);
    HIP_TRY(launch_step_epilogue(x_den, logits, lig_rows, traj_x, traj_c, gen_lig, n_lig, num_classes, 0, tables,
                                 (float)log((double)num_classes), eps, u, traj_x, traj_c, nullptr, (hipStream_t)stream,
                                 t_dev));
    return CBGX_OK;
}

}  // extern "C"

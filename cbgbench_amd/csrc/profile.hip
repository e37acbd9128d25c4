// Optional per-kernel timing with HIP events recorded on the stream the kernels are launched on.
// Enabled between cbgx_profile_begin() and cbgx_profile_end(); off by default (zero overhead: one
// relaxed load per launch).  Used by bench.py for the live roofline measurement.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/cbgx.h"
#include "kernels.h"

namespace cbgx {

struct Span { hipEvent_t a, b; int cls; };
static std::atomic<int> g_on{0};
static std::mutex g_mu;
static std::vector<Span> g_spans;      // recorded
static std::vector<Span> g_pool;       // pre-created, unused
static Span g_cur;
static bool g_open = false;

void profile_mark_begin(int cls, hipStream_t s) {
    if (!g_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pool.empty() || g_open) return;
    g_cur = g_pool.back();
    g_pool.pop_back();
    g_cur.cls = cls;
    if (hipEventRecord(g_cur.a, s) != hipSuccess) { g_pool.push_back(g_cur); return; }
    g_open = true;
}

bool profile_is_on() { return g_on.load(std::memory_order_relaxed) != 0; }

void profile_mark_end(hipStream_t s) {
    if (!g_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_open) return;
    (void)hipEventRecord(g_cur.b, s);
    g_spans.push_back(g_cur);
    g_open = false;
}

}  // namespace cbgx

using namespace cbgx;

extern "C" {

int cbgx_profile_begin(int max_launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (max_launches < 1) return CBGX_E_INVALID;
    while ((int)g_pool.size() < max_launches) {
        Span sp;
        if (hipEventCreate(&sp.a) != hipSuccess) return CBGX_E_HIP;
        if (hipEventCreate(&sp.b) != hipSuccess) return CBGX_E_HIP;
        sp.cls = -1;
        g_pool.push_back(sp);
    }
    g_on.store(1);
    return CBGX_OK;
}

int cbgx_profile_end(double* ms_by_class, int* launches_by_class, int num_classes) {
    g_on.store(0);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!ms_by_class || !launches_by_class || num_classes < K_NUM_CLASSES) return CBGX_E_INVALID;
    for (int c = 0; c < num_classes; ++c) { ms_by_class[c] = 0.0; launches_by_class[c] = 0; }
    int rc = CBGX_OK;
    for (Span& sp : g_spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) != hipSuccess || hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) {
            rc = CBGX_E_HIP;
        } else if (sp.cls >= 0 && sp.cls < num_classes) {
            ms_by_class[sp.cls] += ms;
            launches_by_class[sp.cls] += 1;
        }
        g_pool.push_back(sp);
    }
    g_spans.clear();
    g_open = false;
    return rc;
}

}  // extern "C"

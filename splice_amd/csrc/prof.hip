// Live timing of ONE kernel family for bench.py's roofline leg.
//
// splice_prof_begin(which) arms a family; every host call of that family opens a SpliceProfScope, and every kernel
// launched while the scope is open (SPLICE_LAUNCH, common.h) goes out through hipExtLaunchKernelGGL with a start / stop
// event pair of its own.  Such a pair holds the kernel's own begin / end time stamps -- the quantity rocprofv3's kernel
// trace reports -- instead of the ~5 us that two hipEventRecord calls queued around a launch add (round 2 subtracted a
// calibrated empty-pair cost and over-corrected by ~2 us against rocprofv3).  splice_prof_end() synchronises
// the pairs and returns the summed kernel time, the number of host calls (scopes) and the number of kernels.
//
// which: 1 fc1 GEMM fwd, 2 qkv GEMM fwd (layers 0..depth-2), 3 attention fwd, 4 fc2 GEMM fwd, 5 split-K dgrad GEMMs (fc1^T,
// qkv^T), 6 attention bwd, 7 generator (every conv / BatchNorm / upsample / weight-gradient launch of splice_gen_forward*
// and splice_gen_backward), 8 key self-similarity loss kernels (norms, S*, fused S / MSE / W, dK),
// 9 proj GEMM fwd.
#include <vector>

#include "common.h"

int g_splice_prof_open = 0;

namespace {
struct ProfState {
    int which = 0;
    std::vector<hipEvent_t> ev;   // start / stop pairs
    size_t used = 0;
    int calls = 0;
};
ProfState g_prof;
}   // namespace

bool splice_prof_take(hipEvent_t* start, hipEvent_t* stop) {
    if (g_splice_prof_open <= 0) return false;
    if (g_prof.used + 2 > g_prof.ev.size()) {
        for (int i = 0; i < 512; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return false;
            g_prof.ev.push_back(e);
        }
    }
    *start = g_prof.ev[g_prof.used];
    *stop = g_prof.ev[g_prof.used + 1];
    g_prof.used += 2;
    return true;
}

SpliceProfScope::SpliceProfScope(int which) : on(which != 0 && g_prof.which == which) {
    if (!on) return;
    ++g_splice_prof_open;
    ++g_prof.calls;
}
SpliceProfScope::~SpliceProfScope() {
    if (on) --g_splice_prof_open;
}

extern "C" {

int splice_prof_begin(int which) {
    g_prof.which = which;
    g_prof.used = 0;
    g_prof.calls = 0;
    g_splice_prof_open = 0;
    return SPLICE_OK;
}
int splice_prof_active(void) { return g_prof.which != 0; }

int splice_prof_end_ex(float* total_ms, int* calls, int* kernels) {
    float tot = 0.f;
    int rc = SPLICE_OK;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) {
            rc = SPLICE_ERR_HIP;
            break;
        }
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (calls) *calls = g_prof.calls;
    if (kernels) *kernels = (int)(g_prof.used / 2);
    g_prof.which = 0;
    g_prof.used = 0;
    g_prof.calls = 0;
    g_splice_prof_open = 0;
    return rc;
}
// launches = host calls of the family (a call may launch more than one kernel, e.g. the two-launch attention backward)
int splice_prof_end(float* total_ms, int* launches) { return splice_prof_end_ex(total_ms, launches, nullptr); }
}

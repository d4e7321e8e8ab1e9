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
// 9 proj GEMM fwd, 10 LayerNorm forward + backward (the backward also sums the split-K slabs of the dgrad in front of it),
// 11 the bf16-output dgrad GEMMs (fc2^T x GELU', proj^T + delta row dots).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

int g_splice_prof_open = 0;
#ifdef SPLICE_DEV_SWITCHES
unsigned long long g_splice_dev_skip = getenv("SPLICE_DEV_SKIP") ? strtoull(getenv("SPLICE_DEV_SKIP"), nullptr, 0) : 0ull;
int g_splice_dev_region = 0, g_splice_dev_steps = 0;
#endif

namespace {
struct ProfState {
    int which = 0;
    std::vector<hipEvent_t> ev;   // start / stop pairs
    std::vector<const char*> name;   // per pair: the kernel expression of its SPLICE_LAUNCH (a string literal)
    std::vector<double> flops, bytes;   // per pair: algorithmic work noted by the launcher (0 = not noted)
    size_t used = 0;
    int calls = 0;
    double next_flops = 0, next_bytes = 0;
};
ProfState g_prof;
}   // namespace

void splice_prof_note(double flops, double bytes) {
    if (g_splice_prof_open <= 0) return;
    g_prof.next_flops = flops;
    g_prof.next_bytes = bytes;
}
bool splice_prof_take(hipEvent_t* start, hipEvent_t* stop, const char* kernel_name) {
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
    const size_t pair = g_prof.used / 2;
    if (g_prof.name.size() <= pair) { g_prof.name.resize(pair + 512); g_prof.flops.resize(pair + 512); g_prof.bytes.resize(pair + 512); }
    g_prof.name[pair] = kernel_name;
    g_prof.flops[pair] = g_prof.next_flops;
    g_prof.bytes[pair] = g_prof.next_bytes;
    g_prof.next_flops = g_prof.next_bytes = 0;
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

// detail (may be NULL): one line per distinct kernel of the family, "name\tlaunches\ttotal ms\talgorithmic FLOPs\talgorithmic bytes\n" (the last two are the
// sums the launchers noted, 0 where none did), longest first, cut to detail_len - 1 bytes
int splice_prof_end_detail(float* total_ms, int* calls, int* kernels, char* detail, int detail_len) {
    float tot = 0.f;
    int rc = SPLICE_OK;
    struct Agg { const char* name; int n; double ms, flops, bytes; };
    std::vector<Agg> agg;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) {
            rc = SPLICE_ERR_HIP;
            break;
        }
        tot += ms;
        if (detail) {
            const char* nm = g_prof.name[i / 2];
            size_t k = 0;
            while (k < agg.size() && strcmp(agg[k].name, nm) != 0) ++k;
            if (k == agg.size()) agg.push_back(Agg{nm, 0, 0.0, 0.0, 0.0});
            agg[k].n++; agg[k].ms += ms; agg[k].flops += g_prof.flops[i / 2]; agg[k].bytes += g_prof.bytes[i / 2];
        }
    }
    if (detail && detail_len > 0) {
        std::sort(agg.begin(), agg.end(), [](const Agg& a, const Agg& b) { return a.ms > b.ms; });
        int pos = 0;
        detail[0] = 0;
        for (const Agg& g : agg) {
            const int w = snprintf(detail + pos, (size_t)(detail_len - pos), "%s\t%d\t%.6f\t%.6g\t%.6g\n", g.name, g.n, g.ms, g.flops, g.bytes);
            if (w < 0 || w >= detail_len - pos) { detail[pos] = 0; break; }
            pos += w;
        }
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
int splice_prof_end_ex(float* total_ms, int* calls, int* kernels) { return splice_prof_end_detail(total_ms, calls, kernels, nullptr, 0); }
// launches = host calls of the family (a call may launch more than one kernel, e.g. the two-launch attention backward)
int splice_prof_end(float* total_ms, int* launches) { return splice_prof_end_ex(total_ms, launches, nullptr); }
}

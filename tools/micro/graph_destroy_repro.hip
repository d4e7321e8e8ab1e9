// a stand-alone model of the launch pattern under which long train_model loops died inside the HIP runtime's completion-handler
// thread (profiles/r04_graph_drop_crash.txt), without any code of the library.  What the step engine did at the time, per crop-size change:
//   stream-capture ~600 small kernel nodes on TWO streams (fork / join through events), instantiate, replay ONCE, hipStreamSynchronize,
//   hipGraphExecDestroy at once -- while eager two-stream steps (the same kernels, fork / join events re-recorded every ~2 ms) keep both streams busy.
// Arms (argv[1]):
//   destroy   the round-4 pattern: instantiate -> replay once -> synchronize -> destroy immediately
//   grave     destroy >= 1.5 s later (round 4's graveyard)
//   update    never destroy: ONE executable, hipGraphExecUpdate from every new capture (round 5's policy)
// argv[2] = seconds to run (default 120), argv[3] = kernel nodes per captured step (default 600).  The kernels' launch geometry and arguments change from
// capture to capture the way crop sizes change them (grid sizes, pointers stay), so the update arm updates node parameters, not topology.
// Exit code 0 = ran to the end; a SIGSEGV of the runtime's handler thread kills the process (run under `timeout`, rc in the log).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/graph_destroy_repro.hip -o build/graph_destroy_repro && timeout 300 build/graph_destroy_repro destroy 120
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)

__global__ void work_kernel(float* p, int n, float a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * a + 1.0f;
}

struct Ctx {
    hipStream_t s0, s1;
    hipEvent_t fork[2], join[2];
    float *a, *b;
    int nodes;
};

// one "step": two chains on two streams, forked and joined twice (the generator / ViT branches of the step engine)
static void step(Ctx& c, int n) {
    const int half = c.nodes / 4;
    for (int part = 0; part < 2; ++part) {
        CK(hipEventRecord(c.fork[part], c.s0));
        CK(hipStreamWaitEvent(c.s1, c.fork[part], 0));
        for (int k = 0; k < half; ++k) {
            hipLaunchKernelGGL(work_kernel, dim3((n + 255) / 256), dim3(256), 0, c.s0, c.a, n, 0.999f);
            hipLaunchKernelGGL(work_kernel, dim3((n + 255) / 256), dim3(256), 0, c.s1, c.b, n, 0.999f);
        }
        CK(hipEventRecord(c.join[part], c.s1));
        CK(hipStreamWaitEvent(c.s0, c.join[part], 0));
    }
}

int main(int argc, char** argv) {
    const char* arm = argc > 1 ? argv[1] : "destroy";
    const double seconds = argc > 2 ? atof(argv[2]) : 120.0;
    Ctx c;
    c.nodes = argc > 3 ? atoi(argv[3]) : 600;
    CK(hipStreamCreateWithFlags(&c.s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c.s1, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        CK(hipEventCreateWithFlags(&c.fork[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&c.join[i], hipEventDisableTiming));
    }
    const int NMAX = 224 * 224 * 16;
    CK(hipMalloc(&c.a, NMAX * sizeof(float)));
    CK(hipMalloc(&c.b, NMAX * sizeof(float)));
    CK(hipMemset(c.a, 0, NMAX * sizeof(float)));
    CK(hipMemset(c.b, 0, NMAX * sizeof(float)));
    const bool grave = !strcmp(arm, "grave"), update = !strcmp(arm, "update");
    struct Dead { hipGraphExec_t e; std::chrono::steady_clock::time_point t; };
    std::deque<Dead> graveyard;
    hipGraphExec_t keep = nullptr;
    long captures = 0, replays = 0, eager = 0, destroyed = 0, updated = 0, refused = 0;
    unsigned rng = 12345u;
    const auto t0 = std::chrono::steady_clock::now();
    auto now = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    double next_print = 10.0;
    while (now() < seconds) {
        // a run of eager steps at changing sizes (random crops: a new size nearly every step) ...
        const int run = 20 + (int)((rng = rng * 1664525u + 1013904223u) >> 28);
        for (int i = 0; i < run; ++i) {
            const int side = 213 + (int)((rng = rng * 1664525u + 1013904223u) >> 24) % 12;
            step(c, side * side * 16);
            ++eager;
        }
        // ... then the size repeats: capture, instantiate (or update), replay ONCE, drop
        const int side = 213 + (int)((rng = rng * 1664525u + 1013904223u) >> 24) % 12;
        hipGraph_t g;
        CK(hipStreamBeginCapture(c.s0, hipStreamCaptureModeThreadLocal));
        step(c, side * side * 16);
        CK(hipStreamEndCapture(c.s0, &g));
        ++captures;
        hipGraphExec_t ex = nullptr;
        if (update && keep) {
            CK(hipDeviceSynchronize());
            hipGraphNode_t bad = nullptr;
            hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
            if (hipGraphExecUpdate(keep, g, &bad, &res) == hipSuccess && res == hipGraphExecUpdateSuccess) { ex = keep; ++updated; }
            else { (void)hipGetLastError(); ++refused; }
        }
        if (!ex) {
            CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            if (update) keep = ex;
        }
        CK(hipGraphDestroy(g));
        CK(hipGraphLaunch(ex, c.s0));
        ++replays;
        // the next eager step is already queued behind the replay when the executable is dropped (the engine launches ahead of the GPU)
        step(c, (side - 1) * (side - 1) * 16);
        ++eager;
        if (!update) {
            CK(hipStreamSynchronize(c.s0));
            if (grave) graveyard.push_back({ex, std::chrono::steady_clock::now()});
            else { CK(hipGraphExecDestroy(ex)); ++destroyed; }
        }
        while (!graveyard.empty() && std::chrono::duration<double>(std::chrono::steady_clock::now() - graveyard.front().t).count() > 1.5) {
            CK(hipGraphExecDestroy(graveyard.front().e));
            graveyard.pop_front();
            ++destroyed;
        }
        if (now() > next_print) {
            printf("[%s] %.0f s: %ld eager steps, %ld captures, %ld replays, %ld destroyed, %ld updated, %ld refused\n", arm, now(), eager, captures, replays,
                   destroyed, updated, refused);
            fflush(stdout);
            next_print += 10.0;
        }
    }
    CK(hipDeviceSynchronize());
    printf("[%s] done after %.0f s: %ld eager steps, %ld captures, %ld replays, %ld destroyed, %ld updated, %ld refused -- no fault\n", arm, now(), eager, captures,
           replays, destroyed, updated, refused);
    return 0;
}

// Does a HIP graph keep producer -> consumer visibility between kernel nodes whose workgroups land on different XCDs?
// Each round: P (grid of G blocks) reads X[0] so that every XCD's L2 holds the line; W (1 block) writes X[0] = round;
// R (G blocks) reads X[0] and records it per block.  Under stream semantics every R block must see `round`.
// Run once with plain stream launches and once with the same sequence captured into a graph and replayed.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_coherence.hip -o tools/graph_coherence && ./tools/graph_coherence
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void kP(const int* X, int* sink) { if (threadIdx.x == 0) sink[blockIdx.x] = X[0]; }
__global__ void kW(int* X, const int* round) { if (threadIdx.x == 0 && blockIdx.x == 0) X[0] = *round; }
__global__ void kR(const int* X, int* Y, const int* round, int* bad) {
    if (threadIdx.x == 0) { int v = X[0]; Y[blockIdx.x] = v; if (v != *round) atomicAdd(bad, 1); }
}
__global__ void kInc(int* round) { if (threadIdx.x == 0 && blockIdx.x == 0) *round += 1; }
int main() {
    const int G = 64;
    int *X, *sink, *Y, *round, *bad;
    CHECK(hipMalloc(&X, 4096)); CHECK(hipMalloc(&sink, G * 4)); CHECK(hipMalloc(&Y, G * 4));
    CHECK(hipMalloc(&round, 4)); CHECK(hipMalloc(&bad, 4));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    for (int mode = 0; mode < 2; ++mode) {
        CHECK(hipMemset(X, 0, 4096)); CHECK(hipMemset(round, 0, 4)); CHECK(hipMemset(bad, 0, 4));
        CHECK(hipDeviceSynchronize());
        auto seq = [&](hipStream_t st) {
            hipLaunchKernelGGL(kInc, dim3(1), dim3(64), 0, st, round);
            hipLaunchKernelGGL(kP, dim3(G), dim3(64), 0, st, X, sink);
            hipLaunchKernelGGL(kW, dim3(1), dim3(64), 0, st, X, round);
            hipLaunchKernelGGL(kR, dim3(G), dim3(64), 0, st, X, Y, round, bad);
        };
        const int rounds = 2000;
        if (mode == 0) {
            for (int i = 0; i < rounds; ++i) seq(s);
        } else {
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 10; ++i) seq(s);                 // 10 rounds per graph
            CHECK(hipStreamEndCapture(s, &g));
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < rounds / 10; ++i) CHECK(hipGraphLaunch(ge, s));
        }
        CHECK(hipStreamSynchronize(s));
        int hbad = 0, hround = 0;
        CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hround, round, 4, hipMemcpyDeviceToHost));
        printf("%s: %d rounds, %d stale reads out of %d\n", mode == 0 ? "stream launches" : "graph replays  ", hround, hbad, hround * G);
    }
    return 0;
}

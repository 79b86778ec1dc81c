// Is a captured hipMemsetAsync ordered before the kernel behind it when the graph is replayed?  Models the
// multi-workgroup reduction torch launches for `mean` (ATen Reduce.cuh): per call a 4-byte semaphore is zeroed with
// hipMemsetAsync, every block publishes a partial sum, bumps the semaphore and the block that sees G - 1 sums the
// partials.  If the memset of one replayed node does not take effect before its kernel, a block fires early and sums
// incomplete partials -- the "negative MSE" recorded in DESIGN.md.  The sequence per round: [dirty the semaphore the way a
// finished reduction leaves it] -> memset(sem, 0) -> reduce kernel -> check kernel.
// Variants: plain stream launches, a graph of 1 round, a graph of 25 rounds; with the scratch memory from hipMalloc
// and from hipMallocAsync inside the capture (graph-owned memory, as the torch caching allocator's private pool is).
//   hipcc --offload-arch=gfx950 -O3 tools/graph_memset_order.hip -o tools/graph_memset_order && ./tools/graph_memset_order
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int G = 256, T = 256, N = G * T * 8;

__global__ void k_reduce(const float* x, float* partial, int* sem, float* out, int* early) {
    __shared__ float sh[T];
    __shared__ bool last;
    float acc = 0.f;
    for (int i = blockIdx.x * T + threadIdx.x; i < N; i += G * T) acc += x[i] * x[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = T / 2; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = sh[0];
        __threadfence();
        const int prev = atomicAdd(sem, 1);
        last = (prev == G - 1);
        if (prev >= G) atomicAdd(early, 1);                     // the semaphore was not zero when this launch began
    }
    __syncthreads();
    if (last) {
        float a = 0.f;
        for (int i = threadIdx.x; i < G; i += T) a += ((volatile float*)partial)[i];
        sh[threadIdx.x] = a;
        __syncthreads();
        for (int s = T / 2; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
        if (threadIdx.x == 0) out[0] = sh[0] / N;
    }
}
__global__ void k_poison(float* partial, float* out) {           // between rounds: what a wrong reduction would pick up
    if (blockIdx.x == 0) { for (int i = threadIdx.x; i < G; i += T) partial[i] = -1e30f; if (threadIdx.x == 0) out[0] = -7.f; }
}
__global__ void k_check(const float* out, float expect, int* bad, const int* sem) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float v = out[0];
        if (!(fabsf(v - expect) <= 1e-3f * expect)) atomicAdd(bad, 1);
        if (*sem != G) atomicAdd(bad + 1, 1);
    }
}

int main() {
    float *x, *partial, *out; int *sem, *bad, *early, *sem2;
    CHECK(hipMalloc(&x, N * 4)); CHECK(hipMalloc(&partial, G * 4)); CHECK(hipMalloc(&out, 4)); CHECK(hipMalloc(&sem, 4));
    CHECK(hipMalloc(&sem2, 4));
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&early, 4));
    float* hx = new float[N];
    double ss = 0;
    for (int i = 0; i < N; ++i) { hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f; ss += (double)hx[i] * hx[i]; }
    const float expect = (float)(ss / N);
    CHECK(hipMemcpy(x, hx, N * 4, hipMemcpyHostToDevice));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    auto round = [&](hipStream_t st) {
        hipLaunchKernelGGL(k_poison, dim3(1), dim3(T), 0, st, partial, out);
        hipMemsetAsync(sem, 0, 4, st);                           // the semaphore still holds G from the previous round
        hipLaunchKernelGGL(k_reduce, dim3(G), dim3(T), 0, st, x, partial, sem, out, early);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, st, out, expect, bad, sem);
    };
    // second arrangement: TWO reductions per round with DIFFERENT semaphores (two memset nodes with different
    // destinations in one graph) -- a memset node that replays with another node's parameters shows up here only
    auto round2 = [&](hipStream_t st) {
        hipLaunchKernelGGL(k_poison, dim3(1), dim3(T), 0, st, partial, out);
        hipMemsetAsync(sem, 0, 4, st);
        hipLaunchKernelGGL(k_reduce, dim3(G), dim3(T), 0, st, x, partial, sem, out, early);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, st, out, expect, bad, sem);
        hipLaunchKernelGGL(k_poison, dim3(1), dim3(T), 0, st, partial, out);
        hipMemsetAsync(sem2, 0, 4, st);
        hipLaunchKernelGGL(k_reduce, dim3(G), dim3(T), 0, st, x, partial, sem2, out, early);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, st, out, expect, bad, sem2);
    };
    const int total = 20000;
    for (int mode = 0; mode < 5; ++mode) {
        CHECK(hipMemset(bad, 0, 8)); CHECK(hipMemset(early, 0, 4)); CHECK(hipMemset(sem, 0, 4)); CHECK(hipMemset(sem2, 0, 4));
        CHECK(hipDeviceSynchronize());
        const bool two = mode >= 3;
        const int per = mode == 0 ? 0 : (mode == 1 ? 1 : (mode == 2 ? 25 : (mode == 3 ? 1 : 25)));
        if (per == 0) {
            for (int i = 0; i < total; ++i) round(s);
        } else {
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < per; ++i) { if (two) round2(s); else round(s); }
            CHECK(hipStreamEndCapture(s, &g));
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < total / per; ++i) CHECK(hipGraphLaunch(ge, s));
            CHECK(hipStreamSynchronize(s));
            CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
        }
        CHECK(hipStreamSynchronize(s));
        int hb[2], he;
        CHECK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&he, early, 4, hipMemcpyDeviceToHost));
        printf("%-28s %d rounds: wrong results %d, semaphore != G after the kernel %d, blocks that found the semaphore dirty %d\n",
               mode == 0 ? "stream launches" : (mode == 1 ? "graph of 1 round, replayed" : (mode == 2 ? "graph of 25 rounds, replayed" :
               (mode == 3 ? "2 semaphores, graph of 1 round" : "2 semaphores, graph of 25 rounds"))), total, hb[0], hb[1], he);
    }
    return 0;
}

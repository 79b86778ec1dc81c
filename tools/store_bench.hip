// Micro-benchmark: which launch structure reaches the HBM write ceiling for 16 KiB-per-item f32 images?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st(float* p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = v;
}
// A: persistent, wave w takes items w, w+W, ...
template <bool NT> __global__ __launch_bounds__(256) void kA(float* out, int items) {
    int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, W = (gridDim.x * blockDim.x) >> 6;
    f32x4 z = {0.f, 0.f, 0.f, (float)lane};
    for (int it = wave; it < items; it += W) {
        float* img = out + (size_t)it * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) st<NT>(img + r * 256 + lane * 4, z);
    }
}
// B: persistent, wave w takes a contiguous chunk of items
template <bool NT> __global__ __launch_bounds__(256) void kB(float* out, int items) {
    int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, W = (gridDim.x * blockDim.x) >> 6;
    int per = (items + W - 1) / W, lo = wave * per, hi = lo + per < items ? lo + per : items;
    f32x4 z = {0.f, 0.f, 0.f, (float)lane};
    for (int it = lo; it < hi; ++it) {
        float* img = out + (size_t)it * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) st<NT>(img + r * 256 + lane * 4, z);
    }
}
// C: persistent, a 256-thread block shares one item (4 KiB per wave)
template <bool NT> __global__ __launch_bounds__(256) void kC(float* out, int items) {
    int t = threadIdx.x;
    f32x4 z = {0.f, 0.f, 0.f, (float)t};
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        float* img = out + (size_t)it * 4096;
#pragma unroll
        for (int r = 0; r < 4; ++r) st<NT>(img + r * 1024 + t * 4, z);
    }
}
// D: non-persistent, one wave per item
template <bool NT> __global__ __launch_bounds__(256) void kD(float* out, int items) {
    int lane = threadIdx.x & 63, it = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (it >= items) return;
    f32x4 z = {0.f, 0.f, 0.f, (float)lane};
    float* img = out + (size_t)it * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) st<NT>(img + r * 256 + lane * 4, z);
}
// E: plain linear fill (grid-stride, 16 B per thread), the torch-fill shape
template <bool NT> __global__ __launch_bounds__(256) void kE(float* out, int items) {
    size_t n4 = (size_t)items * 1024, stride = (size_t)gridDim.x * blockDim.x;
    f32x4 z = {0.f, 0.f, 0.f, 1.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) st<NT>(out + i * 4, z);
}
// F: non-persistent, one wave per 16/PARTS KiB piece, pieces in linear order
template <int PARTS> __global__ __launch_bounds__(256) void kF(float* out, long pieces) {
    int lane = threadIdx.x & 63; long pc = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pc >= pieces) return;
    f32x4 z = {0.f, 0.f, 0.f, (float)lane};
    float* p = out + pc * (4096 / PARTS);
#pragma unroll
    for (int r = 0; r < 16 / PARTS; ++r) st<false>(p + r * 256 + lane * 4, z);
}
// G: like F but each wave first does a dependent load (models "metadata before the stores") and ~2 us of ALU work
template <int PARTS> __global__ __launch_bounds__(256) void kG(float* out, long pieces, const int* meta) {
    int lane = threadIdx.x & 63; long pc = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pc >= pieces) return;
    int m = meta[pc & 1023];
    double acc = lane * 0.5 + m;
    for (int i = 0; i < 300; ++i) acc = acc * 1.0000001 + 0.5;      // ~ the half-plane tests of one image quarter
    f32x4 z = {0.f, 0.f, (float)(acc > 1e300), (float)lane};
    float* p = out + pc * (4096 / PARTS);
#pragma unroll
    for (int r = 0; r < 16 / PARTS; ++r) st<false>(p + r * 256 + lane * 4, z);
}
// H: non-persistent, a 256-thread block owns 4 consecutive items: every wave does the per-item prologue of G for ITS item,
// the results meet in LDS, then the block writes the 64 KiB of its 4 items in linear order (4 KiB per store step)
__global__ __launch_bounds__(256) void kH(float* out, long items, const int* meta) {
    __shared__ unsigned long long bits[4][64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6; long it = (long)blockIdx.x * 4 + w;
    int m = meta[it & 1023];
    double acc = lane * 0.5 + m;
    for (int i = 0; i < 300; ++i) acc = acc * 1.0000001 + 0.5;
    bits[w][lane] = (acc > 1e300) ? ~0ull : (unsigned long long)lane;
    __syncthreads();
    float* p = out + (size_t)blockIdx.x * 4 * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int e = r * 1024 + threadIdx.x * 4;                 // element index inside the block's 64 KiB
        unsigned long long b = bits[e >> 12][(e >> 6) & 63];
        unsigned nib = (unsigned)(b >> (e & 63)) & 15u;
        f32x4 z = {(float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)((nib >> 3) & 1u)};
        st<false>(p + e, z);
    }
}
// I: like D (wave per item, prologue of G) but the expansion reads its bits from LDS like H (isolates the store order)
__global__ __launch_bounds__(256) void kI(float* out, long items, const int* meta) {
    __shared__ unsigned long long bits[4][64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6; long it = (long)blockIdx.x * 4 + w;
    int m = meta[it & 1023];
    double acc = lane * 0.5 + m;
    for (int i = 0; i < 300; ++i) acc = acc * 1.0000001 + 0.5;
    bits[w][lane] = (acc > 1e300) ? ~0ull : (unsigned long long)lane;
    float* p = out + (size_t)it * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int e = r * 256 + lane * 4;
        unsigned long long b = bits[w][(e >> 6) & 63];
        unsigned nib = (unsigned)(b >> (e & 63)) & 15u;
        f32x4 z = {(float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)((nib >> 3) & 1u)};
        st<false>(p + e, z);
    }
}
// J: expansion-only kernel: one wave per 16/PARTS KiB piece in linear order; the piece's 64/PARTS row masks come through
// scalar loads (wave-uniform address) from a bit-packed array written by an earlier kernel
template <int PARTS> __global__ __launch_bounds__(256) void kJ(float* out, long pieces, const unsigned long long* bits) {
    const int lane = threadIdx.x & 63;
    const long pc = __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (pc >= pieces) return;
    constexpr int ROWS = 64 / PARTS;                         // rows of this piece, 4 per store
    const unsigned long long* b = bits + pc * ROWS;           // uniform
    float* p = out + pc * (4096 / PARTS);
    const int sub = lane >> 4, col4 = (lane & 15) * 4;
#pragma unroll
    for (int r = 0; r < ROWS; r += 4) {
        unsigned long long m0 = b[r], m1 = b[r + 1], m2 = b[r + 2], m3 = b[r + 3];
        unsigned long long m = sub == 0 ? m0 : (sub == 1 ? m1 : (sub == 2 ? m2 : m3));
        unsigned nib = (unsigned)(m >> col4) & 15u;
        f32x4 z = {(float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)((nib >> 3) & 1u)};
        st<false>(p + (r / 4) * 256 + lane * 4, z);
    }
}
// K: expansion-only like J, but the piece's row masks arrive through ONE vector load (lane r < ROWS loads row r) and
// are handed round with a cross-lane shuffle, so a wave pays one memory latency in front of its stores
template <int PARTS> __global__ __launch_bounds__(256) void kK(float* out, long pieces, const unsigned long long* bits) {
    const int lane = threadIdx.x & 63;
    const long pc = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pc >= pieces) return;
    constexpr int ROWS = 64 / PARTS;
    const unsigned long long mine = lane < ROWS ? bits[pc * ROWS + lane] : 0ull;
    float* p = out + pc * (4096 / PARTS);
    const int sub = lane >> 4, col4 = (lane & 15) * 4;
#pragma unroll
    for (int r = 0; r < ROWS; r += 4) {
        unsigned lo = __shfl((unsigned)mine, r + sub, 64), hi = __shfl((unsigned)(mine >> 32), r + sub, 64);
        unsigned long long m = ((unsigned long long)hi << 32) | lo;
        unsigned nib = (unsigned)(m >> col4) & 15u;
        f32x4 z = {(float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)((nib >> 3) & 1u)};
        st<false>(p + (r / 4) * 256 + lane * 4, z);
    }
}
// L: "zeros first": the wave stores its whole 16 KiB image as zeros before it knows anything (pure stores), then does the
// dependent load + ALU prologue of G and overwrites only the ~3 KiB of row groups that hold pixels
__global__ __launch_bounds__(256) void kL(float* out, long items, const int* meta, int dirty_groups) {
    int lane = threadIdx.x & 63; long it = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (it >= items) return;
    float* p = out + it * 4096;
    f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) st<false>(p + r * 256 + lane * 4, zero);
    int m = meta[it & 1023];
    double acc = lane * 0.5 + m;
    for (int i = 0; i < 300; ++i) acc = acc * 1.0000001 + 0.5;
    f32x4 z = {0.f, 1.f, (float)(acc > 1e300), (float)lane};
    __builtin_amdgcn_s_waitcnt(0);                              // the zero stores of this wave are complete (vmcnt = 0)
    int g0 = (int)((it * 7) & 15);
    for (int g = 0; g < dirty_groups; ++g) st<false>(p + ((g0 + g) & 15) * 256 + lane * 4, z);
}
// M: like G (dependent load + ALU, then 16/PARTS KiB of stores) with the ALU chain a parameter: models PARTS waves per
// image that each redo the (short) half-plane tests of the block's row window and store only their own rows
template <int PARTS> __global__ __launch_bounds__(256) void kM(float* out, long pieces, const int* meta, int alu) {
    int lane = threadIdx.x & 63; long pc = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pc >= pieces) return;
    int m = meta[(pc / PARTS) & 1023];
    double acc = lane * 0.5 + m;
    for (int i = 0; i < alu; ++i) acc = acc * 1.0000001 + 0.5;
    f32x4 z = {0.f, 0.f, (float)(acc > 1e300), (float)lane};
    float* p = out + pc * (4096 / PARTS);
#pragma unroll
    for (int r = 0; r < 16 / PARTS; ++r) st<false>(p + r * 256 + lane * 4, z);
}
template <typename F> void run(const char* name, F launch, float* buf, int items) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, (double)items * 16384 / (ms * 1e-3) / 1e12);
}
int main() {
    int items = 262144;   // 4.29 GB
    float* buf; CHECK(hipMalloc(&buf, (size_t)items * 16384));
    for (int bpc : {4, 8, 16}) {
        int grid = 256 * bpc;
        printf("-- persistent grid %d blocks (%d per CU)\n", grid, bpc);
        run("A strided items", [&] { hipLaunchKernelGGL(kA<false>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("A strided items nt", [&] { hipLaunchKernelGGL(kA<true>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("B contiguous chunks", [&] { hipLaunchKernelGGL(kB<false>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("B contiguous chunks nt", [&] { hipLaunchKernelGGL(kB<true>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("C block per item", [&] { hipLaunchKernelGGL(kC<false>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("C block per item nt", [&] { hipLaunchKernelGGL(kC<true>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("E linear fill", [&] { hipLaunchKernelGGL(kE<false>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
        run("E linear fill nt", [&] { hipLaunchKernelGGL(kE<true>, dim3(grid), dim3(256), 0, 0, buf, items); }, buf, items);
    }
    printf("-- non-persistent\n");
    run("D wave per item", [&] { hipLaunchKernelGGL(kD<false>, dim3(items / 4), dim3(256), 0, 0, buf, items); }, buf, items);
    run("D wave per item nt", [&] { hipLaunchKernelGGL(kD<true>, dim3(items / 4), dim3(256), 0, 0, buf, items); }, buf, items);
    run("F wave per 8 KiB", [&] { hipLaunchKernelGGL(kF<2>, dim3(items * 2 / 4), dim3(256), 0, 0, buf, (long)items * 2); }, buf, items);
    run("F wave per 4 KiB", [&] { hipLaunchKernelGGL(kF<4>, dim3(items * 4 / 4), dim3(256), 0, 0, buf, (long)items * 4); }, buf, items);
    run("F wave per 2 KiB", [&] { hipLaunchKernelGGL(kF<8>, dim3(items * 8 / 4), dim3(256), 0, 0, buf, (long)items * 8); }, buf, items);
    run("F wave per 1 KiB", [&] { hipLaunchKernelGGL(kF<16>, dim3(items * 16 / 4), dim3(256), 0, 0, buf, (long)items * 16); }, buf, items);
    int* meta; CHECK(hipMalloc(&meta, 4096)); CHECK(hipMemset(meta, 0, 4096));
    run("G 16 KiB + load + ALU", [&] { hipLaunchKernelGGL(kG<1>, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta); }, buf, items);
    run("G 4 KiB + load + ALU", [&] { hipLaunchKernelGGL(kG<4>, dim3(items), dim3(256), 0, 0, buf, (long)items * 4, meta); }, buf, items);
    run("G 1 KiB + load + ALU", [&] { hipLaunchKernelGGL(kG<16>, dim3(items * 4), dim3(256), 0, 0, buf, (long)items * 16, meta); }, buf, items);
    run("H block of 4 items, linear", [&] { hipLaunchKernelGGL(kH, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta); }, buf, items);
    run("I wave per item, LDS bits", [&] { hipLaunchKernelGGL(kI, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta); }, buf, items);
    run("D wave per item (again)", [&] { hipLaunchKernelGGL(kD<false>, dim3(items / 4), dim3(256), 0, 0, buf, items); }, buf, items);
    unsigned long long* bitsd; CHECK(hipMalloc(&bitsd, (size_t)items * 512)); CHECK(hipMemset(bitsd, 0x11, (size_t)items * 512));
    run("J expand 1 KiB (s_load bits)", [&] { hipLaunchKernelGGL(kJ<16>, dim3(items * 16 / 4), dim3(256), 0, 0, buf, (long)items * 16, bitsd); }, buf, items);
    run("J expand 2 KiB (s_load bits)", [&] { hipLaunchKernelGGL(kJ<8>, dim3(items * 8 / 4), dim3(256), 0, 0, buf, (long)items * 8, bitsd); }, buf, items);
    run("J expand 4 KiB (s_load bits)", [&] { hipLaunchKernelGGL(kJ<4>, dim3(items * 4 / 4), dim3(256), 0, 0, buf, (long)items * 4, bitsd); }, buf, items);
    run("J expand 16 KiB (s_load bits)", [&] { hipLaunchKernelGGL(kJ<1>, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, bitsd); }, buf, items);
    run("K expand 2 KiB (vector load)", [&] { hipLaunchKernelGGL(kK<8>, dim3(items * 8 / 4), dim3(256), 0, 0, buf, (long)items * 8, bitsd); }, buf, items);
    run("K expand 4 KiB (vector load)", [&] { hipLaunchKernelGGL(kK<4>, dim3(items * 4 / 4), dim3(256), 0, 0, buf, (long)items * 4, bitsd); }, buf, items);
    run("K expand 8 KiB (vector load)", [&] { hipLaunchKernelGGL(kK<2>, dim3(items * 2 / 4), dim3(256), 0, 0, buf, (long)items * 2, bitsd); }, buf, items);
    run("K expand 16 KiB (vector load)", [&] { hipLaunchKernelGGL(kK<1>, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, bitsd); }, buf, items);
    run("L zeros first + 3 dirty groups", [&] { hipLaunchKernelGGL(kL, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta, 3); }, buf, items);
    run("L zeros first + 0 dirty groups", [&] { hipLaunchKernelGGL(kL, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta, 0); }, buf, items);
    run("G 16 KiB + load + ALU (again)", [&] { hipLaunchKernelGGL(kG<1>, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta); }, buf, items);
    for (int alu : {0, 60, 150}) {
        char nm[64];
        snprintf(nm, sizeof(nm), "M 4 KiB + load + %d FMA", alu);
        run(nm, [&] { hipLaunchKernelGGL(kM<4>, dim3(items), dim3(256), 0, 0, buf, (long)items * 4, meta, alu); }, buf, items);
        snprintf(nm, sizeof(nm), "M 2 KiB + load + %d FMA", alu);
        run(nm, [&] { hipLaunchKernelGGL(kM<8>, dim3(items * 2), dim3(256), 0, 0, buf, (long)items * 8, meta, alu); }, buf, items);
        snprintf(nm, sizeof(nm), "M 16 KiB + load + %d FMA", alu);
        run(nm, [&] { hipLaunchKernelGGL(kM<1>, dim3(items / 4), dim3(256), 0, 0, buf, (long)items, meta, alu); }, buf, items);
    }
    run("E linear fill full grid", [&] { hipLaunchKernelGGL(kE<false>, dim3(items * 4), dim3(256), 0, 0, buf, items); }, buf, items);
    return 0;
}

// K11: the ConvBlock stacks of the conv Q-networks (robotoddler/models/cv.py:5-17: conv3x3 - ReLU - conv3x3 - ReLU - MaxPool2d(2),
// four of them in ConvNet, cv.py:41-73) for the TRAINING passes of train_policy_net (successor_dqn.py:157-277): forward,
// input gradient, weight / bias gradient and the pooling pair, hand-written for the f32 matrix cores.
//
// Why: at the CLI's replay batch of 32 one optimiser step of ConvNet is ~130 library launches of ~9 us (Winograd forward /
// backward-data, implicit-GEMM weight gradients wrapped in layout transposes, separate bias / ReLU / pool / bias-gradient
// passes: profiles/r04_b_conv_lockstep_kernels.txt) for ~10 GFLOP -- 1.15 ms where the arithmetic is worth ~0.1 ms.  Here a
// ConvBlock is 3 launches forward and 6 backward.
//
//   k_c3        conv3x3, padding 1, on square W x W images (W = 8, 16, 32, 64), any C_in, C_out a multiple of 16.  The same
//               kernel is the input gradient: dX = conv3x3(G, W^T flipped) -- the weight tensor is addressed through two
//               strides and a flip flag.  v_mfma_f32_16x16x4_f32: M = 16 pixels, N = 16 output channels, K = 4 input
//               channels of one tap; the input patch of a band (+ halo, 16 channels at a time) is staged in LDS, the weights
//               of a tap sit in registers.  Epilogues: raw | + bias, ReLU | x [mask > 0] (the ReLU of the layer below, whose
//               output is this gradient's destination).
//   k_c3_wgrad  dW[co][ci][tap] = sum_{n,y,x} G[n,co,y,x] X[n,ci,y+dy-1,x+dx-1]: M = 16 co, N = 16 ci, K = 4 pixels; both
//               operands from LDS, nine accumulator tiles (one per tap) per wave; the pixel range is split over workgroups
//               whose partial sums a second launch adds in a fixed order (deterministic: no atomics), together with the
//               bias gradient sum G.
//   k_maxpool2 / k_maxpool2_relu_bwd   MaxPool2d(2) and its gradient folded with the ReLU mask (first maximum in scan
//               order takes the gradient, as torch's max_pool2d does).
#include "bridges_device.h"

namespace bridges {

typedef float c3_f32x4 __attribute__((ext_vector_type(4)));

#define C3_EPI_RAW 0
#define C3_EPI_BIAS_RELU 1
#define C3_EPI_MASK 2

// band rows of a workgroup: 4 on the 16- to 64-pixel layers, the whole image on the 8-pixel ones -- at the CLI's batch of 32 that
// makes >= 512 workgroups for every layer but the last block's, and with the register cap of two waves per SIMD (C3_MIN_WAVES)
// two workgroups share a CU, one staging its patch while the other multiplies.  tools/c3_variants.sh (profiles/r04_c3_variants.txt),
// one optimiser step of the U-Net policy: 8 rows / 1 wave 1269 us, 8 / 2 1494 (spills), 4 / 1 1276, 4 / 2 1200.  (16-row bands
// had left half the chip idle on the 32- and 16-pixel layers.)
#ifndef C3_BAND_ROWS_WIDE
#define C3_BAND_ROWS_WIDE 4
#endif
__host__ __device__ constexpr int c3_band_rows(int W) { return W >= 16 ? C3_BAND_ROWS_WIDE : 8; }
// band of the weight-gradient kernel: two operand tiles must fit into 64 KB of static LDS
__host__ __device__ constexpr int c3_wgrad_rows(int W) { return W >= 64 ? 4 : (W == 32 ? 8 : W); }
// smallest size >= raw with size % 64 == rem (LDS plane strides that put the 4 k-lanes of an operand read on distinct banks)
__host__ __device__ constexpr int c3_pad(int raw, int rem) { return ((raw - rem + 63) / 64) * 64 + rem; }

// ---------------------------------------------------------------------------------------------------------------
// x [N, c_in, W, W] (in_mask, optional, same shape: x counts only where in_mask > 0), weights addressed as w[cin * w_sin + cout * w_sout + (flip ? 8 - tap : tap)]:
//   forward:        w = conv.weight [c_out, c_in, 3, 3]: w_sout = c_in * 9, w_sin = 9,  flip = 0
//   input gradient: x = G [N, C_out_layer, ..], "c_out" = C_in_layer:   w_sin = C_in_layer * 9, w_sout = 9, flip = 1
// grid = (N * bands, c_out / 16); a workgroup = one image band x 16 output channels.
// Input channels staged per pass: as many as 64 KB of LDS hold next to their weights -- the kernel is a chain of
// "global round trip, LDS, a few hundred MFMAs" per pass, and on the small images of the deep layers the round trips are
// what it costs (64 -> 128 channels on 8 x 8 images: 8 passes of 16 channels took 39 us, the arithmetic < 2).
__host__ __device__ constexpr int c3_stage(int W, int CIN_CHUNK) { return CIN_CHUNK < 16 ? CIN_CHUNK : (W <= 16 ? 32 : 16); }

#ifndef C3_MIN_WAVES
#define C3_MIN_WAVES 2
#endif
template <int W, int CIN_CHUNK, int EPI>
__global__ __launch_bounds__(256, C3_MIN_WAVES) void k_c3(const float* __restrict__ x, const float* __restrict__ in_mask, const float* __restrict__ w,
                                            const float* __restrict__ bias, const float* __restrict__ mask_src, float* __restrict__ out,
                                            int c_in, int c_out, int w_sin, int w_sout, int flip) {
    constexpr int R = c3_band_rows(W), WP = W + 2, PLANE = c3_pad((R + 2) * WP, 16);
    constexpr int STAGE = c3_stage(W, CIN_CHUNK), SUB = STAGE / CIN_CHUNK;      // sub-chunks of CIN_CHUNK channels per pass
    constexpr int GROUPS = CIN_CHUNK / 4, KSTEPS = 9 * GROUPS;
    constexpr int TILES = R * W / 16, TPW = TILES / 4;              // 16-pixel tiles per band / per wave
    static_assert(TILES % 4 == 0, "a band must give every wave the same number of tiles");
    __shared__ float tile[STAGE * PLANE];
    // a sub-chunk's weights: 16 rows (the index with the large stride in memory: the output channel forward, the input channel
    // for an input gradient) of 16 * 9 contiguous floats (CIN_CHUNK * 9 for the 4-channel first layer), rows padded to an odd stride
    constexpr int WROW = 16 * 9 + 1;
    __shared__ float wl[SUB * 16 * WROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int px = lane & 15, q = lane >> 4;
    constexpr int bands = W / R;
    const int n = blockIdx.x / bands, y0 = (blockIdx.x % bands) * R;
    const int co0 = blockIdx.y * 16;
    c3_f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = (c3_f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's pixel inside a tile: 16 consecutive pixels of a row (W >= 16) or two rows of 8
    const int lp_r = (W >= 16) ? 0 : px / W, lp_c = (W >= 16) ? px : px % W;
    const int n_pass = (c_in + STAGE - 1) / STAGE;
    // staging of a pass: the band's rows are full image rows, so a row is W / 4 aligned float4 loads and the two halo columns of
    // every patch row are always zero (written once here).  (One scalar load per patch element -- 42 per thread on the 64-pixel
    // layers, each with its own address and clamp -- took 464 registers and, with the mask beside it, spilled to scratch.)
    constexpr int V4_ROW = W / 4, V4_ITEMS = STAGE * (R + 2) * V4_ROW, V4_SLOTS = (V4_ITEMS + 255) / 256;
    constexpr int INNER = (CIN_CHUNK < 16 ? CIN_CHUNK : 16) * 9;
    constexpr int WSLOTS = (SUB * 16 * 16 * 9 + 255) / 256;
    for (int i = t; i < STAGE * (R + 2); i += 256) {
        const int c = i / (R + 2), rr = i - c * (R + 2);
        tile[c * PLANE + rr * WP] = 0.f;
        tile[c * PLANE + rr * WP + WP - 1] = 0.f;
    }
    for (int pass = 0; pass < n_pass; ++pass) {
        const int p_base = pass * STAGE;
        if (pass) __syncthreads();
        // ---- every global load of the pass is issued before the first LDS write (unconditional, clamped addresses: a guarded
        //      load is not hoisted and costs a memory latency of its own): the band's input rows (+1 halo row each side) of STAGE
        //      channels -- zeros outside the image / beyond c_in --, the ReLU mask beside them when the input is a gradient at a
        //      ReLU's output (in_mask: x counts only where that activation is > 0, the U-Net's layers), and the weights of the
        //      pass's sub-chunks
        float4 st[V4_SLOTS], mk[V4_SLOTS];
#pragma unroll
        for (int u = 0; u < V4_SLOTS; ++u) {
            const int i = t + 256 * u;
            const int c = i / ((R + 2) * V4_ROW), rem = i - c * ((R + 2) * V4_ROW);
            const int rr = rem / V4_ROW, j = rem - rr * V4_ROW;
            const int y = y0 - 1 + rr;
            const int yc = y < 0 ? 0 : (y >= W ? W - 1 : y);
            const int ch = p_base + c < c_in ? p_base + c : c_in - 1;
            const size_t idx = (((size_t)n * c_in + ch) * W + yc) * W + 4 * j;
            st[u] = *reinterpret_cast<const float4*>(x + idx);
            if (in_mask != nullptr) mk[u] = *reinterpret_cast<const float4*>(in_mask + idx);
        }
        float ws[WSLOTS];
#pragma unroll
        for (int u = 0; u < WSLOTS; ++u) {
            const int i = t + 256 * u;
            const int inner_n = flip ? 16 * 9 : INNER;
            const int sub = i / (16 * inner_n), i2 = i - sub * (16 * inner_n);
            const int r = i2 / inner_n, j = i2 - r * inner_n;
            const int c_base = p_base + sub * CIN_CHUNK;
            bool ok = sub < SUB;
            size_t a;
            if (flip) { ok = ok && c_base + r < c_in; a = (size_t)(ok ? c_base + r : 0) * w_sin + (size_t)co0 * 9 + j; }
            else { ok = ok && c_base + j / 9 < c_in; a = (size_t)(co0 + r) * w_sout + (size_t)(ok ? c_base : 0) * 9 + (ok ? j : 0); }
            const float v = w[a];
            ws[u] = ok ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < V4_SLOTS; ++u) {
            const int i = t + 256 * u;
            if (i < V4_ITEMS) {
                const int c = i / ((R + 2) * V4_ROW), rem = i - c * ((R + 2) * V4_ROW);
                const int rr = rem / V4_ROW, j = rem - rr * V4_ROW;
                const int y = y0 - 1 + rr;
                const bool ok = y >= 0 && y < W && p_base + c < c_in;
                float4 v = st[u];
                if (in_mask != nullptr) {
                    v.x = mk[u].x > 0.f ? v.x : 0.f; v.y = mk[u].y > 0.f ? v.y : 0.f;
                    v.z = mk[u].z > 0.f ? v.z : 0.f; v.w = mk[u].w > 0.f ? v.w : 0.f;
                }
                float* d = tile + c * PLANE + rr * WP + 1 + 4 * j;
                d[0] = ok ? v.x : 0.f; d[1] = ok ? v.y : 0.f; d[2] = ok ? v.z : 0.f; d[3] = ok ? v.w : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < WSLOTS; ++u) {
            const int i = t + 256 * u;
            const int inner_n = flip ? 16 * 9 : INNER;
            const int sub = i / (16 * inner_n), i2 = i - sub * (16 * inner_n);
            const int r = i2 / inner_n, j = i2 - r * inner_n;
            if (sub < SUB) wl[sub * 16 * WROW + r * WROW + j] = ws[u];
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            if (p_base + sub * CIN_CHUNK >= c_in) break;              // (uniform) the last pass of a 48- or 16-channel input
            // ---- B fragments: B[k = q][n = px] for every (tap, channel group) of this sub-chunk
            float bf[KSTEPS];
            const float* wsub = wl + sub * 16 * WROW;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int tap = s / GROUPS, g = s % GROUPS;
                const int cl = 4 * g + q;                            // input channel inside the sub-chunk
                bf[s] = flip ? wsub[cl * WROW + px * 9 + (8 - tap)] : wsub[px * WROW + cl * 9 + tap];
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int ti = wave * TPW + i;
                const int row0 = (W >= 16) ? (ti * 16) / W : ti * 2, col0 = (W >= 16) ? (ti * 16) % W : 0;
                // patch row 0 = image row y0 - 1, column 0 = x = -1
                const int base = (sub * CIN_CHUNK + q) * PLANE + (row0 + lp_r) * WP + col0 + lp_c;
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    const int tap = s / GROUPS, g = s % GROUPS;
                    const int dy = tap / 3, dx = tap % 3;
                    const float a = tile[base + 4 * g * PLANE + dy * WP + dx];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[s], acc[i], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: lane = (output channel co0 + px, pixels 4 q .. 4 q + 3 of the tile)
    const int co = co0 + px;
    const float b = (EPI == C3_EPI_BIAS_RELU) ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ti = wave * TPW + i;
        int row, col;
        if (W >= 16) { row = (ti * 16) / W; col = (ti * 16) % W + 4 * q; }
        else { row = ti * 2 + (4 * q) / W; col = (4 * q) % W; }
        const size_t o = (((size_t)n * c_out + co) * W + (y0 + row)) * W + col;
        c3_f32x4 u = acc[i];
        float4 p;
        if (EPI == C3_EPI_BIAS_RELU) {
            p.x = fmaxf(u[0] + b, 0.f); p.y = fmaxf(u[1] + b, 0.f); p.z = fmaxf(u[2] + b, 0.f); p.w = fmaxf(u[3] + b, 0.f);
        } else if (EPI == C3_EPI_MASK) {
            const float4 m = *reinterpret_cast<const float4*>(mask_src + o);
            p.x = m.x > 0.f ? u[0] : 0.f; p.y = m.y > 0.f ? u[1] : 0.f; p.z = m.z > 0.f ? u[2] : 0.f; p.w = m.w > 0.f ? u[3] : 0.f;
        } else {
            p.x = u[0]; p.y = u[1]; p.z = u[2]; p.w = u[3];
        }
        *reinterpret_cast<float4*>(out + o) = p;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient.  grid = (co_tiles * ci_tiles, splits); split j handles the (image, band) units [j * ups, (j + 1) * ups).
// part [splits, c_out, c_in, 9], part_b [splits, c_out] (written by the ci_tile == 0 workgroups).
template <int W>
__global__ __launch_bounds__(256) void k_c3_wgrad(const float* __restrict__ g, const float* __restrict__ g_mask, const float* __restrict__ x,
                                                  float* __restrict__ part, float* __restrict__ part_b, int N, int c_in, int c_out, int ups) {
    constexpr int R = c3_wgrad_rows(W), WP = W + 2, bands = W / R;
    constexpr int GP = c3_pad(R * W, 4), XP = c3_pad((R + 2) * WP, 4);
    constexpr int KS = R * W / 4;                                   // k-steps (4 pixels each) per unit
    constexpr int RED = 4 * 36 * 64;                                // cross-wave reduction scratch (floats)
    constexpr int LDS = (16 * GP + 16 * XP) > RED ? (16 * GP + 16 * XP) : RED;
    __shared__ __attribute__((aligned(16))) float lds[LDS];
    __shared__ float bsum[4][16];
    float* gt = lds;                                                // [16 co][R * W] (+ pad)
    float* xt = lds + 16 * GP;                                      // [16 ci][(R + 2) * WP] (+ pad)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int px = lane & 15, q = lane >> 4;
    const int ci_tiles = (c_in + 15) / 16;
    const int co0 = (blockIdx.x / ci_tiles) * 16, ci0 = (blockIdx.x % ci_tiles) * 16;
    const int units = N * bands;
    const int u_lo = blockIdx.y * ups, u_hi = (u_lo + ups < units) ? u_lo + ups : units;
    c3_f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (c3_f32x4){0.f, 0.f, 0.f, 0.f};
    float gsum = 0.f;
    // the halo columns of the input patch are zero for every unit (a band's rows are full image rows)
    for (int i = t; i < 16 * (R + 2); i += 256) {
        const int c = i / (R + 2), rr = i - c * (R + 2);
        xt[c * XP + rr * WP] = 0.f;
        xt[c * XP + rr * WP + WP - 1] = 0.f;
    }
    for (int u = u_lo; u < u_hi; ++u) {
        const int n = u / bands, y0 = (u % bands) * R;
        if (u != u_lo) __syncthreads();
        {
            // every global load of the unit before the first LDS write (unconditional, clamped addresses), rows as float4
            constexpr int G4 = 16 * R * W / 4, X4_ROW = W / 4, X4 = 16 * (R + 2) * X4_ROW;
            constexpr int GS = (G4 + 255) / 256, XS = (X4 + 255) / 256;
            float4 gv[GS], gm[GS], xv[XS];
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const int i = t + 256 * k;
                const int c = i / (R * W / 4), p4 = i - c * (R * W / 4);
                const size_t idx = (((size_t)n * c_out + co0 + (c < 16 ? c : 15)) * W + y0) * W + 4 * p4;   // a band's rows are contiguous
                gv[k] = *reinterpret_cast<const float4*>(g + idx);
                if (g_mask != nullptr) gm[k] = *reinterpret_cast<const float4*>(g_mask + idx);   // g = the gradient at a ReLU's output
            }
#pragma unroll
            for (int k = 0; k < XS; ++k) {
                const int i = t + 256 * k;
                const int c = i / ((R + 2) * X4_ROW), rem = i - c * ((R + 2) * X4_ROW);
                const int rr = rem / X4_ROW, j = rem - rr * X4_ROW;
                const int y = y0 - 1 + rr;
                const int yc = y < 0 ? 0 : (y >= W ? W - 1 : y);
                const int cc = c < 16 ? c : 15;
                const int ch = ci0 + cc < c_in ? ci0 + cc : c_in - 1;
                xv[k] = *reinterpret_cast<const float4*>(x + (((size_t)n * c_in + ch) * W + yc) * W + 4 * j);
            }
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const int i = t + 256 * k;
                if (i < G4) {
                    const int c = i / (R * W / 4), p4 = i - c * (R * W / 4);
                    float4 v = gv[k];
                    if (g_mask != nullptr) {
                        v.x = gm[k].x > 0.f ? v.x : 0.f; v.y = gm[k].y > 0.f ? v.y : 0.f;
                        v.z = gm[k].z > 0.f ? v.z : 0.f; v.w = gm[k].w > 0.f ? v.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(gt + c * GP + 4 * p4) = v;
                }
            }
#pragma unroll
            for (int k = 0; k < XS; ++k) {
                const int i = t + 256 * k;
                if (i < X4) {
                    const int c = i / ((R + 2) * X4_ROW), rem = i - c * ((R + 2) * X4_ROW);
                    const int rr = rem / X4_ROW, j = rem - rr * X4_ROW;
                    const int y = y0 - 1 + rr;
                    const bool ok = y >= 0 && y < W && ci0 + c < c_in;
                    float* d = xt + c * XP + rr * WP + 1 + 4 * j;
                    d[0] = ok ? xv[k].x : 0.f; d[1] = ok ? xv[k].y : 0.f; d[2] = ok ? xv[k].z : 0.f; d[3] = ok ? xv[k].w : 0.f;
                }
            }
        }
        __syncthreads();
        for (int s = wave; s < KS; s += 4) {
            const int p = 4 * s + q;                                 // this lane's pixel of the k-step (A: k = q; B: k = q)
            const int row = p / W, col = p - row * W;
            const float a = gt[px * GP + p];                         // A[m = co = px][k = q]
            gsum += a;
            const float* xb = xt + px * XP + row * WP + col;         // B[k = q][n = ci = px], tap (dy, dx) at + dy * WP + dx
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[(tap / 3) * WP + (tap % 3)], acc[tap], 0, 0, 0);
        }
    }
    __syncthreads();
    // ---- add the four waves' tiles in a fixed order; lane holds D[m = co = 4 q + u][n = ci = px] of every tap
    float* red = lds;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int u = 0; u < 4; ++u) red[(wave * 36 + tap * 4 + u) * 64 + lane] = acc[tap][u];
    gsum += __shfl_xor(gsum, 16);
    gsum += __shfl_xor(gsum, 32);
    if (q == 0) bsum[wave][px] = gsum;
    __syncthreads();
    for (int i = t; i < 36 * 64; i += 256) {
        const int slot = i >> 6, l = i & 63;
        const float v = ((red[(0 * 36 + slot) * 64 + l] + red[(1 * 36 + slot) * 64 + l]) + red[(2 * 36 + slot) * 64 + l]) + red[(3 * 36 + slot) * 64 + l];
        const int tap = slot >> 2, u = slot & 3;
        const int co = co0 + 4 * (l >> 4) + u, ci = ci0 + (l & 15);
        if (ci < c_in) part[(((size_t)blockIdx.y * c_out + co) * c_in + ci) * 9 + tap] = v;
    }
    if (part_b && ci0 == 0 && t < 16)
        part_b[(size_t)blockIdx.y * c_out + co0 + t] = ((bsum[0][t] + bsum[1][t]) + bsum[2][t]) + bsum[3][t];
}

// dw[i] = sum_s part[s][i], db[c] = sum_s part_b[s][c] in a FIXED order (deterministic): a workgroup owns 16 outputs, 16 lanes
// per output each add the splits j = lane, lane + 16, ... (four loads in flight), then a fixed butterfly over the 16 lanes.
// (One thread per output walking 512 splits was a chain of 512 dependent-latency loads: 41 us per launch.)
// workgroups a reduction of n = n_w + n_b outputs over `splits` partial sums takes: 16 outputs per workgroup (16 lanes per output
// walk the splits), or -- up to 16 splits -- 256 outputs, one thread each (the deep layers: 37 k-147 k outputs, 4-16 splits: 16 x
// fewer workgroups, the loads of a split coalesced over the outputs).  Both forms add in the same order, s = 0, 1, 2, ...
__host__ __device__ inline int c3_reduce_blocks(int64_t n, int splits) { return (int)(splits <= 16 ? (n + 255) / 256 : (n + 15) / 16); }

__device__ __forceinline__ void c3_reduce_block(const float* __restrict__ part, const float* __restrict__ part_b, float* __restrict__ dw,
                                                float* __restrict__ db, int n_w, int n_b, int splits, int block) {
    if (splits <= 16) {
        const int i = block * 256 + threadIdx.x;
        if (i < n_w + n_b) {
            const bool is_w = i < n_w;
            const float* src = is_w ? part + i : part_b + (i - n_w);
            const size_t stride = is_w ? (size_t)n_w : (size_t)n_b;
            float v = 0.f;
            for (int j = 0; j < splits; ++j) v += src[(size_t)j * stride];
            if (is_w) dw[i] = v; else db[i - n_w] = v;
        }
        return;
    }
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;           // output within the group / split lane
    const int i = block * 16 + o;
    const bool is_w = i < n_w, live = i < n_w + n_b;
    const float* src = is_w ? part + i : part_b + (i - n_w);
    const size_t stride = is_w ? (size_t)n_w : (size_t)n_b;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
        int j = sl;
        for (; j + 48 < splits; j += 64) {
            s0 += src[(size_t)j * stride]; s1 += src[(size_t)(j + 16) * stride];
            s2 += src[(size_t)(j + 32) * stride]; s3 += src[(size_t)(j + 48) * stride];
        }
        for (; j < splits; j += 16) s0 += src[(size_t)j * stride];
    }
    float s = (s0 + s1) + (s2 + s3);
    __shared__ float red[16][17];
    red[sl][o] = s;
    __syncthreads();
    if (threadIdx.x < 16 && block * 16 + threadIdx.x < n_w + n_b) {
        const int oo = threadIdx.x, ii = block * 16 + oo;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][oo];
        if (ii < n_w) dw[ii] = v; else db[ii - n_w] = v;
    }
}

__global__ __launch_bounds__(256) void k_c3_reduce(const float* __restrict__ part, const float* __restrict__ part_b, float* __restrict__ dw,
                                                   float* __restrict__ db, int n_w, int n_b, int splits) {
    c3_reduce_block(part, part_b, dw, db, n_w, n_b, splits, blockIdx.x);
}

// The reductions of SEVERAL layers in one launch (the weight gradients of a whole backward pass: 8 launches of ~4.6 us for
// ConvNet, 18 for the U-Net policy, each waiting for its turn behind the layer's other kernels).  Workgroup b serves the job
// with block_start <= b < block_start of the next job; same arithmetic and order as k_c3_reduce.
__global__ __launch_bounds__(256) void k_reduce_jobs(const bridges_reduce_job* __restrict__ jobs, int n_jobs) {
    int j = 0;
    while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].block_start) ++j;
    const bridges_reduce_job job = jobs[j];
    c3_reduce_block(job.part, job.part_b, job.dw, job.db, job.n_w, job.n_b, job.splits, (int)blockIdx.x - job.block_start);
}

// Bias gradient of a convolution whose weights stay with the library (the U-Net's transposed and 1x1 convolutions):
// part[s][c] = sum of g[n, c, :] over the images n = s, s + S, ...; k_c3_reduce adds the S partial sums in a fixed order.
// torch's own sum over (N, H, W) is a multi-workgroup reduction with a semaphore -- the kind that returned garbage inside a
// replayed HIP graph (vec_dqn._check_graph_losses) -- and not reproducible from run to run; this one is both.
__global__ __launch_bounds__(256) void k_bias_grad_part(const float* __restrict__ g, float* __restrict__ part, int N, int C, int hw,
                                                        int S) {
    __shared__ float red[256];
    const int c = blockIdx.x, s = blockIdx.y, t = threadIdx.x;
    float acc = 0.f;
    for (int n = s; n < N; n += S) {
        const float* p = g + ((size_t)n * C + c) * hw;
        for (int i = t; i < hw; i += 256) acc += p[i];
    }
    red[t] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) part[(size_t)s * C + c] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the U-Net's ConvTranspose2d(kernel 2, stride 2) layers (cv.py:176, 179: 64 -> 32 on 16 x 16, 32 -> 16 on 32 x 32)
// and of its 1x1 output convolution to one channel (cv.py:182), forward of the latter.  y[n, co, 2i + a, 2j + b] = bias[co] +
// sum_ci x[n, ci, i, j] w[ci, co, a, b] is a GEMM per input pixel with K = c_in and N = c_out * 4 -- 134 MFLOP per layer at the
// replay batch, which the library served with a Winograd / implicit-GEMM kernel per direction, three layout transposes and a
// zero fill (~130 us per layer and step).  Plain FMA kernels on LDS tiles of 64 input pixels:
//   k_up2_dx     dx[n, ci, i, j] = sum_{co, a, b} g[n, co, 2i + a, 2j + b] w[ci, co, a, b]; grid (pixels / 64, c_in / 16);
//   k_up2_wgrad  part[s][ci][co][a][b] = sum over the workgroup's pixels of x g, part_b[s][co] = sum of g; k_c3_reduce adds the
//                workgroups' partial sums in a fixed order (deterministic);
//   k_pw1_fwd / k_pw1_bwd  y[n, p] = b + sum_ci x[n, ci, p] w[ci];  dx = g w[ci], partial sums of dw[ci] = sum x g and db = sum g.
// (The forward of the transposed convolution is the inference kernel k_upconv2x2, conv_kernels.hip.)
__global__ __launch_bounds__(256) void k_up2_dx(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, int c_in,
                                                int c_out, int H, int W) {
    extern __shared__ float up_lds[];
    const int K = c_out * 4, HW = H * W;
    float* gs = up_lds;                                             // [K][64]: g of the tile's pixels, k = (co, a, b)
    float* wt = up_lds + K * 64;                                    // [K][16]: w[ci0 + c][k] transposed
    const int t = threadIdx.x, p = t & 63, cig = t >> 6;
    const int pix0 = blockIdx.x * 64, n = pix0 / HW, q0 = pix0 - n * HW, ci0 = blockIdx.y * 16;
    for (int i = t; i < c_out * 2 * 64; i += 256) {                 // (co, a, pixel): one float2 = (b = 0, 1)
        const int pp = i & 63, ca = i >> 6, co = ca >> 1, a = ca & 1;
        const int q = q0 + pp, ii = q / W, jj = q - ii * W;
        const float2 v = *reinterpret_cast<const float2*>(g + (((size_t)n * c_out + co) * 2 * H + 2 * ii + a) * 2 * W + 2 * jj);
        gs[(co * 4 + a * 2) * 64 + pp] = v.x;
        gs[(co * 4 + a * 2 + 1) * 64 + pp] = v.y;
    }
    for (int i = t; i < 16 * K; i += 256) {
        const int c = i / K, k = i - c * K;
        wt[k * 16 + c] = (ci0 + c < c_in) ? w[(size_t)(ci0 + c) * K + k] : 0.f;
    }
    __syncthreads();
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const float gv = gs[k * 64 + p];
        const float4 wv = *reinterpret_cast<const float4*>(wt + k * 16 + cig * 4);
        acc.x += gv * wv.x; acc.y += gv * wv.y; acc.z += gv * wv.z; acc.w += gv * wv.w;
    }
    const int ci = ci0 + cig * 4;
    float* o = dx + ((size_t)n * c_in + ci) * HW + q0 + p;
    if (ci + 0 < c_in) o[0] = acc.x;
    if (ci + 1 < c_in) o[(size_t)HW] = acc.y;
    if (ci + 2 < c_in) o[(size_t)2 * HW] = acc.z;
    if (ci + 3 < c_in) o[(size_t)3 * HW] = acc.w;
}

// TI = c_in / 16 input channels x TJ = c_out * 4 / 16 columns per thread; tiles [blockIdx.x * tps, + tps) of 64 pixels.
template <int TI, int TJ>
__global__ __launch_bounds__(256) void k_up2_wgrad(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ part,
                                                   float* __restrict__ part_b, int c_out, int H, int W, int tiles, int tps) {
    constexpr int CI = 16 * TI, K = 16 * TJ;
    __shared__ float xs[CI * 65];                                   // [ci][pixel], rows padded: lanes of a read differ in ci
    __shared__ float gs[K * 65];                                    // [k][pixel]
    const int t = threadIdx.x, ti = t & 15, tj = t >> 4, HW = H * W;
    float acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = 0.f;
    float bsum = 0.f;
    const int u_lo = blockIdx.x * tps, u_hi = (u_lo + tps < tiles) ? u_lo + tps : tiles;
    for (int u = u_lo; u < u_hi; ++u) {
        const int pix0 = u * 64, n = pix0 / HW, q0 = pix0 - n * HW;
        if (u != u_lo) __syncthreads();
        for (int i = t; i < CI * 64; i += 256) {
            const int pp = i & 63, c = i >> 6;
            xs[c * 65 + pp] = x[((size_t)n * CI + c) * HW + q0 + pp];
        }
        for (int i = t; i < c_out * 2 * 64; i += 256) {
            const int pp = i & 63, ca = i >> 6, co = ca >> 1, a = ca & 1;
            const int q = q0 + pp, ii = q / W, jj = q - ii * W;
            const float2 v = *reinterpret_cast<const float2*>(g + (((size_t)n * c_out + co) * 2 * H + 2 * ii + a) * 2 * W + 2 * jj);
            gs[(co * 4 + a * 2) * 65 + pp] = v.x;
            gs[(co * 4 + a * 2 + 1) * 65 + pp] = v.y;
        }
        __syncthreads();
        for (int pp = 0; pp < 64; ++pp) {
            float xv[TI], gv[TJ];
#pragma unroll
            for (int a = 0; a < TI; ++a) xv[a] = xs[(ti * TI + a) * 65 + pp];
#pragma unroll
            for (int b = 0; b < TJ; ++b) gv[b] = gs[(tj * TJ + b) * 65 + pp];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b) acc[a][b] += xv[a] * gv[b];
        }
        if (t < c_out) {                                            // bias gradient: this tile's sum of g over pixels and (a, b)
            float sb = 0.f;
            for (int pp = 0; pp < 64; ++pp)
                sb += (gs[(t * 4) * 65 + pp] + gs[(t * 4 + 1) * 65 + pp]) + (gs[(t * 4 + 2) * 65 + pp] + gs[(t * 4 + 3) * 65 + pp]);
            bsum += sb;
        }
    }
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) part[(size_t)blockIdx.x * CI * K + (size_t)(ti * TI + a) * K + tj * TJ + b] = acc[a][b];
    if (t < c_out) part_b[(size_t)blockIdx.x * c_out + t] = bsum;
}

// 4 pixels per thread; x [N, c_in, HW], w [c_in], y [N, HW].
__global__ __launch_bounds__(256) void k_pw1_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                 float* __restrict__ y, int c_in, int HW, int64_t quads) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const int64_t e = 4 * i, n = e / HW, q = e - n * HW;
    const float b = bias[0];
    float4 acc = {b, b, b, b};
    for (int c = 0; c < c_in; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)n * c_in + c) * HW + q);
        const float wc = w[c];
        acc.x += v.x * wc; acc.y += v.y * wc; acc.z += v.z * wc; acc.w += v.w * wc;
    }
    *reinterpret_cast<float4*>(y + e) = acc;
}

// dx [N, c_in, HW] = g [N, HW] * w[c]; part[blockIdx.x][c] = this workgroup's sum of x g, part_b[blockIdx.x] = its sum of g.
// A workgroup walks quads blockIdx.x * 256 + t + k * gridDim.x * 256 (c_in <= 32).
__global__ __launch_bounds__(256) void k_pw1_bwd(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ w,
                                                 float* __restrict__ dx, float* __restrict__ part, float* __restrict__ part_b, int c_in, int HW,
                                                 int64_t quads) {
    __shared__ float red[4][33];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float aw[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) aw[c] = 0.f;
    float ab = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < quads; i += (int64_t)gridDim.x * 256) {
        const int64_t e = 4 * i, n = e / HW, q = e - n * HW;
        const float4 gv = *reinterpret_cast<const float4*>(g + e);
        ab += (gv.x + gv.y) + (gv.z + gv.w);
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            if (c < c_in) {
                const size_t o = ((size_t)n * c_in + c) * HW + q;
                const float4 xv = *reinterpret_cast<const float4*>(x + o);
                const float wc = w[c];
                float4 d = {gv.x * wc, gv.y * wc, gv.z * wc, gv.w * wc};
                *reinterpret_cast<float4*>(dx + o) = d;
                aw[c] += (xv.x * gv.x + xv.y * gv.y) + (xv.z * gv.z + xv.w * gv.w);
            }
        }
    }
    // fixed-order reduction: butterfly over the wave's 64 lanes, then the four waves in order
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        float v = aw[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][c] = v;
    }
    for (int o = 32; o > 0; o >>= 1) ab += __shfl_xor(ab, o);
    if (lane == 0) red[wave][32] = ab;
    __syncthreads();
    if (t < c_in) part[(size_t)blockIdx.x * c_in + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    if (t == 32) part_b[blockIdx.x] = ((red[0][32] + red[1][32]) + red[2][32]) + red[3][32];
}

// ---------------------------------------------------------------------------------------------------------------
// y [n*C, H/2, W/2] = MaxPool2d(2)(a [n*C, H, W]); one thread per output pixel pair row (2 outputs: 4 input columns x 2 rows).
__global__ __launch_bounds__(256) void k_maxpool2(const float* __restrict__ a, float* __restrict__ y, int64_t items, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= items) return;
    const int w4 = W >> 2, h2 = H >> 1;
    const int64_t qd = i % w4, rest = i / w4;
    const int64_t r = rest % h2, nc = rest / h2;
    const float* p = a + (nc * H + 2 * r) * W + 4 * qd;
    const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + W);
    float2 o;
    o.x = fmaxf(fmaxf(a0.x, a0.y), fmaxf(a1.x, a1.y));
    o.y = fmaxf(fmaxf(a0.z, a0.w), fmaxf(a1.z, a1.w));
    *reinterpret_cast<float2*>(y + (nc * h2 + r) * (W >> 1) + 2 * qd) = o;
}

// g [n*C, H, W] = gradient at the pre-pool activation a = relu(.) given dy at the pooled output: the FIRST maximum of each
// window in scan order takes dy (torch's max_pool2d backward), times the ReLU's [a > 0].
__global__ __launch_bounds__(256) void k_maxpool2_relu_bwd(const float* __restrict__ a, const float* __restrict__ dy, float* __restrict__ g,
                                                          int64_t items, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= items) return;
    const int w4 = W >> 2, h2 = H >> 1;
    const int64_t qd = i % w4, rest = i / w4;
    const int64_t r = rest % h2, nc = rest / h2;
    const size_t off = (nc * H + 2 * r) * W + 4 * qd;
    const float4 a0 = *reinterpret_cast<const float4*>(a + off), a1 = *reinterpret_cast<const float4*>(a + off + W);
    const float2 d = *reinterpret_cast<const float2*>(dy + (nc * h2 + r) * (W >> 1) + 2 * qd);
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float m = fmaxf(fmaxf(a0.x, a0.y), fmaxf(a1.x, a1.y));
        if (m > 0.f) {
            if (a0.x == m) g0.x = d.x; else if (a0.y == m) g0.y = d.x; else if (a1.x == m) g1.x = d.x; else g1.y = d.x;
        }
    }
    {
        const float m = fmaxf(fmaxf(a0.z, a0.w), fmaxf(a1.z, a1.w));
        if (m > 0.f) {
            if (a0.z == m) g0.z = d.y; else if (a0.w == m) g0.w = d.y; else if (a1.z == m) g1.z = d.y; else g1.w = d.y;
        }
    }
    *reinterpret_cast<float4*>(g + off) = g0;
    *reinterpret_cast<float4*>(g + off + W) = g1;
}

}  // namespace bridges

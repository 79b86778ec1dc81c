// K10: conv3x3 (padding 1) + bias + ReLU [+ MaxPool2d(2)] for the 64-pixel-wide, 16-output-channel layers of the conv
// Q-networks (robotoddler/models/cv.py:5-17 ConvBlock(4,16) / (16,16); :138-254 UNet e11, e12, d41, d42) -- inference
// passes only (acting, targets).  These are the layers the library serves worst: 256 KiB of activations per image and
// too few channels for its implicit-GEMM tiles (16-40 TFLOP/s of the 157 the f32 matrix cores have, plus layout
// transposes and separate bias / ReLU / pool passes around the convolution).
//
// One workgroup = one image x a band of 8 output rows; the band's input rows (+1 halo row each side, all input channels
// of a 16-channel chunk) are staged in LDS as [cin][10][72] (4 zero-padded columns each side keep the 16-B alignment;
// the plane stride of 720 floats = 16 mod 64 banks makes the four channel groups of an operand read conflict-free).
// The product runs on v_mfma_f32_16x16x4_f32: M = 16 pixels along x, N = the 16 output channels, K = 4 input channels of
// one filter tap.  Lane l supplies A[pixel = l & 15][cin = 4 g + (l >> 4)] -- ONE ds_read_b32 whose address is the tile
// base plus a compile-time offset per (tap, channel group) -- and B[cin][cout = l & 15] from registers loaded once.
// A wave owns 2 rows x 4 column tiles (8 accumulator tiles); the result tile has cout on the lane and 4 consecutive
// pixels in the registers, so bias + ReLU and the 2x2 max-pool (row pair in the same lane) happen in registers and
// the stores are 16 B (8 B pooled) per lane.
#include "bridges_device.h"

namespace bridges {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CONV_W 64                       // image width the kernel is built for
#define CONV_BAND 8                     // output rows per workgroup
#define CONV_LDS_W 72                   // 4 + 64 + 4 columns
#define CONV_PLANE ((CONV_BAND + 2) * CONV_LDS_W)      // 720 floats per input channel

// CIN_CHUNK = input channels staged at once (4 or 16), N_CHUNKS chunks in total (C_in = CIN_CHUNK * N_CHUNKS).
// Epilogues: what leaves the kernel after relu(conv + bias)
#define CONV_EPI_PLAIN 0      // out [n,16,H,64]
#define CONV_EPI_POOL 1       // out [n,16,H/2,32] = MaxPool2d(2)
#define CONV_EPI_BOTH 2       // out = plain AND out2 = pooled (U-Net encoder: the skip tensor and the next level's input)
#define CONV_EPI_PROJ 3       // out [n,1,H,64] = sum_c proj_w[c] * relu(...)_c + proj_b (a 1x1 convolution to one channel behind it)

// x2 != nullptr: the input channels come from two tensors, chunk 0 from x (c_in channels), chunk 1 from x2 (c_in2) -- the
// torch.cat([upconv(u), skip], dim=1) in front of a U-Net decoder convolution without materialising it.
template <int CIN_CHUNK, int N_CHUNKS, int EPI>
__global__ __launch_bounds__(256) void k_conv3x3_o16(const float* __restrict__ x, const float* __restrict__ x2,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, float* __restrict__ out2,
                                                     const float* __restrict__ proj_w, const float* __restrict__ proj_b,
                                                     int H, int c_in, int c_in2) {
    constexpr bool POOL = (EPI == CONV_EPI_POOL);
    const int c_w = c_in + c_in2;                                // input channels of the weight tensor
    // c_in = channels the tensors really hold (<= CIN_CHUNK * N_CHUNKS; the missing ones count as zero planes)
    constexpr int GROUPS = CIN_CHUNK / 4;                        // channel groups of 4 per chunk
    constexpr int KSTEPS = 9 * GROUPS;                           // MFMAs per tile and chunk
    __shared__ float tile[CIN_CHUNK * CONV_PLANE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bands = H / CONV_BAND;
    const int n = blockIdx.x / bands, y0 = (blockIdx.x % bands) * CONV_BAND;
    const int px = lane & 15, q = lane >> 4;                     // pixel within the tile / channel within the group of 4
    f32x4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < N_CHUNKS; ++chunk) {
        if (chunk) __syncthreads();                              // the previous chunk's reads are done
        // ---- stage [CIN_CHUNK][10][72]: 16 threads per row (float4 each), zero halo rows / columns
        const bool second = (x2 != nullptr && chunk == 1);
        const int c_here = second ? c_in2 : c_in;                // channels of the tensor this chunk reads
        const int c_base = second ? 0 : chunk * CIN_CHUNK;       // first channel of the chunk inside that tensor
        const float* xin = (second ? x2 : x) + ((size_t)n * c_here + c_base) * H * CONV_W;
        // (all global loads are issued before the first LDS write: one memory latency per band, not one per pass)
        constexpr int ITEMS = CIN_CHUNK * (CONV_BAND + 2) * 18;
        constexpr int SLOTS = (ITEMS + 255) / 256;
        float4 stage[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int i = t + 256 * u;
            const int seg = i % 18, rr = (i / 18) % (CONV_BAND + 2), c = i / (18 * (CONV_BAND + 2));
            const int y = y0 - 1 + rr;
            stage[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < ITEMS && seg >= 1 && seg <= 16 && y >= 0 && y < H && c_base + c < c_here)
                stage[u] = *reinterpret_cast<const float4*>(xin + ((size_t)c * H + y) * CONV_W + 4 * (seg - 1));
        }
        // ---- B fragments of this chunk: B[k = q][cout = px] for every (tap, group)
        float bf[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int tap = s / GROUPS, g = s % GROUPS;
            const int cl = 4 * g + q;                            // channel inside the chunk
            const int cin = (second ? c_in : c_base) + cl;       // its index in the weight tensor
            const bool live = c_base + cl < c_here;
            const int cin_ld = live ? cin : 0;                   // unconditional load: a guarded one is not hoisted over the
            const float wv = w[((size_t)px * c_w + cin_ld) * 9 + tap];    // staging and costs a second memory latency (0.65 vs 0.43 ms)
            bf[s] = live ? wv : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int i = t + 256 * u;
            if (i < ITEMS) {
                const int seg = i % 18, rr = (i / 18) % (CONV_BAND + 2), c = i / (18 * (CONV_BAND + 2));
                *reinterpret_cast<float4*>(&tile[c * CONV_PLANE + rr * CONV_LDS_W + 4 * seg]) = stage[u];
            }
        }
        __syncthreads();
        // ---- 2 rows x 4 column tiles per wave
        const int base = q * CONV_PLANE + (2 * wave) * CONV_LDS_W + 3 + px;      // row 2*wave-1+1, column x0 + px - 1 (+4 pad)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    const int tap = s / GROUPS, g = s % GROUPS;
                    const int dy = tap / 3, dx = tap % 3;
                    const float a = tile[base + 4 * g * CONV_PLANE + (r + dy) * CONV_LDS_W + 16 * c + dx];
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[s], acc[r][c], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: lane = (cout = px, pixels 4 q .. 4 q + 3 of the tile)
    const float b = bias[px];
    const int y = y0 + 2 * wave;
    if (EPI == CONV_EPI_POOL || EPI == CONV_EPI_BOTH) {
        float* o = (EPI == CONV_EPI_BOTH ? out2 : out) + (((size_t)n * 16 + px) * (H / 2) + (y >> 1)) * (CONV_W / 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 u = acc[0][c], v = acc[1][c];
            float2 p;
            p.x = fmaxf(fmaxf(fmaxf(u[0], u[1]), fmaxf(v[0], v[1])) + b, 0.f);
            p.y = fmaxf(fmaxf(fmaxf(u[2], u[3]), fmaxf(v[2], v[3])) + b, 0.f);
            *reinterpret_cast<float2*>(o + 8 * c + 2 * q) = p;
        }
    }
    if (EPI == CONV_EPI_PLAIN || EPI == CONV_EPI_BOTH) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float* o = out + (((size_t)n * 16 + px) * H + (y + r)) * CONV_W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 u = acc[r][c];
                float4 p;
                p.x = fmaxf(u[0] + b, 0.f); p.y = fmaxf(u[1] + b, 0.f); p.z = fmaxf(u[2] + b, 0.f); p.w = fmaxf(u[3] + b, 0.f);
                *reinterpret_cast<float4*>(o + 16 * c + 4 * q) = p;
            }
        }
    }
    if (EPI == CONV_EPI_PROJ) {
        // 1x1 convolution to one channel: the 16 channels of a pixel sit in the 16 lanes px = 0..15 of a lane group
        const float pw = proj_w[px], pb = proj_b[0];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float* o = out + ((size_t)n * H + (y + r)) * CONV_W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 u = acc[r][c];
                float4 p;
                p.x = pw * fmaxf(u[0] + b, 0.f); p.y = pw * fmaxf(u[1] + b, 0.f);
                p.z = pw * fmaxf(u[2] + b, 0.f); p.w = pw * fmaxf(u[3] + b, 0.f);
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {               // sum over the 16 lanes of the group (fixed butterfly order)
                    p.x += __shfl_xor(p.x, m); p.y += __shfl_xor(p.y, m); p.z += __shfl_xor(p.z, m); p.w += __shfl_xor(p.w, m);
                }
                if (px == 0) {
                    p.x += pb; p.y += pb; p.z += pb; p.w += pb;
                    *reinterpret_cast<float4*>(o + 16 * c + 4 * q) = p;
                }
            }
        }
    }
}

}  // namespace bridges

namespace bridges {

// ConvTranspose2d(kernel 2, stride 2) + bias (cv.py:176, 179 UNet upconv3 / upconv4), inference passes:
//   out[n][co][2y + dy][2x + dx] = bias[co] + sum_ci in[n][ci][y][x] * w[ci][co][dy][dx].
// Per image a [H W x C_in] . [C_in x 4 C_out] product whose 384 KiB (upconv4) of traffic bound it; the library runs it as
// a GEMM plus a col2im pass.  Here: v_mfma_f32_16x16x4_f32 with M = 16 input pixels along x, K = 4 input channels,
// N = 16 output channels of one (dy, dx); the A operand is read straight from global memory (lane l: channel 4 g + (l >> 4),
// pixel x0 + (l & 15): 64-B runs) and serves all 4 C_out / 16 column tiles; a lane ends up with 4 consecutive input pixels
// for dx = 0 and dx = 1, i.e. 8 consecutive output pixels: two 16-B stores.
// One wave = one tile of 16 input pixels; CO_TILES = C_out / 16.
template <int C_IN, int CO_TILES>
__global__ __launch_bounds__(256) void k_upconv2x2(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                   long n_tiles) {
    constexpr int C_OUT = 16 * CO_TILES;
    constexpr int KS = C_IN / 4;
    const int lane = threadIdx.x & 63, px = lane & 15, q = lane >> 4;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int xt = W / 16;                                       // tiles per input row
    const long n = tile / ((long)H * xt);
    const int rem = (int)(tile % ((long)H * xt)), y = rem / xt, x0 = (rem % xt) * 16;
    float a[KS];
#pragma unroll
    for (int g = 0; g < KS; ++g) a[g] = x[(((size_t)n * C_IN + 4 * g + q) * H + y) * W + x0 + px];
    const int Wo = 2 * W;
#pragma unroll
    for (int ct = 0; ct < CO_TILES; ++ct) {
        const int co = 16 * ct + px;
        const float bv = bias[co];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < KS; ++g) {
                const float* wp = w + (((size_t)(4 * g + q) * C_OUT + co) * 2 + dy) * 2;        // B[k = q][j = px], dx = 0 / 1
                const float2 wv = *reinterpret_cast<const float2*>(wp);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], wv.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], wv.y, acc1, 0, 0, 0);
            }
            // lane: channel co, input pixels x0 + 4 q + u (u = 0..3) -> output pixels 2 (x0 + 4 q) .. + 7 of row 2 y + dy
            float* o = out + (((size_t)n * C_OUT + co) * (2 * H) + 2 * y + dy) * Wo + 2 * (x0 + 4 * q);
            float4 lo, hi;
            lo.x = acc0[0] + bv; lo.y = acc1[0] + bv; lo.z = acc0[1] + bv; lo.w = acc1[1] + bv;
            hi.x = acc0[2] + bv; hi.y = acc1[2] + bv; hi.z = acc0[3] + bv; hi.w = acc1[3] + bv;
            *reinterpret_cast<float4*>(o) = lo;
            *reinterpret_cast<float4*>(o + 4) = hi;
        }
    }
}

}  // namespace bridges

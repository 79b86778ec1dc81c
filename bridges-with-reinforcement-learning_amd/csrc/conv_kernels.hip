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
template <int CIN_CHUNK, int N_CHUNKS, bool POOL>
__global__ __launch_bounds__(256) void k_conv3x3_o16(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int H, int c_in) {
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
        const float* xin = x + ((size_t)n * c_in + (size_t)chunk * CIN_CHUNK) * H * CONV_W;
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
            if (i < ITEMS && seg >= 1 && seg <= 16 && y >= 0 && y < H && chunk * CIN_CHUNK + c < c_in)
                stage[u] = *reinterpret_cast<const float4*>(xin + ((size_t)c * H + y) * CONV_W + 4 * (seg - 1));
        }
        // ---- B fragments of this chunk: B[k = q][cout = px] for every (tap, group)
        float bf[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int tap = s / GROUPS, g = s % GROUPS;
            const int cin = chunk * CIN_CHUNK + 4 * g + q;
            bf[s] = cin < c_in ? w[((size_t)px * c_in + cin) * 9 + tap] : 0.f;
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
    if (POOL) {
        float* o = out + (((size_t)n * 16 + px) * (H / 2) + (y >> 1)) * (CONV_W / 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 u = acc[0][c], v = acc[1][c];
            float2 p;
            p.x = fmaxf(fmaxf(fmaxf(u[0], u[1]), fmaxf(v[0], v[1])) + b, 0.f);
            p.y = fmaxf(fmaxf(fmaxf(u[2], u[3]), fmaxf(v[2], v[3])) + b, 0.f);
            *reinterpret_cast<float2*>(o + 8 * c + 2 * q) = p;
        }
    } else {
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
}

}  // namespace bridges

// K9: one optimiser step of SuccessorMLP at replay-batch size (robotoddler/models/cv.py:76-105 forward,
// robotoddler/training/successor_dqn.py:157-235 train_policy_net) as a handful of launches.
//
// At batch 32 every layer is a skinny GEMM ([32 x K] . [K x N]): the weights are streamed exactly once per pass and the
// work per weight is 2 * 32 flops -- 16 flop/B, more than the vector ALUs deliver at HBM speed, so the products run on
// the f32 matrix cores (v_mfma_f32_32x32x2_f32: the 32 batch rows ARE the M side of the tile).  The library GEMMs torch
// picks for these shapes use 16 workgroups for the 16.8 MB first layer (48 us against 3.4 us of weight traffic) and
// one optimiser step is ~65 launches of ~5 us each; here it is ~25.
//
// Operand maps (cdna_hip_programming.md, "Fragment layout"): lane l supplies A[row = l & 31][k = l >> 5] and
// B[k = l >> 5][col = l & 31]; the accumulator register r of lane l is C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)]
// [col = l & 31].  The k order inside a tile is free as long as A and B agree, which lets every lane fetch 16 B.
#include "bridges_device.h"

namespace bridges {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));       // rows of K = 4 px + 6 floats are only 8-B aligned

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Sum of p[s * stride], s < n, 32 loads in flight (a dependent chain of n loads costs n memory latencies; with eight in
// flight the 57 first-layer splits were eight round trips of a 4.9 us finishing launch, now two); the order of the additions
// is fixed, so the result is reproducible.
#define SPLIT_SUM_WIDTH 32
__device__ __forceinline__ float split_sum(const float* __restrict__ p, size_t stride, int n) {
    float acc[SPLIT_SUM_WIDTH];
#pragma unroll
    for (int u = 0; u < SPLIT_SUM_WIDTH; ++u) acc[u] = 0.f;
    for (int s = 0; s < n; s += SPLIT_SUM_WIDTH) {
        float v[SPLIT_SUM_WIDTH];
#pragma unroll
        for (int u = 0; u < SPLIT_SUM_WIDTH; ++u) v[u] = (s + u < n) ? p[(size_t)(s + u) * stride] : 0.f;
#pragma unroll
        for (int u = 0; u < SPLIT_SUM_WIDTH; ++u) acc[u] += v[u];
    }
#pragma unroll
    for (int w = SPLIT_SUM_WIDTH / 2; w >= 1; w >>= 1) {                 // pairwise tree, fixed
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] = acc[2 * u] + acc[2 * u + 1];
    }
    return acc[0];
}

// Waves 1..3 of a workgroup hand their 32x32 accumulators to wave 0 (fixed summation order: deterministic).
__device__ __forceinline__ void reduce_to_wave0(f32x16& acc, float (*red)[16][64], int wave, int lane) {
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward: out[m][n] = sum_k x[m][k] * W[n][k] over the K range of this workgroup's split.
// grid = (ceil(N / 32), splits, rows / 32), 4 waves share the K range.  splits == 1: y = act(out + bias) directly,
// else part[split][m][n] = out and k_lin_fwd_finish adds the splits up.
__global__ __launch_bounds__(256) void k_lin_fwd(int K, int N, int kchunk, const float* __restrict__ x,
                                                 const float* __restrict__ W, const float* __restrict__ bias, int relu,
                                                 float* __restrict__ y, float* __restrict__ part,
                                                 const int64_t* __restrict__ x_block) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 32, split = blockIdx.y, m0 = blockIdx.z * 32;
    const int rows = gridDim.z * 32;
    if (x_block) x += (size_t)(*x_block) * rows * K;              // block *x_block of a [n_blocks * rows, K] array of inputs
    const int kbeg = split * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int kq = ((kend - kbeg + 31) / 32) * 8;                 // k values per wave, a multiple of 8
    const int wbeg = kbeg + wave * kq;
    const int wend = (wbeg + kq < kend) ? wbeg + kq : kend;
    const int row = lane & 31, half = lane >> 5;
    const int nrow = (n0 + row < N) ? n0 + row : N - 1;           // columns >= N are computed on a valid row, never stored
    const float* xp = x + (size_t)(m0 + row) * K + 4 * half;
    const float* wp = W + (size_t)nrow * K + 4 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int k0 = wbeg;
    for (; k0 + 64 <= wend; k0 += 64) {                           // 16 loads in flight, then 32 MFMAs
        f4u a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const f4u*>(xp + k0 + 8 * u);
            b[u] = *reinterpret_cast<const f4u*>(wp + k0 + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = mfma32(a[u].x, b[u].x, acc);
            acc = mfma32(a[u].y, b[u].y, acc);
            acc = mfma32(a[u].z, b[u].z, acc);
            acc = mfma32(a[u].w, b[u].w, acc);
        }
    }
    for (; k0 + 8 <= wend; k0 += 8) {
        const f4u a = *reinterpret_cast<const f4u*>(xp + k0);
        const f4u b = *reinterpret_cast<const f4u*>(wp + k0);
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
    }
    if (k0 < wend) {                                              // fewer than 8 k values left
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k0 + 4 * half + j;
            const bool ok = kk < wend;
            const float a = ok ? xp[k0 + j] : 0.f;
            const float b = ok ? wp[k0 + j] : 0.f;
            acc = mfma32(a, b, acc);
        }
    }
    reduce_to_wave0(acc, red, wave, lane);
    if (wave == 0) {
        const int n = n0 + row;
        if (n < N) {
            const float bv = part ? 0.f : bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mfma_row(r, lane);
                if (part) {
                    part[((size_t)split * rows + m) * N + n] = acc[r];
                } else {
                    float v = acc[r] + bv;
                    if (relu) v = v > 0.f ? v : 0.f;
                    y[(size_t)m * N + n] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_lin_fwd_finish(int rows, int N, int splits, const float* __restrict__ part,
                                                        const float* __restrict__ bias, int relu, float* __restrict__ y) {
    const size_t total = (size_t)rows * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = split_sum(part + i, total, splits) + bias[i % N];
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

// Adam folded into the weight-gradient jobs of a layer WITHOUT an input gradient (the first layer: nothing in the step
// reads its weights after the forward pass, so they may change in place): the 32x32 gradient tile in the accumulators
// goes straight into m, v and W -- the gradient never exists in memory (for SuccessorMLP's first layer that is 16.8 MB
// not written and not read again, and 4.2 M of the 6.4 M parameters out of the separate Adam launch).  Same arithmetic as
// k_adam_flat (dqn_kernels.hip).
struct AdamFold {
    float* W; float* bias; float* mW; float* vW; float* mb; float* vb;
    const float* step;                  // device float: number of THIS update (already incremented)
    double lr, beta1, beta2, eps;
    // the other layers' parameters (one contiguous range of the flat buffers): updated by extra workgroups of the same
    // launch -- when the first layer's backward runs, every other gradient is complete and nothing reads those weights
    // any more, so the whole optimiser update rides in the last launch of the step
    float* rest_p; const float* rest_g; float* rest_m; float* rest_v; long long rest_n;
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt, float beta2,
                                         float w1, float w2, float eps) {
    m = m + (g - m) * w1;
    v = beta2 * v + w2 * g * g;
    p = p - step_size * m / (sqrtf(v) / bc2_sqrt + eps);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of one Linear layer, both products in one launch.  dz [rows][N] = gradient at the layer's pre-activation.
//   jobs [0, n_dw_jobs): dW[n][k] = sum_b dz[b][n] * a[b][k] (32x32 tiles, the batch is the reduction); the jobs of
//     the first k group also write db[n] = sum_b dz[b][n].
//   jobs [n_dw_jobs, ...): dxpart[split][b][k] = sum_{n in split} dz[b][n] * W[n][k]  (k_lin_dx_finish adds the splits
//     and applies the ReLU mask of the layer below; with one split act_mask is given and the masked sum is final).
// Book-keeping of the step that needs every row's loss: losses[*counter] = sum_b loss_rows[b] (row order: the same float as
// k_loss_log), ++*counter, ++*adam_step.  It rides in the head layer's backward launch -- the first launch behind the loss
// kernel, which neither reads the batch counter nor the optimiser's step count -- as one thread's work beside the launch's
// jobs: the loss rows come from a finished kernel (plain loads), nothing waits for it.  (Until late in round 3 the loss
// kernel's last-arriving workgroup did this behind an agent-scope ticket: four device-scope round trips at the end of a
// 10 us kernel on the step's critical chain.)
struct LossLog {
    const float* loss_rows; int batch; float* losses; int n_losses; int64_t* counter; float* adam_step;
};
__device__ __forceinline__ void loss_log_apply(const LossLog& g) {
    float l = 0.f;
    for (int b = 0; b < g.batch; ++b) l += g.loss_rows[b];
    const int64_t c = *g.counter;
    if (g.losses && c >= 0 && c < g.n_losses) g.losses[c] = l;
    *g.counter = c + 1;
    if (g.adam_step) *g.adam_step = *g.adam_step + 1.f;
}

// (the ADAM form is bound by memory: two waves per SIMD -- <= 256 registers -- let one workgroup's loads run under the other's
// update arithmetic and stores; at the 276 registers the compiler takes unasked, every CU held a single workgroup)
template <bool ADAM>
__global__ __launch_bounds__(ADAM ? 512 : 256) __attribute__((amdgpu_waves_per_eu(ADAM ? 2 : 1, ADAM ? 2 : 8))) void k_lin_bwd(int rows, int K, int N, const float* __restrict__ dz,
                                                 const float* __restrict__ a, const float* __restrict__ W,
                                                 float* __restrict__ dW, float* __restrict__ db,
                                                 float* __restrict__ dxpart, const float* __restrict__ act_mask,
                                                 int n_dw_jobs, int ktiles_per_job, int nsplit, int nchunk, AdamFold ad,
                                                 const int64_t* __restrict__ a_block, int a_block_bias, LossLog log) {
    __shared__ float red[3][16][64];
    if constexpr (!ADAM) {
        if (log.counter && blockIdx.x == 0 && threadIdx.x == 0) loss_log_apply(log);      // (a_block is null in such a launch)
    }
    // block *a_block + a_block_bias of a [n_blocks * rows, K] array of layer inputs (bias -1: the step's loss kernel has
    // already advanced the batch counter the forward pass read)
    if (a_block) a += (size_t)(*a_block + a_block_bias) * rows * K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, half = lane >> 5;
    const int n_ntiles = (N + 31) / 32, n_ktiles = (K + 31) / 32;
    int job = blockIdx.x;
    if (job < n_dw_jobs) {
        const int nt = job % n_ntiles, kg = job / n_ntiles;
        const int n0 = nt * 32;
        const int ncol = (n0 + row < N) ? n0 + row : N - 1;
        const int kt_end = ((kg + 1) * ktiles_per_job < n_ktiles) ? (kg + 1) * ktiles_per_job : n_ktiles;
        const int nw = blockDim.x >> 6;                               // the waves of a workgroup take adjacent k tiles
        int kt = kg * ktiles_per_job + wave;
        const bool want_db = (kg == 0 && wave == 0);
        if (rows == 32) {
            // the common case (one batch tile): A fragments once per wave, B fragments of the next tile in flight while
            // this tile's MFMAs and stores run
            float av[16], bv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) av[t] = dz[(size_t)(2 * t + half) * N + ncol];
            float step_size = 0.f, bc2_sqrt = 1.f, w1 = 0.f, w2 = 0.f;
            if constexpr (ADAM) {
                const double t = (double)*ad.step;
                step_size = (float)(ad.lr / (1.0 - pow(ad.beta1, t)));
                bc2_sqrt = (float)sqrt(1.0 - pow(ad.beta2, t));
                w1 = (float)(1.0 - ad.beta1); w2 = (float)(1.0 - ad.beta2);
            }
            if (want_db) {
                float sdb = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) sdb += av[t];            // rows 2t + half; the other half sits 32 lanes away
                sdb += __shfl_xor(sdb, 32);
                if (half == 0 && n0 + row < N) {
                    if constexpr (ADAM) {
                        const int o = n0 + row;
                        float pb = ad.bias[o], mb = ad.mb[o], vb = ad.vb[o];
                        adam_one(pb, sdb, mb, vb, step_size, bc2_sqrt, (float)ad.beta2, w1, w2, (float)ad.eps);
                        ad.bias[o] = pb; ad.mb[o] = mb; ad.vb[o] = vb;
                    } else {
                        db[n0 + row] = sdb;
                    }
                }
            }
            if (kt < kt_end) {
                const int kcol = (kt * 32 + row < K) ? kt * 32 + row : K - 1;
#pragma unroll
                for (int t = 0; t < 16; ++t) bv[t] = a[(size_t)(2 * t + half) * K + kcol];
            }
            for (; kt < kt_end; kt += nw) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) acc = mfma32(av[t], bv[t], acc);
                const int k = kt * 32 + row;
                if (kt + nw < kt_end) {
                    const int kcol = ((kt + nw) * 32 + row < K) ? (kt + nw) * 32 + row : K - 1;
#pragma unroll
                    for (int t = 0; t < 16; ++t) bv[t] = a[(size_t)(2 * t + half) * K + kcol];
                }
                if (k < K) {
                    if constexpr (ADAM) {
                        float pw[16], pm[16], pv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {                // all loads of the tile first
                            const int n = n0 + mfma_row(r, lane);
                            const size_t o = (size_t)(n < N ? n : N - 1) * K + k;
                            pw[r] = ad.W[o]; pm[r] = ad.mW[o]; pv[r] = ad.vW[o];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int n = n0 + mfma_row(r, lane);
                            if (n < N) {
                                const size_t o = (size_t)n * K + k;
                                adam_one(pw[r], acc[r], pm[r], pv[r], step_size, bc2_sqrt, (float)ad.beta2, w1, w2, (float)ad.eps);
                                ad.W[o] = pw[r]; ad.mW[o] = pm[r]; ad.vW[o] = pv[r];
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int n = n0 + mfma_row(r, lane);
                            if (n < N) dW[(size_t)n * K + k] = acc[r];
                        }
                    }
                }
            }
            return;
        }
        if (want_db && half == 0 && n0 + row < N) {
            float sdb = 0.f;
            for (int b = 0; b < rows; ++b) sdb += dz[(size_t)b * N + n0 + row];
            db[n0 + row] = sdb;
        }
        for (; kt < kt_end; kt += nw) {
            const int k0 = kt * 32;
            const int kcol = (k0 + row < K) ? k0 + row : K - 1;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int mb = 0; mb < rows; mb += 32) {
                float av[16], bv[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const size_t b = (size_t)(mb + 2 * t + half);
                    av[t] = dz[b * N + ncol];                     // A[i = n][kk = b]
                    bv[t] = a[b * K + kcol];                      // B[kk = b][j = k]
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) acc = mfma32(av[t], bv[t], acc);
            }
            const int k = k0 + row;
            if (k < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + mfma_row(r, lane);
                    if (n < N) dW[(size_t)n * K + k] = acc[r];
                }
            }
        }
        return;
    }
    job -= n_dw_jobs;
    if constexpr (ADAM) {                                          // flat Adam over the other layers' range
        const double t = (double)*ad.step;
        const float step_size = (float)(ad.lr / (1.0 - pow(ad.beta1, t)));
        const float bc2_sqrt = (float)sqrt(1.0 - pow(ad.beta2, t));
        const float w1 = (float)(1.0 - ad.beta1), w2 = (float)(1.0 - ad.beta2), b2 = (float)ad.beta2, eps = (float)ad.eps;
        const long long n4 = ad.rest_n >> 2;                       // the range is padded to whole float4s
        const long long stride = (long long)(gridDim.x - n_dw_jobs) * blockDim.x;
        float4* p4 = reinterpret_cast<float4*>(ad.rest_p);
        const float4* g4 = reinterpret_cast<const float4*>(ad.rest_g);
        float4* m4 = reinterpret_cast<float4*>(ad.rest_m);
        float4* v4 = reinterpret_cast<float4*>(ad.rest_v);
        for (long long i = (long long)job * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
            adam_one(pp.x, gg.x, mm.x, vv.x, step_size, bc2_sqrt, b2, w1, w2, eps);
            adam_one(pp.y, gg.y, mm.y, vv.y, step_size, bc2_sqrt, b2, w1, w2, eps);
            adam_one(pp.z, gg.z, mm.z, vv.z, step_size, bc2_sqrt, b2, w1, w2, eps);
            adam_one(pp.w, gg.w, mm.w, vv.w, step_size, bc2_sqrt, b2, w1, w2, eps);
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
        return;
    }
    const int kt = job % n_ktiles, split = (job / n_ktiles) % nsplit, mt = job / (n_ktiles * nsplit);
    const int k0 = kt * 32, m0 = mt * 32;
    const int kcol = (k0 + row < K) ? k0 + row : K - 1;
    const int nbeg = split * nchunk;
    const int nend = (nbeg + nchunk < N) ? nbeg + nchunk : N;
    const int nq = ((nend - nbeg + 31) / 32) * 8;
    const int wbeg = nbeg + wave * nq;
    const int wend = (wbeg + nq < nend) ? wbeg + nq : nend;
    const float* dzp = dz + (size_t)(m0 + row) * N + 4 * half;   // A[i = b][kk = n]
    const float* wp = W + (size_t)(4 * half) * K + kcol;          // B[kk = n][j = k]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int n = wbeg;
    for (; n + 32 <= wend; n += 32) {
        f4u av[4];
        float bv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = *reinterpret_cast<const f4u*>(dzp + n + 8 * u);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][j] = wp[(size_t)(n + 8 * u + j) * K];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = mfma32(av[u].x, bv[u][0], acc);
            acc = mfma32(av[u].y, bv[u][1], acc);
            acc = mfma32(av[u].z, bv[u][2], acc);
            acc = mfma32(av[u].w, bv[u][3], acc);
        }
    }
    for (; n < wend; n += 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n + 4 * half + j;
            const bool ok = nn < wend;
            const float av = ok ? dzp[n + j] : 0.f;
            const float bv = ok ? wp[(size_t)(n + j) * K] : 0.f;
            acc = mfma32(av, bv, acc);
        }
    }
    reduce_to_wave0(acc, red, wave, lane);
    if (wave == 0 && k0 + row < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mfma_row(r, lane);
            const size_t o = ((size_t)split * rows + m) * K + k0 + row;
            float v = acc[r];
            if (act_mask && !(act_mask[o] > 0.f)) v = 0.f;       // single split: dxpart IS dz of the layer below
            dxpart[o] = v;
        }
    }
}

// dz_below[b][k] = (sum over splits of dxpart) masked by the ReLU of the layer below (act = its output, > 0 passes).
__global__ __launch_bounds__(256) void k_lin_dx_finish(int rows, int K, int nsplit, const float* __restrict__ dxpart,
                                                       const float* __restrict__ act, float* __restrict__ dz_below) {
    const size_t total = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = split_sum(dxpart + i, total, nsplit);
        if (act && !(act[i] > 0.f)) v = 0.f;
        dz_below[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Input rows of one replay batch: x[b] = [block | action | reward | obstacle | binary] (cv.py:100-103) for the batch
// `*counter` of the static per-call arrays (block_all / action_all [n][px], binary_all [n][nf]); rows >= batch are 0.
__global__ __launch_bounds__(256) void k_mlp_input(int batch, int rows, int px, int nf, const int64_t* __restrict__ counter,
                                                   const float* __restrict__ block_all, const float* __restrict__ action_all,
                                                   const float* __restrict__ binary_all, const float* __restrict__ reward,
                                                   const float* __restrict__ obstacle, float* __restrict__ x) {
    const int K = 4 * px + nf;
    const size_t total = (size_t)rows * K;
    // counter == nullptr: blockIdx.y = the batch, x the [n_batches * rows, K] array of all batches of a train_policy_net call
    const size_t cb = counter ? (size_t)(*counter) : (size_t)blockIdx.y;
    if (!counter) x += cb * total;
    const size_t base = cb * batch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / K), c = (int)(i % K);
        float v = 0.f;
        if (b < batch) {
            const size_t src = base + b;
            if (c < px) v = block_all[src * px + c];
            else if (c < 2 * px) v = action_all[src * px + (c - px)];
            else if (c < 3 * px) v = reward[c - 2 * px];
            else if (c < 4 * px) v = obstacle[c - 3 * px];
            else v = binary_all[src * nf + (c - 4 * px)];
        }
        x[i] = v;
    }
}

// What k_loss_log does, inside the loss kernel: the row workgroup that arrives LAST sums the per-row losses in row order
// (the same order, hence the same float), logs them, advances the batch counter and the optimiser's step count, and
// re-arms the ticket.  Inter-workgroup hand-off in the write-through form of cdna_hip_programming.md, Guideline 16 /
// "In-launch split-K reduction": the one handed-off value of a workgroup, loss_rows[b], is stored by thread 0 with an
// agent-scope (sc1, write-through) store -- no release fence, which would write the whole L2 back, the 1 MB of dy this
// launch has just stored included --, thread 0 drains its stores (s_waitcnt vmcnt(0)) and draws its ticket with an
// agent-scope atomic; the last arriver reads every row with agent-scope (sc1) loads.  The batch counter is read by every
// workgroup BEFORE it draws its ticket, so the last arriver may advance it.
__device__ __forceinline__ void loss_log_ticket(int batch, float* loss_rows, float* losses, int n_losses, int64_t* counter,
                                                int32_t* ticket, float* adam_step, int t) {
    if (t != 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int drawn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (drawn != (int)gridDim.x - 1) return;
    float l = 0.f;
    for (int b = 0; b < batch; ++b) l += __hip_atomic_load(&loss_rows[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t c = *counter;
    if (losses && c >= 0 && c < n_losses) losses[c] = l;
    *counter = c + 1;
    if (adam_step) *adam_step = *adam_step + 1.f;
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Head and loss (cv.py:104-108, successor_dqn.py:215-232): y [rows][2 px + 2 nf] = (psi0 | psi1 | binary part).
//   q[b]   = sum_j softmax(psi0, psi1)[1][j] * reward[j]            (softmax over the two channels = sigmoid(psi1 - psi0))
//   loss   = [use_q] mean_b (q - q_t)^2 + [use_sf] mean_b mean_j (psi0 - sf_t)^2
//   dy     = d loss / d y.  One workgroup per batch row: the q reduction and the gradient that needs it stay together.
#define LOSS_THREADS 1024
#define LOSS_CACHE 4                    // pixels per thread kept in registers between the two passes (64x64 / 1024)
__global__ __launch_bounds__(LOSS_THREADS) void k_successor_loss(int batch, int px, int nf, const float* __restrict__ y,
                                                                 const float* __restrict__ reward,
                                                                 const int64_t* __restrict__ counter,
                                                                 const float* __restrict__ q_target_all,
                                                                 const float* __restrict__ sf_target_all, int use_q,
                                                                 int use_sf, float* __restrict__ dy,
                                                                 float* __restrict__ loss_rows, float* __restrict__ q_out,
                                                                 float* __restrict__ losses, int n_losses,
                                                                 int64_t* __restrict__ counter_inc, int32_t* __restrict__ ticket,
                                                                 float* __restrict__ adam_step) {
    __shared__ double s_q[LOSS_THREADS / 64], s_l[LOSS_THREADS / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int N = 2 * px + 2 * nf;
    float* dyr = dy + (size_t)b * N;
    if (b >= batch) {
        for (int j = t; j < N; j += LOSS_THREADS) dyr[j] = 0.f;
        if (t == 0) { __hip_atomic_store(&loss_rows[b], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q_out[b] = 0.f; }
        if (ticket) loss_log_ticket(batch, loss_rows, losses, n_losses, counter_inc, ticket, adam_step, t);
        return;
    }
    const float* yr = y + (size_t)b * N;
    const size_t src = (size_t)(*counter) * batch + b;
    const float* sft = use_sf ? sf_target_all + src * px : nullptr;
    // pass 1: q = sum_j sigmoid(psi1 - psi0) * reward, squared error of psi0; f64 sums (their order does not matter then)
    float sig[LOSS_CACHE], rw[LOSS_CACHE], err[LOSS_CACHE];
    double qs = 0.0, ls = 0.0;
#pragma unroll
    for (int i = 0; i < LOSS_CACHE; ++i) {
        const int j = t + i * LOSS_THREADS;
        sig[i] = 0.f; rw[i] = 0.f; err[i] = 0.f;
        if (j < px) {
            const float p0 = yr[j], p1 = yr[px + j];
            sig[i] = 1.f / (1.f + expf(p0 - p1));
            rw[i] = reward[j];
            if (use_sf) err[i] = p0 - sft[j];
            qs += (double)(sig[i] * rw[i]);
            ls += (double)(err[i] * err[i]);
        }
    }
    for (int j = t + LOSS_CACHE * LOSS_THREADS; j < px; j += LOSS_THREADS) {      // images larger than the cache
        const float p0 = yr[j], p1 = yr[px + j];
        const float sg = 1.f / (1.f + expf(p0 - p1));
        qs += (double)(sg * reward[j]);
        if (use_sf) { const float e = p0 - sft[j]; ls += (double)(e * e); }
    }
    qs = wave_sum_d(qs);
    ls = wave_sum_d(ls);
    if (lane == 0) { s_q[wave] = qs; s_l[wave] = ls; }
    __syncthreads();
    double qd = 0.0, ld = 0.0;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) { qd += s_q[w]; ld += s_l[w]; }
    const float q = (float)qd, lsf = (float)ld;
    const float inv_b = 1.f / (float)batch;
    const float dq = use_q ? 2.f * (q - q_target_all[src]) * inv_b : 0.f;
    const float csf = 2.f * inv_b / (float)px;
    // pass 2: d loss / d psi
#pragma unroll
    for (int i = 0; i < LOSS_CACHE; ++i) {
        const int j = t + i * LOSS_THREADS;
        if (j < px) {
            const float g1 = dq * rw[i] * (sig[i] * (1.f - sig[i]));
            dyr[px + j] = g1;
            dyr[j] = use_sf ? csf * err[i] - g1 : -g1;
        }
    }
    for (int j = t + LOSS_CACHE * LOSS_THREADS; j < px; j += LOSS_THREADS) {
        const float p0 = yr[j], p1 = yr[px + j];
        const float sg = 1.f / (1.f + expf(p0 - p1));
        const float g1 = dq * reward[j] * (sg * (1.f - sg));
        dyr[px + j] = g1;
        dyr[j] = use_sf ? csf * (p0 - sft[j]) - g1 : -g1;
    }
    for (int j = 2 * px + t; j < N; j += LOSS_THREADS) dyr[j] = 0.f;
    if (t == 0) {
        float l = 0.f;
        if (use_q) { const float d = q - q_target_all[src]; l += d * d * inv_b; }
        if (use_sf) l += lsf * inv_b / (float)px;
        __hip_atomic_store(&loss_rows[b], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through: read by the last arriver
        q_out[b] = q;
    }
    if (ticket) loss_log_ticket(batch, loss_rows, losses, n_losses, counter_inc, ticket, adam_step, t);
}

// losses[*counter] = sum_b loss_rows[b]; ++*counter.
__global__ void k_loss_log(int batch, const float* __restrict__ loss_rows, float* __restrict__ losses, int n_losses,
                           int64_t* __restrict__ counter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float l = 0.f;
        for (int b = 0; b < batch; ++b) l += loss_rows[b];
        const int64_t c = *counter;
        if (c >= 0 && c < n_losses) losses[c] = l;
        *counter = c + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The middle layers of the step (SuccessorMLP, successor_dqn.py:366: 256-128-64-128-256 -- four Linear + ReLU of 8-128 KB of
// weights on 32 rows) as ONE launch each way, WITHOUT any traffic between workgroups.  A launch per layer runs 6-8 us of
// which < 2 us is work: launch, first memory round trip, drain.  Here a workgroup per 32-column tile of the stack's LAST layer
// (8 of 1024 threads) computes everything that tile depends on itself -- the three layers in front of it in full, redundantly
// in every workgroup (768 of its 832 MFMAs), activations handed from layer to layer through LDS, __syncthreads between
// layers -- so there is no inter-workgroup protocol to get right and nothing to wait for across the chip.  Every weight
// fragment a workgroup needs (<= 80 VGPRs of its 1024 threads) and the biases are requested before the first MFMA: one
// memory latency per launch instead of one per layer.  Backward the same way round: a workgroup per 32-column tile of the
// gradient the stack hands down, the three input-gradient products above it in full in each; the 80 weight-gradient tiles
// (which only need the dz every workgroup holds) are dealt one to a wave.
// Measured alternatives, all bit-identical, none kept: eight workgroups with an in-launch barrier between layers (sc1
// hand-off): ~9 us per barrier, slower than the launches; ONE 1024-thread workgroup for everything: one CU's four matrix
// pipes are the floor (1280 MFMAs forward = 8.5 us, 2560 backward): 24 / 37 us kernels, slower too; round 2's 256-thread
// single workgroup.
// Tile by tile the arithmetic is that of k_lin_fwd (one split) / k_lin_bwd (one split, rows == 32): a tile's K (or N) range
// in four contiguous quarters, one wave each, summed ((q0 + q1) + q2) + q3 -- bit-identical results
// (tests/test_gpu_mlp_step.py::test_middle_layer_stack_equals_the_per_layer_launches).
#define MID_PAD 4                                            /* LDS row padding (floats): conflict-free 16-B row reads */
struct MidPtrs {
    const float* W[4]; const float* bias[4];
    float* dW[4]; float* db[4];
    float* act[5];               // act[l] [32][D_l]: input of middle layer l; act[l + 1] its output
    float* dz[5];                // dz[l + 1] [32][D_{l+1}]: gradient at layer l's pre-activation; dz[0]: handed below the stack
    // optional (backward): an Adam update of another layer's flat parameter range by the workgroups behind the stack's own --
    // the head's, whose gradient is complete and whose weights nothing reads any more when this launch runs.  59 MB of
    // streaming traffic that would otherwise sit in the step's last launch (where the first layer's folded update already
    // fills the memory system) runs beside the eight workgroups of the chain, which leave the memory system idle.
    float* rest_p; const float* rest_g; float* rest_m; float* rest_v; long long rest_n;
    const float* step; double lr, beta1, beta2, eps;
};
// forward job = (tile << 2) | quarter of a layer [K -> N]: weight fragments and the tile's bias value
template <int K>
__device__ __forceinline__ void mid_fwd_load(const float* __restrict__ W, const float* __restrict__ bias, int job, bool live, int lane,
                                             f4u (&w)[K / 32], float& bv) {
    if (live) {
        const float* wp = W + (size_t)((job >> 2) * 32 + (lane & 31)) * K + (job & 3) * (K / 4) + 4 * (lane >> 5);
#pragma unroll
        for (int u = 0; u < K / 32; ++u) w[u] = *reinterpret_cast<const f4u*>(wp + 8 * u);
        bv = bias[(job >> 2) * 32 + (lane & 31)];
    }
}
// x in LDS [32][K + MID_PAD] -> y = relu(x . W^T + b) of the job's tile into LDS [32][N + MID_PAD] (ys) and / or memory (yg)
template <int K, int N>
__device__ __forceinline__ void mid_fwd_job(const f4u (&w)[K / 32], float bv, int job, bool live, const float* xs, float* ys,
                                            float* __restrict__ yg, float (*red)[3][16][64], int lane, int rows_left = 32) {
    const int row = lane & 31, half = lane >> 5;
    const int nt = job >> 2, q = job & 3;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (live) {
        const float* xp = xs + row * (K + MID_PAD) + q * (K / 4) + 4 * half;
#pragma unroll
        for (int u = 0; u < K / 32; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(xp + 8 * u);
            acc = mfma32(a.x, w[u].x, acc);
            acc = mfma32(a.y, w[u].y, acc);
            acc = mfma32(a.z, w[u].z, acc);
            acc = mfma32(a.w, w[u].w, acc);
        }
        if (q > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[nt & 3][q - 1][r][lane] = acc[r];
        }
    }
    __syncthreads();
    if (live && q == 0) {
        const int n = nt * 32 + row;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = ((acc[r] + red[nt & 3][0][r][lane]) + red[nt & 3][1][r][lane]) + red[nt & 3][2][r][lane];
            v += bv;
            v = v > 0.f ? v : 0.f;
            const int m = mfma_row(r, lane);
            if (ys) ys[m * (N + MID_PAD) + n] = v;
            if (yg && m < rows_left) yg[(size_t)m * N + n] = v;
        }
    }
    __syncthreads();
}

// the same job with the result going to memory only, row stride y_stride, rows < rows_left
template <int K, int N>
__device__ __forceinline__ void mid_fwd_job_strided(const f4u (&w)[K / 32], float bv, int job, bool live, const float* xs,
                                                    float* __restrict__ yg, int64_t y_stride, float (*red)[3][16][64], int lane,
                                                    int rows_left) {
    const int row = lane & 31, half = lane >> 5;
    const int nt = job >> 2, q = job & 3;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (live) {
        const float* xp = xs + row * (K + MID_PAD) + q * (K / 4) + 4 * half;
#pragma unroll
        for (int u = 0; u < K / 32; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(xp + 8 * u);
            acc = mfma32(a.x, w[u].x, acc);
            acc = mfma32(a.y, w[u].y, acc);
            acc = mfma32(a.z, w[u].z, acc);
            acc = mfma32(a.w, w[u].w, acc);
        }
        if (q > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[nt & 3][q - 1][r][lane] = acc[r];
        }
    }
    __syncthreads();
    if (live && q == 0) {
        const int n = nt * 32 + row;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = ((acc[r] + red[nt & 3][0][r][lane]) + red[nt & 3][1][r][lane]) + red[nt & 3][2][r][lane];
            v += bv;
            v = v > 0.f ? v : 0.f;
            const int m = mfma_row(r, lane);
            if (m < rows_left) yg[(size_t)m * y_stride + n] = v;
        }
    }
    __syncthreads();
}

// grid = D4 / 32 workgroups of 1024 threads
template <int D0, int D1, int D2, int D3, int D4>
__global__ __launch_bounds__(1024) void k_mid_fwd(MidPtrs p) {
    constexpr int M01 = D0 > D1 ? D0 : D1, M23 = D2 > D3 ? D2 : D3, DM = M01 > M23 ? M01 : M23;
    __shared__ __attribute__((aligned(16))) float xa[32 * (DM + MID_PAD)], xb[32 * (DM + MID_PAD)];
    __shared__ float red[4][3][16][64];
    static_assert((D1 / 32) * 4 <= 16 && (D2 / 32) * 4 <= 16 && (D3 / 32) * 4 <= 16, "a layer in front of the last has at most 4 tiles");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool first = blockIdx.x == 0;                            // one workgroup writes the shared intermediate activations
    const bool l0 = wave < (D1 / 32) * 4, l1 = wave < (D2 / 32) * 4, l2 = wave < (D3 / 32) * 4, l3 = wave < 4;
    const int job3 = ((int)blockIdx.x << 2) | (wave & 3);
    f4u w0[D0 / 32], w1[D1 / 32], w2[D2 / 32], w3[D3 / 32];
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    mid_fwd_load<D0>(p.W[0], p.bias[0], wave, l0, lane, w0, b0);
    mid_fwd_load<D1>(p.W[1], p.bias[1], wave, l1, lane, w1, b1);
    mid_fwd_load<D2>(p.W[2], p.bias[2], wave, l2, lane, w2, b2);
    mid_fwd_load<D3>(p.W[3], p.bias[3], job3, l3, lane, w3, b3);
    for (int i = threadIdx.x; i < 32 * D0 / 4; i += 1024) {
        const int m = i / (D0 / 4), c = i % (D0 / 4);
        *reinterpret_cast<float4*>(&xa[m * (D0 + MID_PAD) + 4 * c]) = reinterpret_cast<const float4*>(p.act[0])[i];
    }
    __syncthreads();
    mid_fwd_job<D0, D1>(w0, b0, wave, l0, xa, xb, first ? p.act[1] : nullptr, red, lane);
    mid_fwd_job<D1, D2>(w1, b1, wave, l1, xb, xa, first ? p.act[2] : nullptr, red, lane);
    mid_fwd_job<D2, D3>(w2, b2, wave, l2, xa, xb, first ? p.act[3] : nullptr, red, lane);
    mid_fwd_job<D3, D4>(w3, b3, job3, l3, xb, nullptr, p.act[4], red, lane);
}

// The same layers for MANY rows (the acting / target forward between the first layer and the head, cv.py:20-38 inside
// SuccessorMLP.forward), two layers per launch: y = relu(relu(x' . Wa^T + ba) . Wb^T + bb) with x' = relu(x) if RELU_IN (x the
// pre-activation of the layer in front) else x.  A workgroup keeps the weight fragments of both layers in registers (all four
// layers' 160 registers per thread do not fit beside the accumulators at any workgroup size that fills the matrix pipes: 108
// spills at 16 waves, 39 at 8) and walks over 32-row tiles (grid-stride): tile into LDS, layer a into LDS, layer b to memory; the
// next tile's rows are requested while the current one computes.  Two launches replace four library GEMM calls (20-77 us of
// host time each), their ReLU passes and two of the three [n, 64..256] intermediates; tile by tile the arithmetic of k_mid_fwd /
// k_lin_fwd, so the acting forward and the training forward agree bit for bit on these layers.
template <int DA, int DB, int DC, bool RELU_IN>
__global__ __launch_bounds__(1024) void k_rows2(const float* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ Wb,
                                                const float* __restrict__ bb, int n_rows, const float* __restrict__ x, int64_t x_stride,
                                                float* __restrict__ y, int64_t y_stride) {
    constexpr int JA = (DB / 32) * 4, JB = (DC / 32) * 4, RB = (JB + 15) / 16, PER = 32 * DA / 1024;     // PER floats of a tile per thread
    static_assert(JA <= 16 && RB <= 2 && (PER == 8 || PER == 2), "layer widths outside what this kernel covers");
    __shared__ __attribute__((aligned(16))) float xa[32 * (DA + MID_PAD)], xb[32 * (DB + MID_PAD)];
    __shared__ float red[4][3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool la = wave < JA;
    f4u wa[DA / 32], wb[RB][DB / 32];
    float bva = 0.f, bvb[RB];
    mid_fwd_load<DA>(Wa, ba, wave, la, lane, wa, bva);
#pragma unroll
    for (int r = 0; r < RB; ++r) { bvb[r] = 0.f; mid_fwd_load<DB>(Wb, bb, r * 16 + wave, r * 16 + wave < JB, lane, wb[r], bvb[r]); }
    const int n_tiles = (n_rows + 31) / 32;
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    // this thread's PER consecutive floats of a tile: row e / DA, column e % DA with e = PER * threadIdx.x
    const int trow = (PER * threadIdx.x) / DA, tcol = (PER * threadIdx.x) % DA;
    float v[PER];
#define ROWS2_FETCH(t_)                                                                                        \
    {                                                                                                          \
        int r_ = (t_) * 32 + trow;                                                                             \
        r_ = r_ < n_rows ? r_ : n_rows - 1;                      /* rows beyond n read row n - 1, never stored */ \
        const float* src_ = x + (size_t)r_ * x_stride + tcol;                                                  \
        if constexpr (PER == 8) {                                                                              \
            const float4 q0_ = *reinterpret_cast<const float4*>(src_), q1_ = *reinterpret_cast<const float4*>(src_ + 4); \
            v[0] = q0_.x; v[1] = q0_.y; v[2] = q0_.z; v[3] = q0_.w; v[4] = q1_.x; v[5] = q1_.y; v[6] = q1_.z; v[7] = q1_.w; \
        } else {                                                                                               \
            const float2 q0_ = *reinterpret_cast<const float2*>(src_);                                         \
            v[0] = q0_.x; v[1] = q0_.y;                                                                        \
        }                                                                                                      \
    }
    ROWS2_FETCH(tile)
    for (; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < PER; ++j) xa[trow * (DA + MID_PAD) + tcol + j] = (RELU_IN && !(v[j] > 0.f)) ? 0.f : v[j];
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < n_tiles) ROWS2_FETCH(next)                          // in flight under this tile's layers
        mid_fwd_job<DA, DB>(wa, bva, wave, la, xa, xb, nullptr, red, lane);
        float* yt = y + (size_t)tile * 32 * y_stride;
        const int left = n_rows - tile * 32;
#pragma unroll
        for (int r = 0; r < RB; ++r) mid_fwd_job_strided<DB, DC>(wb[r], bvb[r], r * 16 + wave, r * 16 + wave < JB, xb, yt, y_stride, red, lane, left);
    }
#undef ROWS2_FETCH
}

// backward job = (k tile << 2) | quarter of N: W fragments B[kk = n][j = k]
template <int K, int N>
__device__ __forceinline__ void mid_dx_load(const float* __restrict__ W, int job, bool live, int lane, float (&w)[N / 8]) {
    if (live) {
        const float* wp = W + (size_t)((job & 3) * (N / 4) + 4 * (lane >> 5)) * K + (job >> 2) * 32 + (lane & 31);
#pragma unroll
        for (int u = 0; u < N / 32; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[4 * u + j] = wp[(size_t)(8 * u + j) * K];
        }
    }
}
// dz_below tile = (dz . W) * mask: dz in LDS [32][N + MID_PAD], mask bytes [32][K]; result to LDS (zs) and / or memory (zg)
template <int K, int N>
__device__ __forceinline__ void mid_dx_job(const float (&w)[N / 8], int job, bool live, const float* dzs, const unsigned char* mask,
                                           float* zs, float* __restrict__ zg, float (*red)[3][16][64], int lane) {
    const int row = lane & 31, half = lane >> 5;
    const int kt = job >> 2, q = job & 3;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (live) {
        const float* dp = dzs + row * (N + MID_PAD) + q * (N / 4) + 4 * half;
#pragma unroll
        for (int u = 0; u < N / 32; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(dp + 8 * u);
            acc = mfma32(a.x, w[4 * u + 0], acc);
            acc = mfma32(a.y, w[4 * u + 1], acc);
            acc = mfma32(a.z, w[4 * u + 2], acc);
            acc = mfma32(a.w, w[4 * u + 3], acc);
        }
        if (q > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[kt & 3][q - 1][r][lane] = acc[r];
        }
    }
    __syncthreads();
    if (live && q == 0) {
        const int k = kt * 32 + row;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = ((acc[r] + red[kt & 3][0][r][lane]) + red[kt & 3][1][r][lane]) + red[kt & 3][2][r][lane];
            const int m = mfma_row(r, lane);
            if (!mask[m * K + k]) v = 0.f;
            if (zs) zs[m * (K + MID_PAD) + k] = v;
            if (zg) zg[(size_t)m * K + k] = v;
        }
    }
    __syncthreads();
}
template <int K>
__device__ __forceinline__ void mid_mask_load(const float* __restrict__ act, unsigned char* mask) {    // [act > 0] as bytes
    for (int i = threadIdx.x; i < 32 * K / 4; i += 1024) {
        const float4 v = reinterpret_cast<const float4*>(act)[i];
        uchar4 m;
        m.x = v.x > 0.f; m.y = v.y > 0.f; m.z = v.z > 0.f; m.w = v.w > 0.f;
        reinterpret_cast<uchar4*>(mask)[i] = m;
    }
}
template <int K>
__device__ __forceinline__ void mid_dw_load(const float* __restrict__ act, int kt, int lane, float (&bv)[16]) {
    const float* p = act + (size_t)(lane >> 5) * K + kt * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 16; ++t) bv[t] = p[(size_t)(2 * t) * K];
}
template <int K, int N>
__device__ __forceinline__ void mid_dw_tile(const float* dzs, const float (&bv)[16], float* __restrict__ dW, float* __restrict__ db, int nt,
                                            int kt, int lane) {
    const int row = lane & 31, half = lane >> 5;
    float av[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) av[t] = dzs[(2 * t + half) * (N + MID_PAD) + nt * 32 + row];
    if (kt == 0) {
        float sdb = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) sdb += av[t];
        sdb += __shfl_xor(sdb, 32);
        if (half == 0) db[nt * 32 + row] = sdb;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = mfma32(av[t], bv[t], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) dW[(size_t)(nt * 32 + mfma_row(r, lane)) * K + kt * 32 + row] = acc[r];
}

// grid = D0 / 32 workgroups of 1024 threads; needs (D0 / 32) * 16 >= the number of weight-gradient tiles
template <int D0, int D1, int D2, int D3, int D4>
__global__ __launch_bounds__(1024) void k_mid_bwd(MidPtrs p) {
    __shared__ __attribute__((aligned(16))) float z4[32 * (D4 + MID_PAD)], z3[32 * (D3 + MID_PAD)], z2[32 * (D2 + MID_PAD)], z1[32 * (D1 + MID_PAD)];
    __shared__ float red[4][3][16][64];
    __shared__ __attribute__((aligned(16))) unsigned char m3[32 * D3], m2[32 * D2], m1[32 * D1], m0[32 * D0];
    constexpr int J3 = (D4 / 32) * (D3 / 32), J2 = (D3 / 32) * (D2 / 32), J1 = (D2 / 32) * (D1 / 32), J0 = (D1 / 32) * (D0 / 32);
    static_assert((D3 / 32) * 4 <= 16 && (D2 / 32) * 4 <= 16 && (D1 / 32) * 4 <= 16, "a gradient above the last has at most 4 tiles");
    static_assert(J3 + J2 + J1 + J0 <= (D0 / 32) * 16, "one weight-gradient tile per wave");
    if ((int)blockIdx.x >= D0 / 32) {                                 // the riders: flat Adam over the given range
        if (p.rest_n > 0) {
            // let the chain's eight workgroups put their weight / activation requests into the memory system first (their
            // latency is the launch's critical path; the riders have ~18 us for ~12 us of streaming)
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(127);
            const double t = (double)*p.step;
            const float step_size = (float)(p.lr / (1.0 - pow(p.beta1, t)));
            const float bc2_sqrt = (float)sqrt(1.0 - pow(p.beta2, t));
            const float w1 = (float)(1.0 - p.beta1), w2 = (float)(1.0 - p.beta2), b2 = (float)p.beta2, eps = (float)p.eps;
            const long long n4 = p.rest_n >> 2;
            const long long stride = (long long)(gridDim.x - D0 / 32) * blockDim.x;
            float4* p4 = reinterpret_cast<float4*>(p.rest_p);
            const float4* g4 = reinterpret_cast<const float4*>(p.rest_g);
            float4* m4 = reinterpret_cast<float4*>(p.rest_m);
            float4* v4 = reinterpret_cast<float4*>(p.rest_v);
            for (long long i = (long long)(blockIdx.x - D0 / 32) * blockDim.x + threadIdx.x; i < n4; i += stride) {
                float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
                adam_one(pp.x, gg.x, mm.x, vv.x, step_size, bc2_sqrt, b2, w1, w2, eps);
                adam_one(pp.y, gg.y, mm.y, vv.y, step_size, bc2_sqrt, b2, w1, w2, eps);
                adam_one(pp.z, gg.z, mm.z, vv.z, step_size, bc2_sqrt, b2, w1, w2, eps);
                adam_one(pp.w, gg.w, mm.w, vv.w, step_size, bc2_sqrt, b2, w1, w2, eps);
                p4[i] = pp; m4[i] = mm; v4[i] = vv;
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool l3 = wave < (D3 / 32) * 4, l2 = wave < (D2 / 32) * 4, l1 = wave < (D1 / 32) * 4, l0 = wave < 4;
    const int job0 = ((int)blockIdx.x << 2) | (wave & 3);
    float w3[D4 / 8], w2[D3 / 8], w1[D2 / 8], w0[D1 / 8];
    mid_dx_load<D3, D4>(p.W[3], wave, l3, lane, w3);
    mid_dx_load<D2, D3>(p.W[2], wave, l2, lane, w2);
    mid_dx_load<D1, D2>(p.W[1], wave, l1, lane, w1);
    mid_dx_load<D0, D1>(p.W[0], job0, l0, lane, w0);
    for (int i = threadIdx.x; i < 32 * D4 / 4; i += 1024) {
        const int m = i / (D4 / 4), c = i % (D4 / 4);
        *reinterpret_cast<float4*>(&z4[m * (D4 + MID_PAD) + 4 * c]) = reinterpret_cast<const float4*>(p.dz[4])[i];
    }
    mid_mask_load<D3>(p.act[3], m3);
    mid_mask_load<D2>(p.act[2], m2);
    mid_mask_load<D1>(p.act[1], m1);
    mid_mask_load<D0>(p.act[0], m0);
    __syncthreads();
    mid_dx_job<D3, D4>(w3, wave, l3, z4, m3, z3, nullptr, red, lane);
    // this wave's weight-gradient tile: layer 3 first, nt fastest inside a layer; its act fragments are requested now that the
    // largest set of W fragments is dead (128 VGPRs per thread at 1024 threads), and arrive under the rest of the chain
    const int g = (int)blockIdx.x * 16 + wave;
    float bv[16];
    if (g < J3) mid_dw_load<D3>(p.act[3], g / (D4 / 32), lane, bv);
    else if (g < J3 + J2) mid_dw_load<D2>(p.act[2], (g - J3) / (D3 / 32), lane, bv);
    else if (g < J3 + J2 + J1) mid_dw_load<D1>(p.act[1], (g - J3 - J2) / (D2 / 32), lane, bv);
    else if (g < J3 + J2 + J1 + J0) mid_dw_load<D0>(p.act[0], (g - J3 - J2 - J1) / (D1 / 32), lane, bv);
    mid_dx_job<D2, D3>(w2, wave, l2, z3, m2, z2, nullptr, red, lane);
    mid_dx_job<D1, D2>(w1, wave, l1, z2, m1, z1, nullptr, red, lane);
    mid_dx_job<D0, D1>(w0, job0, l0, z1, m0, nullptr, p.dz[0], red, lane);
    if (g < J3) mid_dw_tile<D3, D4>(z4, bv, p.dW[3], p.db[3], g % (D4 / 32), g / (D4 / 32), lane);
    else if (g < J3 + J2) mid_dw_tile<D2, D3>(z3, bv, p.dW[2], p.db[2], (g - J3) % (D3 / 32), (g - J3) / (D3 / 32), lane);
    else if (g < J3 + J2 + J1) mid_dw_tile<D1, D2>(z2, bv, p.dW[1], p.db[1], (g - J3 - J2) % (D2 / 32), (g - J3 - J2) / (D2 / 32), lane);
    else if (g < J3 + J2 + J1 + J0) mid_dw_tile<D0, D1>(z1, bv, p.dW[0], p.db[0], (g - J3 - J2 - J1) % (D1 / 32), (g - J3 - J2 - J1) / (D1 / 32), lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Head of the factored acting forward, fused: q[r] = sum_j w[j] * sigmoid(h[r,:] . Wd[j,:] + bd[j]) for n rows, K = 256
// hidden units, N = px outputs (cv.py:101-104 with channel 1 of the softmax over the two successor channels =
// sigmoid(psi1 - psi0); SuccessorMLP.q_from_first_layer).  The [n, 4096] product is never written: the library GEMM
// stored 737 MB for the 45 k candidate rows of a lock-step and k_sigmoid_dot read them back.
// One workgroup = 128 rows (a 32-row slab per wave, held in 128 VGPRs for the whole kernel: lane l keeps
// A[row = l & 31][k = 8u + 4 (l >> 5) + {0..3}], u < 32) x all N columns in tiles of 32: the tile of Wd (32 x 256) is staged
// in LDS (row stride 260 floats: conflict-free 16-B reads), double-buffered, the next tile's global loads in flight under
// this tile's 128 v_mfma_f32_32x32x2_f32; the sigmoid-weighted sum of a tile runs on the vector pipes from the accumulators.
#define HEAD_K 256
#define HEAD_BN 32
#define HEAD_LDS_STRIDE (HEAD_K + 4)
__global__ __launch_bounds__(256) void k_head_sigmoid_dot(int n_rows, int N, const float* __restrict__ h, int64_t h_stride,
                                                          const float* __restrict__ Wd, const float* __restrict__ bd,
                                                          const float* __restrict__ w, float* __restrict__ out,
                                                          int tiles_per_split) {
    __shared__ __attribute__((aligned(16))) float sB[2][HEAD_BN * HEAD_LDS_STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, half = lane >> 5;
    const int r0 = blockIdx.x * 128 + wave * 32;
    const int my_row = (r0 + row < n_rows) ? r0 + row : n_rows - 1;      // rows beyond n are computed on a valid row, never stored
    // A slab of this wave in registers
    f4u a[HEAD_K / 8];
    {
        const float* hp = h + (size_t)my_row * h_stride + 4 * half;
#pragma unroll
        for (int u = 0; u < HEAD_K / 8; ++u) a[u] = *reinterpret_cast<const f4u*>(hp + 8 * u);
    }
    // staging of a Wd tile: thread t copies 32 floats of row t >> 3
    const int srow = threadIdx.x >> 3, sseg = threadIdx.x & 7;
    float4 s0, s1, s2, s3, s4, s5, s6, s7;                               // named, not an array behind a lambda: that went to scratch
#define HEAD_LOAD_TILE(n0_)                                                                                             \
    {                                                                                                                   \
        const int nr_ = ((n0_) + srow < N) ? (n0_) + srow : N - 1;                                                      \
        const float4* src_ = reinterpret_cast<const float4*>(Wd + (size_t)nr_ * HEAD_K + sseg * 32);                    \
        s0 = src_[0]; s1 = src_[1]; s2 = src_[2]; s3 = src_[3]; s4 = src_[4]; s5 = src_[5]; s6 = src_[6]; s7 = src_[7]; \
    }
#define HEAD_STORE_TILE(buf_)                                                                                           \
    {                                                                                                                   \
        float4* dst_ = reinterpret_cast<float4*>(&sB[buf_][srow * HEAD_LDS_STRIDE + sseg * 32]);                        \
        dst_[0] = s0; dst_[1] = s1; dst_[2] = s2; dst_[3] = s3; dst_[4] = s4; dst_[5] = s5; dst_[6] = s6; dst_[7] = s7; \
    }
    float qacc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = 0.f;
    // blockIdx.y = a range of column tiles (the launch is cut so that the workgroups fill the CUs evenly); its sums go to
    // out[blockIdx.y * n_rows + r] and k_head_sum adds the ranges up in a fixed order
    const int all_tiles = (N + HEAD_BN - 1) / HEAD_BN;
    const int t_beg = blockIdx.y * tiles_per_split;
    const int n_tiles = (t_beg + tiles_per_split < all_tiles) ? t_beg + tiles_per_split : all_tiles;
    out += (size_t)blockIdx.y * n_rows;
    HEAD_LOAD_TILE(t_beg * HEAD_BN);
    HEAD_STORE_TILE(t_beg & 1);
    __syncthreads();
    for (int t = t_beg; t < n_tiles; ++t) {
        const int buf = t & 1, n0 = t * HEAD_BN;
        if (t + 1 < n_tiles) HEAD_LOAD_TILE(n0 + HEAD_BN);                  // in flight under the MFMAs below
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bp = &sB[buf][row * HEAD_LDS_STRIDE + 4 * half];
        float4 b[4], bn[4];                                             // four LDS reads ahead of the MFMAs that use them
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float4*>(bp + 8 * j);
#pragma unroll
        for (int u = 0; u < HEAD_K / 8; u += 4) {
            if (u + 4 < HEAD_K / 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const float4*>(bp + 8 * (u + 4 + j));
            }
            __builtin_amdgcn_sched_barrier(0);                          // keep the reads above the 16 MFMAs they hide under
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = mfma32(a[u + j].x, b[j].x, acc);
                acc = mfma32(a[u + j].y, b[j].y, acc);
                acc = mfma32(a[u + j].z, b[j].z, acc);
                acc = mfma32(a[u + j].w, b[j].w, acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = bn[j];
        }
        const int col = n0 + row;
        const float wj = col < N ? w[col] : 0.f, bj = col < N ? bd[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) qacc[r] += wj * __builtin_amdgcn_rcpf(1.f + __expf(-(acc[r] + bj)));
        if (t + 1 < n_tiles) HEAD_STORE_TILE(buf ^ 1);
        __syncthreads();
    }
    // sum over the 32 columns a half-wave holds (lanes of one half share the rows mfma_row(r, lane))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = qacc[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
        const int rr = r0 + mfma_row(r, lane);
        if (row == 0 && rr < n_rows) out[rr] = v;
    }
}

__global__ __launch_bounds__(256) void k_head_sum(int n_rows, int splits, const float* __restrict__ part, float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    float v = part[r];
    for (int s = 1; s < splits; ++s) v += part[(size_t)s * n_rows + r];
    out[r] = v;
}

}  // namespace bridges

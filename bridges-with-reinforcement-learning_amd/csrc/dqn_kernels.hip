// K7: fused DQN target / soft-update ops (robotoddler/training/successor_dqn.py).
#include "bridges_device.h"

namespace bridges {

// update_target_net (successor_dqn.py:280-288): target = policy * tau + target * (1 - tau), f32, 16 B per lane.
// The two products are rounded separately, as torch does (no FMA: -ffp-contract=off).
__global__ __launch_bounds__(256) void k_soft_update(float* __restrict__ target, const float* __restrict__ policy,
                                                     int64_t n, float tau, float omt) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* t4 = reinterpret_cast<float4*>(target);
    const float4* p4 = reinterpret_cast<const float4*>(policy);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 t = t4[i], p = p4[i];
        t.x = p.x * tau + t.x * omt;
        t.y = p.y * tau + t.y * omt;
        t.z = p.z * tau + t.z * omt;
        t.w = p.w * tau + t.w * omt;
        t4[i] = t;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        target[i] = policy[i] * tau + target[i] * omt;
}

// One Adam step (torch.optim.Adam, amsgrad = False, weight_decay = 0, maximize = False; the optimiser of
// successor_dqn.py:640) over ONE flat float32 buffer of parameters / gradients / moments, 16 B per lane:
//   m = m + (g - m) * (1 - beta1);  v = beta2 * v + (1 - beta2) * g * g;
//   p = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// (the operation order of torch's fused kernel).  t = *step, a device float the caller has ALREADY incremented for this
// step (the fused SuccessorMLP step does it in its loss kernel); the bias corrections are formed in float64.
// 7 x 4 B of traffic per parameter (torch's multi-tensor launch moves the same bytes at ~3.9 TB/s: 46 us for the 6.4 M
// parameters of SuccessorMLP; this launch streams them in one grid).
__global__ __launch_bounds__(256) void k_adam_flat(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const float* __restrict__ step, double lr,
                                                   double beta1_d, double beta2_d, double eps_d) {
    // every coefficient is formed in float64 from the optimiser's float64 hyper-parameters and rounded once (1 - 0.999f is
    // off by 5e-5 relative, which would show in v)
    const double t = (double)*step;
    const float step_size = (float)(lr / (1.0 - pow(beta1_d, t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2_d, t));
    const float w1 = (float)(1.0 - beta1_d), w2 = (float)(1.0 - beta2_d), beta2 = (float)beta2_d, eps = (float)eps_d;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
#define ADAM_ONE(P, G, M, V)                                         \
    do {                                                             \
        M = M + (G - M) * w1;                                        \
        V = beta2 * V + w2 * G * G;                                  \
        P = P - step_size * M / (sqrtf(V) / bc2_sqrt + eps);         \
    } while (0)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        ADAM_ONE(pp.x, gg.x, mm.x, vv.x);
        ADAM_ONE(pp.y, gg.y, mm.y, vv.y);
        ADAM_ONE(pp.z, gg.z, mm.z, vv.z);
        ADAM_ONE(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        ADAM_ONE(pp, gg, mm, vv);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
#undef ADAM_ONE
}

// The same update over MANY tensors in one launch (the conv Q-networks: 34-60 parameter tensors of 16 .. 295 k elements, each
// with its own gradient tensor from autograd).  One workgroup per 1024-element chunk of a tensor (chunk_slot / chunk_off name
// the tensor and the chunk's first element): 426 k parameters are ~450 workgroups.  torch's fused multi-tensor launch cuts
// tensors into 64 k-element chunks -- ~40 workgroups for ConvNet, 40 us per step where the bytes are worth 2.
// *step = number of updates done so far (this one is *step + 1; nothing in the launch writes *step, the caller advances it);
// the first chunk of a tensor also writes the new count into the tensor's own step word (torch.optim.Adam's state['step']).
__global__ __launch_bounds__(256) void k_adam_multi(const bridges_adam_slot* __restrict__ slots, const int32_t* __restrict__ chunk_slot,
                                                    const int32_t* __restrict__ chunk_off, const float* __restrict__ step, double lr,
                                                    double beta1_d, double beta2_d, double eps_d) {
    const bridges_adam_slot s = slots[chunk_slot[blockIdx.x]];
    const int64_t base = (int64_t)chunk_off[blockIdx.x] * 1024;
    const double t = (double)*step + 1.0;
    const float step_size = (float)(lr / (1.0 - pow(beta1_d, t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2_d, t));
    const float w1 = (float)(1.0 - beta1_d), w2 = (float)(1.0 - beta2_d), beta2 = (float)beta2_d, eps = (float)eps_d;
    if (base == 0 && threadIdx.x == 0 && s.step) *s.step = (float)t;
#define ADAM_ONE(P, G, M, V)                                         \
    do {                                                             \
        M = M + (G - M) * w1;                                        \
        V = beta2 * V + w2 * G * G;                                  \
        P = P - step_size * M / (sqrtf(V) / bc2_sqrt + eps);         \
    } while (0)
    const int64_t i = base + 4 * (int64_t)threadIdx.x;
    const bool vec = ((((uintptr_t)s.p) | ((uintptr_t)s.g) | ((uintptr_t)s.m) | ((uintptr_t)s.v)) & 15) == 0;
    if (vec && i + 3 < s.n) {
        float4 pp = *reinterpret_cast<float4*>(s.p + i), gg = *reinterpret_cast<const float4*>(s.g + i);
        float4 mm = *reinterpret_cast<float4*>(s.m + i), vv = *reinterpret_cast<float4*>(s.v + i);
        ADAM_ONE(pp.x, gg.x, mm.x, vv.x);
        ADAM_ONE(pp.y, gg.y, mm.y, vv.y);
        ADAM_ONE(pp.z, gg.z, mm.z, vv.z);
        ADAM_ONE(pp.w, gg.w, mm.w, vv.w);
        *reinterpret_cast<float4*>(s.p + i) = pp; *reinterpret_cast<float4*>(s.m + i) = mm; *reinterpret_cast<float4*>(s.v + i) = vv;
    } else {
        for (int64_t j = i; j < i + 4 && j < s.n; ++j) {
            float pp = s.p[j], gg = s.g[j], mm = s.m[j], vv = s.v[j];
            ADAM_ONE(pp, gg, mm, vv);
            s.p[j] = pp; s.m[j] = mm; s.v[j] = vv;
        }
    }
#undef ADAM_ONE
}

// train_policy_net target construction (successor_dqn.py:197-213, 222, 230).  One workgroup per transition:
// segmented first-argmax over its next-action rows, then the q target and (optionally) the successor-feature
// target row, 16 B per lane.
__global__ __launch_bounds__(256) void k_td_target(int n_trans, const int32_t* __restrict__ seg_lo, const int32_t* __restrict__ seg_hi,
                                                   const float* __restrict__ next_q, const float* __restrict__ next_sf,
                                                   int64_t sf_row_stride, const float* __restrict__ action_raster,
                                                   const float* __restrict__ lin_reward, const uint8_t* __restrict__ done,
                                                   float gamma, int sf_dim, float* __restrict__ q_target,
                                                   float* __restrict__ sf_target, int32_t* __restrict__ argmax_row) {
    __shared__ float s_val[256];
    __shared__ int s_idx[256];
    const int i = blockIdx.x, t = threadIdx.x;
    const int lo = seg_lo[i], hi = seg_hi[i];
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int j = lo + t; j < hi; j += 256) {
        float v = next_q[j];
        if (v > best || (v == best && j < bidx) || bidx == 0x7fffffff) { best = v; bidx = j; }   // NaN-free inputs
    }
    s_val[t] = best; s_idx[t] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            float v = s_val[t + o]; int k = s_idx[t + o];
            if (k != 0x7fffffff && (s_idx[t] == 0x7fffffff || v > s_val[t] || (v == s_val[t] && k < s_idx[t]))) {
                s_val[t] = v; s_idx[t] = k;
            }
        }
        __syncthreads();
    }
    const int row = s_idx[0];
    const bool dn = done[i] != 0;
    if (t == 0) {
        float nq = dn ? 0.f : s_val[0];
        q_target[i] = lin_reward[i] + gamma * nq;
        argmax_row[i] = row;
    }
    if (sf_dim > 0) {
        const float* src = next_sf + (int64_t)row * sf_row_stride;
        const float* ar = action_raster + (int64_t)i * sf_dim;
        float* dst = sf_target + (int64_t)i * sf_dim;
        const int n4 = sf_dim >> 2;
        for (int k = t; k < n4; k += 256) {
            float4 a = reinterpret_cast<const float4*>(ar)[k];
            float4 s = dn ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(src)[k];
            float4 o;
            o.x = a.x + gamma * s.x; o.y = a.y + gamma * s.y; o.z = a.z + gamma * s.z; o.w = a.w + gamma * s.w;
            reinterpret_cast<float4*>(dst)[k] = o;
        }
        for (int k = (n4 << 2) + t; k < sf_dim; k += 256) dst[k] = ar[k] + gamma * (dn ? 0.f : src[k]);
    }
}

// K8: first layer of an MLP that consumes flattened binary 64x64 rasters, fed with the bit-packed rasters themselves
// (cv.py:95-97 concatenates block / action / reward / obstacle images and multiplies by W1; a block covers ~35 of the
// 4096 pixels, so the product with its raster is the sum of ~35 rows of the transposed weight slice):
//   out[r, :] = base[base_row[r], :] + sum over set pixels p of bits[bits_row[r]] (ascending p) of wt[p, :]
// One wave per output row, 16 B per lane; wt (4 MiB for d = 256) stays in L2.
__global__ __launch_bounds__(256) void k_bits_linear(int n_rows, const uint64_t* __restrict__ bits,
                                                     const int64_t* __restrict__ bits_row, const float* __restrict__ wt,
                                                     int d, const float* __restrict__ base,
                                                     const int64_t* __restrict__ base_row, float* __restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    for (int rv = wave; rv < n_rows; rv += nwaves) {
        const int r = __builtin_amdgcn_readfirstlane(rv);
        const int64_t src = bits_row ? bits_row[r] : (int64_t)r;
        const uint64_t mine = bits[(size_t)src * IMG + lane];               // image row `lane`
        const uint64_t rows_nz = __ballot(mine != 0ull);
        const float* b = base ? base + (size_t)(base_row ? base_row[r] : 0) * d : nullptr;
        for (int c0 = 0; c0 < d; c0 += 4 * WAVE) {
            const int col = c0 + 4 * lane;
            const bool act = col < d;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b && act) acc = *reinterpret_cast<const float4*>(b + col);
            uint64_t rem = rows_nz;
            while (rem) {
                const int rr = __builtin_ctzll(rem);
                rem &= rem - 1ull;
                uint64_t m = shfl_u64(mine, rr);                             // uniform: the mask of image row rr
                const float* w = wt + (size_t)rr * IMG * d + col;
                while (m) {
                    const int cc = __builtin_ctzll(m);
                    m &= m - 1ull;
                    if (act) {
                        const float4 v = *reinterpret_cast<const float4*>(w + (size_t)cc * d);
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                }
            }
            if (act) *reinterpret_cast<float4*>(out + (size_t)r * d + col) = acc;
        }
    }
}

// Count-based exploration of EpsilonGreedy (successor_dqn.py:112-131) on the bit-packed rasters, both directions:
//   k_bits_dot:        out[r] = sum over the set pixels of bits[bits_row[r]] of img[slot[r]]  (= sum(step_images[step] * a_r))
//   k_bits_accumulate: img[slot[r]] += weight[r] * raster(bits[bits_row[r]])                  (= step_images[step] += a_sel)
// One wave per row, lane = image row: a raster holds ~35 pixels, so a lane walks the handful of set bits of its own row
// instead of a [n, 4096] product / a [n, 64, 64] float image.  The images hold small integer counts: float sums of them
// are exact in any order, so neither the wave reduction nor the float atomics change a bit of the result.
__global__ __launch_bounds__(256) void k_bits_dot(int n_rows, const uint64_t* __restrict__ bits, const int64_t* __restrict__ bits_row,
                                                  const float* __restrict__ img, const int64_t* __restrict__ slot,
                                                  float* __restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    for (int rv = wave; rv < n_rows; rv += nwaves) {
        const int r = __builtin_amdgcn_readfirstlane(rv);
        const int64_t src = bits_row ? bits_row[r] : (int64_t)r;
        uint64_t m = bits[(size_t)src * IMG + lane];
        const float* row = img + ((size_t)slot[r] * IMG + lane) * IMG;
        double acc = 0.0;
        while (m) {
            const int x = __builtin_ctzll(m);
            m &= m - 1ull;
            acc += (double)row[x];
        }
        acc = wave_sum_d(acc);
        if (lane == 0) out[r] = (float)acc;
    }
}

__global__ __launch_bounds__(256) void k_bits_accumulate(int n_rows, const uint64_t* __restrict__ bits,
                                                         const int64_t* __restrict__ bits_row, const float* __restrict__ weight,
                                                         const int64_t* __restrict__ slot, float* __restrict__ img) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    for (int rv = wave; rv < n_rows; rv += nwaves) {
        const int r = __builtin_amdgcn_readfirstlane(rv);
        const float w = weight ? weight[r] : 1.f;
        if (w == 0.f) continue;
        const int64_t src = bits_row ? bits_row[r] : (int64_t)r;
        uint64_t m = bits[(size_t)src * IMG + lane];
        float* row = img + ((size_t)slot[r] * IMG + lane) * IMG;
        while (m) {
            const int x = __builtin_ctzll(m);
            m &= m - 1ull;
            atomicAdd(row + x, w);
        }
    }
}

// Head of the factored acting forward: q[r] = sum_j w[j] * sigmoid(d[r, j])  (cv.py:101-104: softmax over the two successor
// channels, channel 1, times the reward map, summed over the image) in ONE pass over d instead of sigmoid / mul / sum
// passes.  One wave per row, 16 B per lane per trip; lane partial sums in f32, the 64 partials added in f64.
__global__ __launch_bounds__(256) void k_sigmoid_dot(int n_rows, const float* __restrict__ d, int64_t row_stride,
                                                     const float* __restrict__ w, int k, float* __restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    for (int r = wave; r < n_rows; r += nwaves) {
        const float* row = d + (size_t)r * row_stride;
        float acc = 0.f;
        // four 16-B loads of the row in flight per lane and trip (a 64 x 64 image: four trips), the sigmoid through the
        // hardware reciprocal (1 ulp; a full float32 division is ~10 instructions per pixel and made the pass ALU-heavy)
#define SIGDOT_ONE(X, WW)                                                     \
    acc += WW.x * __builtin_amdgcn_rcpf(1.f + __expf(-X.x));                 \
    acc += WW.y * __builtin_amdgcn_rcpf(1.f + __expf(-X.y));                 \
    acc += WW.z * __builtin_amdgcn_rcpf(1.f + __expf(-X.z));                 \
    acc += WW.w * __builtin_amdgcn_rcpf(1.f + __expf(-X.w));
        int j = 4 * lane;
        for (; j + 12 * WAVE < k; j += 16 * WAVE) {
            const float4 x0 = *reinterpret_cast<const float4*>(row + j), x1 = *reinterpret_cast<const float4*>(row + j + 4 * WAVE);
            const float4 x2 = *reinterpret_cast<const float4*>(row + j + 8 * WAVE), x3 = *reinterpret_cast<const float4*>(row + j + 12 * WAVE);
            const float4 w0 = *reinterpret_cast<const float4*>(w + j), w1 = *reinterpret_cast<const float4*>(w + j + 4 * WAVE);
            const float4 w2 = *reinterpret_cast<const float4*>(w + j + 8 * WAVE), w3 = *reinterpret_cast<const float4*>(w + j + 12 * WAVE);
            SIGDOT_ONE(x0, w0) SIGDOT_ONE(x1, w1) SIGDOT_ONE(x2, w2) SIGDOT_ONE(x3, w3)
        }
        for (; j < k; j += 4 * WAVE) {
            const float4 x = *reinterpret_cast<const float4*>(row + j);
            const float4 ww = *reinterpret_cast<const float4*>(w + j);
            SIGDOT_ONE(x, ww)
        }
#undef SIGDOT_ONE
        const double total = wave_sum_d((double)acc);
        if (lane == 0) out[r] = (float)total;
    }
}

// ---- inference epilogues of the conv nets (robotoddler/models/cv.py: Conv2d -> ReLU [-> MaxPool2d(2)]) -------------
// torch runs "+ bias", "relu" and "maxpool" as three passes over the activation tensor (9.3 GB of traffic per 2048
// rows of the ConvNet, a third of its forward time); these do them in one.  Adding the channel's bias and clamping at
// zero are monotone, so max(x_i) + b -> relu gives bit for bit what pool(relu(x_i + b)) gives.
// x [n, C, hw] contiguous (NCHW), hw % 4 == 0; in place.
__global__ __launch_bounds__(256) void k_bias_relu(float* __restrict__ x, const float* __restrict__ bias, int64_t n4,
                                                   int hw4, int C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* x4 = reinterpret_cast<float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float b = bias[(i / hw4) % C];
        float4 v = x4[i];
        v.x = fmaxf(v.x + b, 0.f); v.y = fmaxf(v.y + b, 0.f); v.z = fmaxf(v.z + b, 0.f); v.w = fmaxf(v.w + b, 0.f);
        x4[i] = v;
    }
}

// out[nc, H/2, W/2] = relu(max over the 2x2 window of x[nc, H, W] + bias[c]); W % 4 == 0, H % 2 == 0.
// One thread: 4 input columns of 2 rows -> 2 outputs.
__global__ __launch_bounds__(256) void k_bias_relu_pool2(const float* __restrict__ x, const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t n_items, int H, int W, int C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int w4 = W >> 2, h2 = H >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) {
        const int64_t q = i % w4, rest = i / w4;
        const int64_t r = rest % h2, nc = rest / h2;
        const float b = bias[nc % C];
        const float* p = x + (nc * H + 2 * r) * W + 4 * q;
        const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + W);
        float2 o;
        o.x = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a1.x, a1.y)) + b, 0.f);
        o.y = fmaxf(fmaxf(fmaxf(a0.z, a0.w), fmaxf(a1.z, a1.w)) + b, 0.f);
        *reinterpret_cast<float2*>(out + (nc * h2 + r) * (W >> 1) + 2 * q) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// EpsilonGreedy.select (successor_dqn.py:98-132) for every env of the vectorised loop in one launch: per env, over its rows
// seg[e] .. seg[e + 1] of the Q pass, the greedy row = first maximum of q, the exploring row = first minimum of the overlap
// `join` of the candidate with the count image of the env's episode step (count-based exploration); the env explores when
// its uniform draw u[e] <= eps (and the call is not greedy).  An env without rows gets row 0 with weight 0.  The rows of
// env e are seg_lo[e] .. seg_hi[e] (envs in the same state share their representative's rows, rep[e]: the compact index of
// the chosen row is then a candidate of the REPRESENTATIVE, whose candidates are env e's own, one for one).  Replaces two
// segmented arg-max launches and ~20 element-wise / index launches.  One wave per env.
//   sel_compact[e] = idx[row]  (compact candidate index), sel_index[e] = sel_compact - cand_offset[rep[e]] (>= 0),
//   q_sel[e] = q[row] (0 without rows), explore_w[e] = 1.0 if the env explored and has rows else 0.0
__global__ __launch_bounds__(256) void k_eps_greedy_select(int E, int n_rows, const int32_t* __restrict__ seg_lo,
                                                           const int32_t* __restrict__ seg_hi, const float* __restrict__ q,
                                                           const float* __restrict__ join, const float* __restrict__ u, float eps,
                                                           int greedy, const int64_t* __restrict__ idx,
                                                           const int32_t* __restrict__ cand_offset, const int32_t* __restrict__ rep,
                                                           int64_t* __restrict__ sel_compact,
                                                           int32_t* __restrict__ sel_index, float* __restrict__ q_sel,
                                                           float* __restrict__ explore_w) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const int lo = seg_lo[e], hi = seg_hi[e];
    const bool explore = !greedy && u[e] <= eps;
    // first maximum of q, or of -join (= first minimum of join), over the env's rows
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lo + lane; j < hi; j += 64) {
        const float v = explore ? -join[j] : q[j];
        if (bi == 0x7fffffff || v > best || (v == best && j < bi)) { best = v; bi = j; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float v = __shfl_xor(best, m);
        const int k = __shfl_xor(bi, m);
        if (k != 0x7fffffff && (bi == 0x7fffffff || v > best || (v == best && k < bi))) { best = v; bi = k; }
    }
    if (lane == 0) {
        const bool has = hi > lo;
        int row = has ? bi : 0;
        if (row > n_rows - 1) row = n_rows - 1;
        const int64_t c = idx[row];
        sel_compact[e] = c;
        const int64_t rel = c - (int64_t)cand_offset[rep ? rep[e] : e];
        sel_index[e] = rel > 0 ? (int32_t)rel : 0;
        q_sel[e] = has ? q[row] : 0.f;
        explore_w[e] = (explore && has) ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Transition records of the vectorised loop (robotoddler/training/records.py: one float64 row per transition, the
// compact form of the reference's Transition, successor_dqn.py:27-44; layout BRIDGES_REC_* of the header).  The torch
// formulation of these three steps was ~95 slice / cast / index launches of a few microseconds per lock-step.
//   k_record_state:   before step(): the state s of every env and the chosen candidate -> rec[:, 0 : REC_REWARD], stable(s)
//   k_record_result:  after step():  reward, lin_reward, done | no_actions, stable(s') -> rec, valid_step -> valid
//   k_replay_unpack:  sampled records -> the replay env's state arrays holding s' = s + the action block with the
//                     occupancy update of gym_env.py:228-232, the candidate count of s', the block ranges of s' and s
// One 64-lane workgroup per env / record, lane k = block slot k.
__global__ __launch_bounds__(64) void k_record_state(int E, int K, const int32_t* __restrict__ n_blocks,
                                                     const int32_t* __restrict__ blk_shape, const double* __restrict__ blk_pose,
                                                     const uint8_t* __restrict__ blk_occ, const uint8_t* __restrict__ step_flags,
                                                     const int64_t* __restrict__ sel_row, const int32_t* __restrict__ cand_desc,
                                                     const double* __restrict__ cand_pose, double* __restrict__ rec) {
    const int e = blockIdx.x, k = threadIdx.x;
    if (e >= E) return;
    double* r = rec + (size_t)e * BRIDGES_REC_WIDTH;
    for (int c = k; c < BRIDGES_REC_WIDTH; c += 64) r[c] = 0.0;        // slots >= K, td_error
    __syncthreads();
    if (k < K) {
        r[BRIDGES_REC_SHAPE + k] = (double)blk_shape[(size_t)e * K + k];
        const double* p = blk_pose + ((size_t)e * K + k) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[BRIDGES_REC_POSE + 4 * k + j] = p[j];
        r[BRIDGES_REC_OCC + k] = (double)blk_occ[(size_t)e * K + k];
    }
    if (k == 0) {
        const int nb = n_blocks[e];
        r[BRIDGES_REC_NB] = (double)nb;
        const int64_t row = sel_row[e];
        const int32_t* d = cand_desc + row * 4;                            // (target_block, target_face, shape, face)
        r[BRIDGES_REC_ASHAPE] = (double)d[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[BRIDGES_REC_APOSE + j] = cand_pose[row * 4 + j];
        r[BRIDGES_REC_ATB] = (double)d[0];
        r[BRIDGES_REC_ATF] = (double)d[1];
        r[BRIDGES_REC_AFACE] = (double)d[3];
        // 'stable' of s: a freshly reset env is stable (stability.py:53-56), else the frozen verdict of the last step
        r[BRIDGES_REC_STABLE_S] = (nb == 0 || step_flags[(size_t)e * 8 + 1]) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(256) void k_record_result(int E, const float* __restrict__ reward, const float* __restrict__ lin_reward,
                                                       const uint8_t* __restrict__ step_flags, double* __restrict__ rec,
                                                       uint8_t* __restrict__ valid) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const uint8_t* f = step_flags + (size_t)e * 8;
    double* r = rec + (size_t)e * BRIDGES_REC_WIDTH;
    r[BRIDGES_REC_REWARD] = (double)reward[e];
    r[BRIDGES_REC_LIN] = (double)lin_reward[e];
    r[BRIDGES_REC_DONE] = (f[5] | f[6]) ? 1.0 : 0.0;                     // done | no_actions
    r[BRIDGES_REC_STABLE_N] = f[1] ? 1.0 : 0.0;
    valid[e] = f[0] != 0;
}

__global__ __launch_bounds__(64) void k_replay_unpack(int E, int n_rec, int K, const double* __restrict__ rec,
                                                      const int32_t* __restrict__ shape_faces, int n_shapes, int n_groups, int n_ground, int n_off,
                                                      int32_t* __restrict__ n_blocks, int32_t* __restrict__ blk_shape,
                                                      double* __restrict__ blk_pose, uint8_t* __restrict__ blk_occ,
                                                      int32_t* __restrict__ n_cand, int32_t* __restrict__ ranges_next,
                                                      int32_t* __restrict__ ranges_prev, float* __restrict__ lin,
                                                      float* __restrict__ stable_s, uint8_t* __restrict__ done,
                                                      uint8_t* __restrict__ stable_n) {
    const int e = blockIdx.x, k = threadIdx.x;
    if (e >= E) return;
    const double* r = rec + (size_t)(e < n_rec ? e : 0) * BRIDGES_REC_WIDTH;     // padding envs repeat the first record
    const int nb = (int)r[BRIDGES_REC_NB];
    const int slot = nb < K - 1 ? nb : K - 1;
    const int tb = (int)r[BRIDGES_REC_ATB];
    int free_faces = 0;
    if (k < K) {
        int shape = (int)r[BRIDGES_REC_SHAPE + k];
        int occ = (int)r[BRIDGES_REC_OCC + k] & 255;
        double p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = r[BRIDGES_REC_POSE + 4 * k + j];
        if (k == slot) {                                                   // the action block
            shape = (int)r[BRIDGES_REC_ASHAPE];
            occ = (1 << (int)r[BRIDGES_REC_AFACE]) & 255;
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = r[BRIDGES_REC_APOSE + j];
        }
        if (tb >= 0 && k == tb) occ |= (1 << (int)r[BRIDGES_REC_ATF]) & 255;   // the face it was put on
        blk_shape[(size_t)e * K + k] = shape;
        blk_occ[(size_t)e * K + k] = (uint8_t)occ;
#pragma unroll
        for (int j = 0; j < 4; ++j) blk_pose[((size_t)e * K + k) * 4 + j] = p[j];
        if (k <= nb && k < K) free_faces = shape_faces[shape < 0 ? 0 : (shape < n_shapes ? shape : n_shapes - 1)] - __builtin_popcount(occ & ((1 << BRIDGES_MAX_VERTS) - 1));
    }
    int nfree = free_faces;                                                // lanes >= K hold 0
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) nfree += __shfl_xor(nfree, m);
    if (k == 0) {
        n_blocks[e] = nb + 1;
        n_cand[e] = n_groups * (n_ground + nfree * n_off);     // raw count: bridges_env_refresh clamps to the env's a_max and flags it
        ranges_next[2 * e] = e * K;
        ranges_next[2 * e + 1] = e * K + nb + 1;
        ranges_prev[2 * e] = e * K;
        ranges_prev[2 * e + 1] = e * K + nb;
        lin[e] = (float)r[BRIDGES_REC_LIN];
        stable_s[e] = (float)r[BRIDGES_REC_STABLE_S];
        done[e] = r[BRIDGES_REC_DONE] > 0.5;
        stable_n[e] = r[BRIDGES_REC_STABLE_N] > 0.5;
    }
}

}  // namespace bridges

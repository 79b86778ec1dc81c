// Stand-alone operators: the same device code as the lock-step kernels, on caller-shaped batches.
// Used by the single-environment assembly_gym API, by replay re-rasterisation and by the parity tests.
#include "bridges_device.h"
#include "rbe_device.h"

namespace bridges {

// K1: create_block / align_frames_2d (gym_env.py:204-216, geometry.py:39-50); one thread per placement.
__global__ void k_place(const bridges_shape* shapes, int n, const double* frame1, const int32_t* shape_id,
                        const int32_t* face, const double* ox, const double* oy, double* pose, double* verts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Frame2 f1;
    f1.cx = frame1[6 * i + 0]; f1.cz = frame1[6 * i + 1];
    f1.tx = frame1[6 * i + 2]; f1.tz = frame1[6 * i + 3];
    f1.nx = frame1[6 * i + 4]; f1.nz = frame1[6 * i + 5];
    const bridges_shape& s = shapes[shape_id[i]];
    int f = face[i];
    double px, pz, c, sn;
    align_place(f1, s.fcx[f], s.fcz[f], s.fnx[f], s.fnz[f], ox[i], oy[i], px, pz, c, sn);
    pose[4 * i + 0] = px; pose[4 * i + 1] = pz; pose[4 * i + 2] = c; pose[4 * i + 3] = sn;
    for (int k = 0; k < MAXV; ++k) {
        double wx = 0.0, wz = 0.0;
        if (k < s.nv) {
            double rx, rz;
            rot2(s.vx[k], s.vz[k], c, sn, rx, rz);
            wx = px + rx; wz = pz + rz;
        }
        verts[(size_t)i * MAXV * 2 + 2 * k] = wx;
        verts[(size_t)i * MAXV * 2 + 2 * k + 1] = wz;
    }
}

// K1 as AssemblyGym.create_block states it (gym_env.py:204-216): the target frame is the floor (target_face < 0,
// assembly_env.py:339-340) or face `target_face` of a posed block given by its world vertices.
__global__ void k_create_block(const bridges_shape* shapes, int n, const double* target_verts,
                               const int32_t* target_shape, const int32_t* target_face, const int32_t* shape_id,
                               const int32_t* face, const double* ox, const double* oy, double* pose, double* verts,
                               double* frame_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Frame2 f1;
    if (target_face[i] < 0) {
        f1.cx = 0.0; f1.cz = 0.0; f1.tx = 1.0; f1.tz = 0.0; f1.nx = 0.0; f1.nz = 1.0;
    } else {
        const bridges_shape& st = shapes[target_shape[i]];
        const double* v = target_verts + (size_t)i * MAXV * 2;
        const int tf = target_face[i];
        f1 = edge_frame(v[2 * st.fa[tf]], v[2 * st.fa[tf] + 1], v[2 * st.fb[tf]], v[2 * st.fb[tf] + 1]);
    }
    if (frame_out) {
        double* fo = frame_out + 6 * i;
        fo[0] = f1.cx; fo[1] = f1.cz; fo[2] = f1.tx; fo[3] = f1.tz; fo[4] = f1.nx; fo[5] = f1.nz;
    }
    const bridges_shape& s = shapes[shape_id[i]];
    int f = face[i];
    double px, pz, c, sn;
    align_place(f1, s.fcx[f], s.fcz[f], s.fnx[f], s.fnz[f], ox[i], oy[i], px, pz, c, sn);
    pose[4 * i + 0] = px; pose[4 * i + 1] = pz; pose[4 * i + 2] = c; pose[4 * i + 3] = sn;
    for (int k = 0; k < MAXV; ++k) {
        double wx = 0.0, wz = 0.0;
        if (k < s.nv) {
            double rx, rz;
            rot2(s.vx[k], s.vz[k], c, sn, rx, rz);
            wx = px + rx; wz = pz + rz;
        }
        verts[(size_t)i * MAXV * 2 + 2 * k] = wx;
        verts[(size_t)i * MAXV * 2 + 2 * k + 1] = wz;
    }
}

// Pose a shape directly: world vertices = pos + R(c,s) * local vertices (Block.__init__, assembly_env.py:146-153).
__global__ void k_pose_block(const bridges_shape* shapes, int n, const int32_t* shape_id, const double* pose, double* verts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bridges_shape& s = shapes[shape_id[i]];
    const double px = pose[4 * i], pz = pose[4 * i + 1], c = pose[4 * i + 2], sn = pose[4 * i + 3];
    for (int k = 0; k < MAXV; ++k) {
        double wx = 0.0, wz = 0.0;
        if (k < s.nv) {
            double rx, rz;
            rot2(s.vx[k], s.vz[k], c, sn, rx, rz);
            wx = px + rx; wz = pz + rz;
        }
        verts[(size_t)i * MAXV * 2 + 2 * k] = wx;
        verts[(size_t)i * MAXV * 2 + 2 * k + 1] = wz;
    }
}

// World face frames of posed blocks: [n,6,6] (centre.xz, tangent.xz, normal.xz) -- Shape.get_face_frame_2d on a
// Block (assembly_env.py:118-124).
__global__ void k_face_frames(const bridges_shape* shapes, int n, const int32_t* shape_id, const double* verts, double* frames) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bridges_shape& s = shapes[shape_id[i]];
    const double* v = verts + (size_t)i * MAXV * 2;
    for (int f = 0; f < MAXV; ++f) {
        double* o = frames + ((size_t)i * MAXV + f) * 6;
        if (f < s.nv) {
            Frame2 fr = edge_frame(v[2 * s.fa[f]], v[2 * s.fa[f] + 1], v[2 * s.fb[f]], v[2 * s.fb[f] + 1]);
            o[0] = fr.cx; o[1] = fr.cz; o[2] = fr.tx; o[3] = fr.tz; o[4] = fr.nx; o[5] = fr.nz;
        } else {
            for (int k = 0; k < 6; ++k) o[k] = 0.0;
        }
    }
}

// Shape.contains_2d (assembly_env.py:126-137) for arbitrary sample points: inside[i] = all faces of the posed outline
// satisfy ((p.x - c.x) * n.x) + ((p.z - c.z) * n.z) <= 0.  One thread per point, one posed block per call.
__global__ void k_contains_points(const bridges_shape* shapes, int shape_id, const double* verts /*[6,2]*/, int n,
                                  const double* points /*[n,2]*/, uint8_t* inside) {
    __shared__ double fr[MAXV][4];
    const bridges_shape& s = shapes[shape_id];
    if (threadIdx.x < s.nv) {
        const int f = threadIdx.x;
        Frame2 e = edge_frame(verts[2 * s.fa[f]], verts[2 * s.fa[f] + 1], verts[2 * s.fb[f]], verts[2 * s.fb[f] + 1]);
        fr[f][0] = e.cx; fr[f][1] = e.cz; fr[f][2] = e.nx; fr[f][3] = e.nz;
    }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double px = points[2 * i], pz = points[2 * i + 1];
    bool in = true;
    for (int f = 0; f < s.nv; ++f) {
        double d = (px - fr[f][0]) * fr[f][2] + (pz - fr[f][1]) * fr[f][3];
        in = in && (d <= 0.0);
    }
    inside[i] = in;
}

// K4: one wave per posed outline (world vertices in shape-vertex order).
__global__ __launch_bounds__(256) void k_raster_generic(const bridges_shape* shapes, int n, const double* verts,
                                                        const int32_t* shape_id, const double* gx, const double* gy,
                                                        int size, uint64_t* bits, float* img) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    for (int it = wave; it < n; it += nwaves) {
        const bridges_shape& sh = shapes[shape_id[it]];
        uint64_t b = raster_outline(verts + (size_t)it * MAXV * 2, sh.nv, sh.fa, sh.fb, gx, gy, size, lane);
        if (bits) bits[(size_t)it * IMG + lane] = b;
        if (img) write_f32_image(img + (size_t)it * IMG * IMG, b, lane);
    }
}

// render_blocks_2d (rendering.py:105-113) at ANY image size: the union of n posed outlines on a W x H pixel grid
// (grid_x [W], grid_y [H]; the reference's default is 512 x 512, which its plotting helpers use).  One thread per pixel, the
// outlines' face frames staged in LDS 16 blocks at a time; the pixel test is the path's own, operation for operation
// (oracle/raster.py contains_2d): ((X - c.x) * n.x) + ((Y - c.z) * n.z) <= 0 for every face.  out [H, W] u8 (row 0 = top).
#define RENDER_TILE 16
__global__ __launch_bounds__(256) void k_render_blocks(const bridges_shape* shapes, int n, const double* __restrict__ verts,
                                                       const int32_t* __restrict__ shape_id, const double* __restrict__ gx, int W,
                                                       const double* __restrict__ gy, int H, uint8_t* __restrict__ out) {
    __shared__ double fr[RENDER_TILE][MAXV][4];
    __shared__ int nv_s[RENDER_TILE];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < (int64_t)W * H;
    const int r = live ? (int)(p / W) : 0, q = live ? (int)(p % W) : 0;
    const double X = gx[q], Y = gy[r];
    bool any = false;
    for (int b0 = 0; b0 < n; b0 += RENDER_TILE) {
        __syncthreads();
        const int nt = n - b0 < RENDER_TILE ? n - b0 : RENDER_TILE;
        for (int i = threadIdx.x; i < nt * MAXV; i += blockDim.x) {
            const int b = i / MAXV, f = i % MAXV;
            const bridges_shape& sh = shapes[shape_id[b0 + b]];
            if (f == 0) nv_s[b] = sh.nv;
            if (f < sh.nv) {
                const double* v = verts + (size_t)(b0 + b) * MAXV * 2;
                const Frame2 e = edge_frame(v[2 * sh.fa[f]], v[2 * sh.fa[f] + 1], v[2 * sh.fb[f]], v[2 * sh.fb[f] + 1]);
                fr[b][f][0] = e.cx; fr[b][f][1] = e.cz; fr[b][f][2] = e.nx; fr[b][f][3] = e.nz;
            }
        }
        __syncthreads();
        for (int b = 0; b < nt && !any; ++b) {
            bool inside = true;
            for (int f = 0; f < nv_s[b]; ++f) {
                const double d = ((X - fr[b][f][0]) * fr[b][f][2]) + ((Y - fr[b][f][1]) * fr[b][f][3]);
                inside = inside && d <= 0.0;
            }
            any = inside;
        }
    }
    if (live) out[p] = any ? 1 : 0;
}

// get_action_features + filter_actions + the linear reward for n posed candidate outlines against ONE state (the
// stand-alone form of what k_raster does inside the lock-step; robotoddler/training/successor_dqn.py:84-94, 397-401,
// robotoddler/utils/actions.py:71-82, gym_env.py:304-323): raster of every outline, mask[i] = all vertices inside
// [xlim, ylim] (and z >= 0) within 1e-6 AND no pixel shared with the state or the obstacle raster, lin[i] = sum(raster *
// reward_map) through the map's float64 row prefix sums.  One wave per candidate.
__global__ __launch_bounds__(256) void k_action_features(const bridges_shape* shapes, int n, const double* verts,
                                                         const int32_t* shape_id, const double* gx, const double* gy, int size,
                                                         double xlim0, double xlim1, double ylim0, double ylim1,
                                                         const uint64_t* state_bits, const uint64_t* obstacle_bits,
                                                         const double* reward_prefix, uint64_t* bits, float* img,
                                                         uint8_t* mask, float* lin) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    const uint64_t occ = (state_bits ? state_bits[lane] : 0ull) | (obstacle_bits ? obstacle_bits[lane] : 0ull);
    for (int it = wave; it < n; it += nwaves) {
        const bridges_shape& sh = shapes[shape_id[it]];
        const double* v = verts + (size_t)it * MAXV * 2;
        const uint64_t b = raster_outline(v, sh.nv, sh.fa, sh.fb, gx, gy, size, lane);
        bool out_of_bounds = false;
        if (lane < sh.nv) {
            const double eps = 1e-6, vx = v[2 * lane], vz = v[2 * lane + 1];
            out_of_bounds = vx < xlim0 - eps || vx > xlim1 + eps || vz < ylim0 - eps || vz > ylim1 + eps || vz < -eps;
        }
        const bool inb = __ballot(out_of_bounds) == 0ull;
        const bool overlap = __ballot((b & occ) != 0ull) != 0ull;
        double p_hi = 0.0, p_lo = 0.0;
        if (reward_prefix) raster_reward_fetch(b, reward_prefix, lane, p_hi, p_lo);
        if (bits) bits[(size_t)it * IMG + lane] = b;
        if (img) write_f32_image(img + (size_t)it * IMG * IMG, b, lane);
        const double l = raster_reward_sum(p_hi, p_lo);
        if (lane == 0) {
            if (mask) mask[it] = (uint8_t)(inb && !overlap);
            if (lin) lin[it] = (float)l;
        }
    }
}

__global__ void k_bits_or(int n_groups, const int32_t* ranges, const uint64_t* bits, uint64_t* out) {
    int g = blockIdx.x, lane = threadIdx.x;
    uint64_t acc = 0ull;
    for (int i = ranges[2 * g]; i < ranges[2 * g + 1]; ++i) acc |= bits[(size_t)i * IMG + lane];
    out[(size_t)g * IMG + lane] = acc;
}

__global__ __launch_bounds__(256) void k_bits_to_f32(int n, const uint64_t* bits, float* img) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    for (int it = wave; it < n; it += nwaves)
        write_f32_image(img + (size_t)it * IMG * IMG, bits[(size_t)it * IMG + lane], lane);
}

// K2+K3: is_stable_rbe on independent assemblies; fixed_mask bit b = block b is_static.
__global__ __launch_bounds__(WAVE) void k_stability(const bridges_shape* shapes, int n, int K, const double* pose_all,
                                                    const double* verts_all, const int32_t* shape_all,
                                                    const int32_t* n_blocks, const uint32_t* fixed_mask, double mu,
                                                    double density, double floor_hw, double floor_depth,
                                                    uint8_t* stable, double* info, double* lp_ws, int64_t ws_stride,
                                                    double tension_tol, double* forces) {
    __shared__ __attribute__((aligned(16))) double lds_tab[LP_TAB_LDS > (sizeof(FaceLds) / 8) ? LP_TAB_LDS : (sizeof(FaceLds) / 8)];
    __shared__ LpScratch S;
    FaceLds& F = *reinterpret_cast<FaceLds*>(lds_tab);
    double* tab = lds_tab;
    const int e = blockIdx.x, lane = threadIdx.x;
    const int nb = n_blocks[e];
    const double* pose = pose_all + (size_t)e * K * 4;
    const double* verts = verts_all + (size_t)e * K * MAXV * 2;
    const int32_t* shape_id = shape_all + (size_t)e * K;
    // workspace layout per assembly: [0,8*MAXIF) interface geometry, [8*MAXIF,9*MAXIF) interface bodies (int32
    // pairs), then the tableau overflow area
    double* ws = lp_ws + (size_t)e * ws_stride;
    double* if_geom = ws;
    int32_t* if_body = reinterpret_cast<int32_t*>(ws + 8 * MAXIF);
    double* tab_ws = ws + 9 * MAXIF;
    const int64_t tab_cap = ws_stride - 9 * MAXIF;
    const uint32_t fm = fixed_mask[e];
    const bool bad_mask = false;
    const long long t0 = clock64();
    stage_faces(F, 0, 1 + nb * MAXV, verts, shape_id, shapes, floor_hw, lane);
    __syncthreads();
    int n_if = 0;
    bool overflow = false;
    for (int b = 0; b < nb; ++b) {
        n_if = append_interfaces(F, b, shape_id, shapes, floor_depth, n_if, if_body, if_geom, lane, &overflow);
        __syncthreads();
    }
    bool err = false;
    double w = 0.0;
    int piv = 0;
    const long long t1 = clock64();
    bool too_big = false;
    AsmView A = env_view(nb, pose, shape_id, shapes, n_if, if_body, if_geom);
    if (tension_tol > 0.0) {
        // is_stable_rbe_penalty (stability.py:75-88): contact points may also PULL (one column per point along -n).  "Some
        // equilibrium has a total tension <= tol" is again a feasibility question: the tension columns enter the force
        // budget with the coefficient S_MAX / tol, so  sum x + (S_MAX / tol) sum t <= S_MAX  bounds the total tension by
        // tol (and the compressive total by what is left of S_MAX).
        A.n_tens = n_if;
        A.tens_coef = LP_S_MAX * density / tension_tol;
        if (6 * n_if > LP_MAX_COLS) { err = true; A.n_tens = 0; }
    }
    bool st = rbe_stable(tab, LP_TAB_LDS, LP_MAX_COLS, tab_ws, tab_cap, S, A, fm, mu, density, lane, &w, &piv, &err, &too_big);
    err = err || too_big;
    if (forces) {
        // basic solution of a feasible verdict (lp_verify left it in S.rowr): per contact point the compressive normal
        // force c_np = a + b of the two cone generators, the tension c_nn and the tangential force mu (a - b)
        __syncthreads();
        const bool have = st && !err && n_if > 0 && __popc(((nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u)) & ~fm)) > 0;
        for (int i = lane; i < MAXIF * 2; i += WAVE) {
            const int k = i >> 1, ip = i & 1;
            double cnp = 0.0, cnn = 0.0, ft = 0.0;
            if (have && k < n_if) {
                const double a = S.rowr[4 * k + 2 * ip], b = S.rowr[4 * k + 2 * ip + 1];
                cnp = a + b; ft = mu * (a - b);
                if (A.n_tens) cnn = S.rowr[4 * n_if + 2 * k + ip];
            }
            double* f = forces + ((size_t)e * MAXIF * 2 + i) * 3;
            f[0] = cnp; f[1] = cnn; f[2] = ft;
        }
    }
    if (lane == 0) {
        stable[e] = (uint8_t)(st && !bad_mask);
        const long long t2 = clock64();
        info[8 * e + 0] = w;
        info[8 * e + 1] = (double)n_if;
        info[8 * e + 2] = (double)piv;
        info[8 * e + 3] = bad_mask ? 2.0 : ((err || overflow) ? 1.0 : 0.0);
        info[8 * e + 4] = (double)(t1 - t0);     // shader cycles: face staging + interface detection
        info[8 * e + 5] = (double)(t2 - t1);     // shader cycles: tableau build + simplex
        info[8 * e + 6] = 0.0;
        info[8 * e + 7] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The rows a Q-network is fed: compact indices of the valid (filtered, actions.py:71-82) candidates of every env, env-major,
// with their owning env and the per-env row ranges.  torch.nonzero + gather did this in six launches and made the host
// wait twice (nonzero reads its count back); here the count travels to a host word behind the two launches.
//   k_valid_scan: seg[e] = sum of n_valid[0 .. e), seg[E] = total -> *h_total (host-visible), one workgroup
//   k_valid_fill: one wave per env, ballot compaction of its mask bytes in candidate order
// With rep != nullptr (k_env_match: rep[e] = the first env that holds exactly env e's state) only the envs that represent
// their group get rows; every env's row range seg_lo[e] .. seg_hi[e] is the range of its representative -- envs in the same
// state hold the same candidates in the same order, so they share the rows (and everything computed from them).
__global__ __launch_bounds__(1024) void k_valid_scan(int E, const int32_t* __restrict__ n_valid, const int32_t* __restrict__ rep,
                                                     int32_t* __restrict__ seg, int32_t* __restrict__ h_total) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) { carry_s = 0; seg[0] = 0; }
    __syncthreads();
    for (int base = 0; base < E; base += 1024) {
        const int i = base + t;
        int incl = (i < E && (rep == nullptr || rep[i] == i)) ? n_valid[i] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int wbase = carry_s;
        for (int k = 0; k < wv; ++k) wbase += wave_tot[k];
        if (i < E) seg[i + 1] = wbase + incl;
        __syncthreads();
        if (t == 1023) carry_s = wbase + incl;
        __syncthreads();
    }
    if (t == 0) *h_total = carry_s;
}

__global__ __launch_bounds__(256) void k_valid_fill(int E, const int32_t* __restrict__ cand_offset, const int32_t* __restrict__ n_cand,
                                                    const uint8_t* __restrict__ cand_mask, const int32_t* __restrict__ seg,
                                                    const int32_t* __restrict__ rep, int32_t* __restrict__ seg_lo,
                                                    int32_t* __restrict__ seg_hi, int64_t* __restrict__ idx,
                                                    int64_t* __restrict__ row_env) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const int r = rep ? rep[e] : e;
    if (lane == 0 && seg_lo) { seg_lo[e] = seg[r]; seg_hi[e] = seg[r + 1]; }
    if (r != e) return;                                                   // the representative's rows serve this env
    const int off = cand_offset[e], n = n_cand[e], end = seg[e + 1];
    int pos = seg[e];
    for (int base = 0; base < n; base += 64) {
        const int a = base + lane;
        const bool m = a < n && cand_mask[off + a] != 0;
        const unsigned long long b = __ballot(m);
        const int p = pos + __popcll(b & ((1ull << lane) - 1ull));
        if (m && p < end) {                                               // p < end: n_valid and the mask always agree; never write past the env's range
            idx[p] = off + a;
            row_env[p] = e;
        }
        pos += __popcll(b);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Environments in the same state.  Thousands of envs of one task pass through the same early states (every freshly reset
// env holds the empty assembly; a near-deterministic policy sends many of them down the same paths), and everything a
// Q-network is asked about a state -- its candidate rows and their values -- is a function of the state alone.
//   k_env_hash:  hkey[e] = 64-bit hash of (n_blocks, the live slots' shape / pose bits / occupancy, flag[e])
//   k_env_match: rep[e] = the smallest env index with the same hash whose state equals env e's WORD FOR WORD (else e itself:
//                a hash collision only costs the sharing, never correctness).  E^2 / 64 comparisons per wave: 4096 envs, ~10 us.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(256) void k_env_hash(int E, int K, const int32_t* __restrict__ n_blocks, const int32_t* __restrict__ blk_shape,
                                                  const double* __restrict__ blk_pose, const uint8_t* __restrict__ blk_occ,
                                                  const uint8_t* __restrict__ flag, uint64_t* __restrict__ hkey) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const int nb = n_blocks[e];
    uint64_t h = 0ull;
    if (lane < nb && lane < K) {
        const size_t s = (size_t)e * K + lane;
        const uint64_t* p = reinterpret_cast<const uint64_t*>(blk_pose + s * 4);
        uint64_t a = mix64((uint64_t)(uint32_t)blk_shape[s] | ((uint64_t)blk_occ[s] << 32));
        a = mix64(a ^ p[0]); a = mix64(a ^ p[1]); a = mix64(a ^ p[2]); a = mix64(a ^ p[3]);
        h = mix64(a + (uint64_t)(lane + 1) * 0x9E3779B97F4A7C15ull);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) h += __shfl_xor(h, m);
    if (lane == 0) hkey[e] = mix64(h + (uint64_t)nb * 0xD6E8FEB86659FD93ull + (flag ? (uint64_t)flag[e] : 0ull));
}

__global__ __launch_bounds__(256) void k_env_match(int E, int K, const int32_t* __restrict__ n_blocks, const int32_t* __restrict__ blk_shape,
                                                   const double* __restrict__ blk_pose, const uint8_t* __restrict__ blk_occ,
                                                   const uint8_t* __restrict__ flag, const uint64_t* __restrict__ hkey,
                                                   int32_t* __restrict__ rep) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const uint64_t mine = hkey[e];
    int r = e;
    for (int base = 0; base < e; base += 64) {                           // ascending: the first chunk with a hit holds the minimum
        const int j = base + lane;
        const unsigned long long hit = __ballot(j < e && hkey[j] == mine);
        if (hit) { r = base + (__ffsll((long long)hit) - 1); break; }
    }
    bool same = true;
    if (r != e) {
        const int nb = n_blocks[e];
        same = nb == n_blocks[r] && (flag == nullptr || flag[e] == flag[r]);
        if (same && lane < nb && lane < K) {
            const size_t a = (size_t)e * K + lane, b = (size_t)r * K + lane;
            const uint64_t* pa = reinterpret_cast<const uint64_t*>(blk_pose + a * 4);
            const uint64_t* pb = reinterpret_cast<const uint64_t*>(blk_pose + b * 4);
            same = blk_shape[a] == blk_shape[b] && blk_occ[a] == blk_occ[b] && pa[0] == pb[0] && pa[1] == pb[1] && pa[2] == pb[2] &&
                   pa[3] == pb[3];
        }
        same = __ballot(!same) == 0ull;
    }
    if (lane == 0) rep[e] = same ? r : e;
}

}  // namespace bridges

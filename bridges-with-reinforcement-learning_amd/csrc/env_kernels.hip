// Vectorised assembly_gym lock-step on gfx950: one 64-lane wavefront per environment for the
// task logic (k_step, k_enumerate, k_select) and a persistent wave-per-image rasteriser (k_raster).
//
// Reference semantics (paths relative to the reference root):
//   k_step      AssemblyGym.step + stabilities_freezing + sparse_reward + terminated
//               (assembly_gym/assembly_gym/envs/gym_env.py:11-22, 141-145, 218-253, 325-333),
//               lin_reward rule of rollout_episode (robotoddler/training/successor_dqn.py:397-401)
//   k_enumerate generate_actions (robotoddler/utils/actions.py:7-52) + create_block
//               (gym_env.py:204-216) + the bounds half of collision_on_action (gym_env.py:304-323)
//   k_raster    render_blocks_2d / Shape.contains_2d (assembly_gym/assembly_gym/utils/rendering.py:105-113,
//               assembly_env.py:126-137), the overlap half of filter_actions (actions.py:71-82) and
//               sum(action_features * reward_features) (successor_dqn.py:399-401)
#include "bridges_device.h"
#include "rbe_device.h"

namespace bridges {

// ---------------------------------------------------------------------------------------------
__device__ inline void reset_env(const DevCtx& c, int e, int lane) {
    c.b.state_bits[(size_t)e * IMG + lane] = 0ull;
    if (lane < c.K) c.b.blk_occ[(size_t)e * c.K + lane] = 0;
    if (lane == 0) {
        c.b.n_blocks[e] = 0;
        c.b.n_if[e] = 0;
        c.b.targets_left[e] = (c.n_targets >= 32) ? 0xffffffffu : ((1u << c.n_targets) - 1u);
        c.b.needs_reset[e] = 0;
    }
}

// number of raw candidates of a state: n_groups * (n_ground + n_free_faces * n_offsets).  NOT clamped to the env's capacity
// a_max: k_scan does that for every producer of n_cand (k_reset, k_step, bridges_replay_unpack, a host that loads states),
// raises bit 3 of the env's F_LP_ERROR flag and counts the truncation (stats: cand_overflow) -- a candidate set cut short
// is data loss the reference cannot have (actions.py:7-52 enumerates everything) and must not pass silently.
__device__ inline int count_candidates(const DevCtx& c, int e, int nb, int lane) {
    int nfree = 0;
    for (int i0 = 0; i0 < nb * MAXV; i0 += WAVE) {
        int i = i0 + lane;
        bool fr = false;
        if (i < nb * MAXV) {
            int b = i / MAXV, f = i % MAXV;
            int nv = c.tt->shapes[c.b.blk_shape[(size_t)e * c.K + b]].nv;
            fr = f < nv && !((c.b.blk_occ[(size_t)e * c.K + b] >> f) & 1);
        }
        nfree += __popcll(__ballot(fr));
    }
    return c.n_groups * (c.n_ground + nfree * c.n_offsets);
}

__global__ __launch_bounds__(WAVE) void k_reset(DevCtx c) {
    int e = blockIdx.x, lane = threadIdx.x;
    reset_env(c, e, lane);
    if (lane < 8) c.b.step_flags[(size_t)e * 8 + lane] = 0;
    int nc = count_candidates(c, e, 0, lane);
    if (lane == 0) {
        c.b.reward[e] = 0.f;
        c.b.lin_reward[e] = 0.f;
        c.b.n_reached[e] = 0;
        c.b.draw_counter[e] = 0;
        c.b.n_cand[e] = nc;
    }
}

// ---------------------------------------------------------------------------------------------
// One wave per environment.  The kernel is a chain of dependent steps, and beside another env group's rasteriser
// (which keeps the HBM write queues full) every dependent global-memory round trip costs microseconds, so the loads
// are organised in exactly two levels: (0) everything addressed by the env id alone -- counters, the placed blocks,
// the contact list, the shape table, the persisted tableau's header -- is requested up front and staged in LDS;
// (1) what the selected candidate and the header point to (candidate pose / vertices / raster, tableau cells).
// After that the wave works on LDS and registers only; results leave as stores nobody waits for.
#define STEP_IF_LDS 32      // contacts kept in LDS (the rest of a longer list is read in place from global memory)
#define STEP_TAB_LDS 4096   // doubles of LDS tableau in k_step (32 KiB: carriers included, 12 blocks with 17 contacts still fit;
                            // 4 workgroups per CU, which costs nothing: a 2048-env group needs two rounds at 5 per CU as well)
struct StepStage {          // dead once the contacts are found: shares its storage with the tableau
    bridges_shape shapes[8];
    double pose[MAXK * 4];
    double verts[MAXK * MAXV * 2];
    int32_t shape_id[MAXK];
    int32_t occ[MAXK];
};
struct StepKeep {           // what the LP reads of the assembly
    double if_geom[STEP_IF_LDS * 8];
    int32_t if_body[STEP_IF_LDS * 2];
    double cen[MAXK * 2];
    double vol[MAXK];
};
static_assert(sizeof(StepStage) <= STEP_TAB_LDS * sizeof(double), "staging area must fit into the tableau storage");

__global__ __launch_bounds__(WAVE) void k_step(DevCtx c) {
    __shared__ __attribute__((aligned(16))) double tab[STEP_TAB_LDS];
    __shared__ LpScratch S;
    __shared__ StepKeep Lk;
    StepStage& L = *reinterpret_cast<StepStage*>(tab);
    // latency-bound kernel that usually runs beside another env group's bandwidth-bound rasteriser: take the
    // instruction arbiter's priority so the dependent pivot chain is not stretched by the co-resident store waves
    __builtin_amdgcn_s_setprio(3);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int K = c.K;
    uint8_t* flags = c.b.step_flags + (size_t)e * 8;
    const long long ts0 = DIAG(c, 8) ? wall_clock64() : 0;      // debug bit3: per-env phase stamps (tools/kstep_phases.py)
    int32_t* shape_id_g = c.b.blk_shape + (size_t)e * K;
    double* pose_g = c.b.blk_pose + (size_t)e * K * 4;
    double* verts_g = c.b.blk_verts + (size_t)e * K * MAXV * 2;
    uint8_t* occ_g = c.b.blk_occ + (size_t)e * K;
    int32_t* if_body_g = c.b.if_body + (size_t)e * MAXIF * 2;
    double* if_geom_g = c.b.if_geom + (size_t)e * MAXIF * 8;
    double* ws = c.b.lp_ws + (size_t)e * c.b.lp_ws_stride;
    const WarmHdr* hdr_g = reinterpret_cast<const WarmHdr*>(ws);

    // ---- level 0: everything the env id addresses ----
    const uint8_t need_reset = c.b.needs_reset[e];
    const int a = c.b.sel_index[e];
    const int off = c.b.cand_offset[e];
    const int nb = c.b.n_blocks[e];                 // index of the new block
    const int n_if_old = c.b.n_if[e];
    uint32_t left = c.b.targets_left[e];
    const uint64_t sbits = c.b.state_bits[(size_t)e * IMG + lane];
    WarmPre W;
    W.magic = hdr_g->magic; W.n_blocks = hdr_g->n_blocks; W.n_if = hdr_g->n_if; W.stride = hdr_g->stride;
    W.half = hdr_g->half; W.m = hdr_g->m;
    W.basis_lane = hdr_g->basis[lane];
    {
        const double* src = reinterpret_cast<const double*>(c.tt->shapes);
        double* dst = reinterpret_cast<double*>(L.shapes);
        const int nd = c.n_shapes * (int)(sizeof(bridges_shape) / 8);
        for (int i = lane; i < nd; i += WAVE) dst[i] = src[i];
        for (int i = lane; i < nb * 4; i += WAVE) L.pose[i] = pose_g[i];
        for (int i = lane; i < nb * MAXV * 2; i += WAVE) L.verts[i] = verts_g[i];
        if (lane < K) { L.shape_id[lane] = shape_id_g[lane]; L.occ[lane] = occ_g[lane]; }
        const int n0 = n_if_old < STEP_IF_LDS ? n_if_old : STEP_IF_LDS;
        for (int i = lane; i < n0 * 8; i += WAVE) Lk.if_geom[i] = if_geom_g[i];
        for (int i = lane; i < n0 * 2; i += WAVE) Lk.if_body[i] = if_body_g[i];
    }

    if (need_reset) {                               // reset-only lock-step (previous state had no valid action)
        reset_env(c, e, lane);
        if (lane < 8) flags[lane] = 0;
        if (lane == 0) {
            c.b.reward[e] = 0.f;
            c.b.lin_reward[e] = 0.f;
            c.b.n_reached[e] = 0;
        }
        if (lane == 0) c.b.n_cand[e] = c.n_groups * c.n_ground;     // raw count: k_scan clamps to a_max and flags a truncation
        return;
    }

    // ---- level 1: the selected candidate, the cells of the persisted tableau ----
    const size_t ci = (size_t)off + a;
    const int dsc = lane < 4 ? c.b.cand_desc[ci * 4 + lane] : 0;
    const double cpose = lane < 4 ? c.b.cand_pose[ci * 4 + lane] : 0.0;
    const double cvert = lane < MAXV * 2 ? c.b.cand_verts[ci * MAXV * 2 + lane] : 0.0;
    const uint64_t cbits = c.b.cand_bits[ci * IMG + lane];
    const float base = c.b.cand_lin[ci];
    warm_prefetch(W, ws, nb, n_if_old, lane);
    const int tb = __builtin_amdgcn_readlane(dsc, 0), tf = __builtin_amdgcn_readlane(dsc, 1);
    const int sh = __builtin_amdgcn_readlane(dsc, 2), fc = __builtin_amdgcn_readlane(dsc, 3);

    // ---- append the block (gym_env.py:220-232): LDS copy for this step, global copy for the next ones ----
    if (lane < 4) { L.pose[nb * 4 + lane] = cpose; pose_g[nb * 4 + lane] = cpose; }
    if (lane < MAXV * 2) { L.verts[nb * MAXV * 2 + lane] = cvert; verts_g[nb * MAXV * 2 + lane] = cvert; }
    c.b.state_bits[(size_t)e * IMG + lane] = sbits | cbits;
    __syncthreads();                                // level-0 staging and the new block are in LDS
    if (lane == 0) {
        L.shape_id[nb] = sh;
        shape_id_g[nb] = sh;
        L.occ[nb] = (int32_t)(1u << fc);
        occ_g[nb] = (uint8_t)(1u << fc);
        if (tb >= 0) {
            L.occ[tb] |= (int32_t)(1u << tf);
            occ_g[tb] = (uint8_t)L.occ[tb];
        }
        c.b.n_blocks[e] = nb + 1;
    }
    __syncthreads();
    const bridges_shape* shapes = L.shapes;
    const bridges_shape& shn = shapes[sh];
    // per block: world centroid and volume for the LP (the staging area becomes the tableau), free faces for the
    // candidate count of the next state
    int nf = 0;
    if (lane <= nb) {
        const bridges_shape& sb = shapes[L.shape_id[lane]];
        double rgx, rgz;
        rot2(sb.gx, sb.gz, L.pose[4 * lane + 2], L.pose[4 * lane + 3], rgx, rgz);
        Lk.cen[2 * lane] = L.pose[4 * lane] + rgx;
        Lk.cen[2 * lane + 1] = L.pose[4 * lane + 1] + rgz;
        Lk.vol[lane] = sb.volume;
        nf = sb.nv - __popc((uint32_t)L.occ[lane] & ((1u << sb.nv) - 1u));
    }

    // ---- targets (gym_env.py:163-169, compas Box.contains_point tol 1e-6) ----
    double vx = 0.0, vz = 0.0;
    const int nvn = shn.nv;
    bool hasv = lane < nvn;
    if (hasv) { vx = L.verts[nb * MAXV * 2 + 2 * lane]; vz = L.verts[nb * MAXV * 2 + 2 * lane + 1]; }
    double x0 = wave_min_d(hasv ? vx : 1e300), x1 = wave_max_d(hasv ? vx : -1e300);
    double z0 = wave_min_d(hasv ? vz : 1e300), z1 = wave_max_d(hasv ? vz : -1e300);
    {
        double cx = (x0 + x1) * 0.5, cz = (z0 + z1) * 0.5, hx = (x1 - x0) * 0.5, hz = (z1 - z0) * 0.5;
        double hy = shn.depth * 0.5;
        // the reference removes from targets_remaining while iterating over it (gym_env.py:164-168): CPython's list
        // iterator then skips the element that moves into the freed slot, so the open target after a reached one is not
        // tested for this block.  Reproduced literally (oracle/env.py does the same).
        bool skip = false;
        for (int t = 0; t < c.n_targets; ++t) {
            if (!((left >> t) & 1u)) continue;
            if (skip) { skip = false; continue; }
            bool in = fabs(c.targets[t][0] - cx) < hx + 1e-6 && fabs(c.targets[t][1]) < hy + 1e-6 &&
                      fabs(c.targets[t][2] - cz) < hz + 1e-6;
            if (in) { left &= ~(1u << t); skip = true; }
        }
    }
    const int n_reached = c.n_targets - __popc(left);
    const long long ts1 = DIAG(c, 8) ? wall_clock64() : 0;

    // ---- contact interfaces of the new block (assembly_env.py:281-304): its faces against the floor and every
    //      older block's faces, frames computed on the fly from the LDS vertices (same pair order and arithmetic as
    //      append_interfaces / oracle/rbe.py face_pair_contact); hits go to the persistent list and its LDS copy ----
    int n_if = n_if_old;
    bool overflow = false;
    {
        const double* nverts = L.verts + nb * MAXV * 2;
        const int totalp = (1 + nb * MAXV) * MAXV;
        for (int p0 = 0; p0 < totalp; p0 += WAVE) {
            const int p = p0 + lane;
            bool hit = false;
            double g[8];
            int bodyA = -1;
            if (p < totalp) {
                const int qa = p / MAXV, fn = p % MAXV;
                if (fn < shn.nv) {
                    const int ia = shn.fa[fn], ib = shn.fb[fn];
                    const double aBx = nverts[2 * ia], aBz = nverts[2 * ia + 1], bBx = nverts[2 * ib], bBz = nverts[2 * ib + 1];
                    const Frame2 fB = edge_frame(aBx, aBz, bBx, bBz);
                    if (qa == 0) {
                        hit = face_pair_contact_v(-c.floor_hw, 0.0, c.floor_hw, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, aBx, aBz, bBx,
                                                  bBz, fB.cx, fB.cz, fB.nx, fB.nz, fmin(c.floor_depth, shn.depth), g);
                    } else {
                        bodyA = (qa - 1) / MAXV;
                        const int f = (qa - 1) % MAXV;
                        const bridges_shape& sa = shapes[L.shape_id[bodyA]];
                        if (f < sa.nv) {
                            const double* v = L.verts + bodyA * MAXV * 2;
                            const double ax = v[2 * sa.fa[f]], az = v[2 * sa.fa[f] + 1], bx = v[2 * sa.fb[f]], bz = v[2 * sa.fb[f] + 1];
                            const Frame2 fr = edge_frame(ax, az, bx, bz);
                            hit = face_pair_contact_v(ax, az, bx, bz, fr.cx, fr.cz, fr.tx, fr.tz, fr.nx, fr.nz, aBx, aBz, bBx, bBz,
                                                      fB.cx, fB.cz, fB.nx, fB.nz, fmin(sa.depth, shn.depth), g);
                        }
                    }
                }
            }
            const uint64_t bal = __ballot(hit);
            const int idx = n_if + __popcll(bal & ((1ull << lane) - 1ull));
            if (hit && idx < MAXIF) {
                if_body_g[2 * idx] = bodyA;
                if_body_g[2 * idx + 1] = nb;
#pragma unroll
                for (int k = 0; k < 8; ++k) if_geom_g[8 * idx + k] = g[k];
                if (idx < STEP_IF_LDS) {
                    Lk.if_body[2 * idx] = bodyA;
                    Lk.if_body[2 * idx + 1] = nb;
#pragma unroll
                    for (int k = 0; k < 8; ++k) Lk.if_geom[8 * idx + k] = g[k];
                }
            }
            n_if += __popcll(bal);
        }
        if (n_if > MAXIF) { overflow = true; n_if = MAXIF; }
    }
    __syncthreads();

    // ---- stability with the last block frozen / nothing frozen (gym_env.py:238-245, 325-333) ----
    bool err = false;
    bool st_frozen = true, st_free = true;
    const long long ts2 = DIAG(c, 8) ? wall_clock64() : 0;
    bool warm_used = false, resolved = false;
    int lp_diag = 0;
    if (!DIAG(c, 1)) {
        AsmView A;                                  // centroids / volumes / contacts from LDS; the block arrays are gone
        A.pose = nullptr; A.shape_id = nullptr; A.shapes = nullptr; A.n_blocks = nb + 1;
        A.cand_b = -1; A.cand_pose = nullptr; A.cand_shape = 0; A.cen = Lk.cen; A.vol = Lk.vol; A.n_tens = 0; A.tens_coef = 1.0;
        A.n_if = n_if; A.n_if0 = n_if < STEP_IF_LDS ? n_if : STEP_IF_LDS;
        A.if_body0 = Lk.if_body; A.if_geom0 = Lk.if_geom;
        A.if_body1 = if_body_g + 2 * STEP_IF_LDS; A.if_geom1 = if_geom_g + 8 * STEP_IF_LDS;
        rbe_both(tab, STEP_TAB_LDS, ws, c.b.lp_ws_stride, S, A, n_if_old, W, c.mu, c.density, lane, &st_frozen, &st_free, &err,
                 &warm_used, &lp_diag, c.b.lp_snap ? c.b.lp_snap + (size_t)e * c.b.lp_snap_stride : nullptr, &resolved);
    }

    const long long ts3 = DIAG(c, 8) ? wall_clock64() : 0;
    // ---- reward / termination (gym_env.py:11-22, 141-145) ----
    const bool all_reached = left == 0;
    const bool terminated = !st_frozen || all_reached;
    // gym_env.py:143; a state that fills its K block slots is truncated as well (without a step limit the reference
    // would go on; the arrays here cannot), so slot nb == K is never written
    const bool truncated = (c.max_steps > 0 && (nb + 1) >= c.max_steps) || (nb + 1) >= K;
    const bool done = terminated || truncated;
    float reward = !st_frozen ? -1.f : (all_reached ? (float)n_reached : (float)(-1 + n_reached));
    float lin = st_free ? base : (st_frozen ? base / 100.f : 0.f);          // successor_dqn.py:397-401

    if (lane == 0) {
        flags[F_VALID] = 1; flags[F_STABLE_FROZEN] = st_frozen; flags[F_STABLE_UNFROZEN] = st_free;
        flags[F_TERMINATED] = terminated; flags[F_TRUNCATED] = truncated; flags[F_DONE] = done;
        flags[F_NO_ACTIONS] = 0; flags[F_LP_ERROR] = (uint8_t)((err ? 1 : 0) | (overflow ? 2 : 0) | (resolved ? 4 : 0));   // bit 2: warm verdict re-solved cold
        c.b.reward[e] = reward;
        c.b.lin_reward[e] = lin;
        c.b.n_reached[e] = n_reached;
        c.b.n_if[e] = n_if;
        c.b.targets_left[e] = left;
    }
    int nb_after = nb + 1;
    if (done) {                                     // auto-reset (after the stores above: same lanes, program order)
        reset_env(c, e, lane);
        nb_after = 0;
    }
    int nfree = 0;
    for (int b = 0; b < nb_after; ++b) nfree += __builtin_amdgcn_readlane(nf, b);
    const int nc = c.n_groups * (c.n_ground + nfree * c.n_offsets);     // raw count: k_scan clamps to a_max and flags a truncation
    if (lane == 0) c.b.n_cand[e] = nc;
    if (DIAG(c, 8) && lane == 0) {               // 100 MHz wall clock: start, after append, after interfaces, after LPs, end
        const long long ts4 = wall_clock64();
        double* st = ws + c.b.lp_ws_stride - 8;     // the last 8 doubles of the env's workspace are never used otherwise
        st[0] = (double)ts0; st[1] = (double)(ts1 - ts0); st[2] = (double)(ts2 - ts1); st[3] = (double)(ts3 - ts2);
        st[4] = (double)(ts4 - ts3); st[5] = (double)(nb + 1); st[6] = (double)n_if + (warm_used ? 0.5 : 0.0) + 100.0 * (double)lp_diag; st[7] = (double)ts4;
    }
}

// ---------------------------------------------------------------------------------------------
// exclusive prefix sum of n_cand -> cand_offset[E+1], the clamp of every env's raw candidate count to the capacity a_max
// (a truncated env gets bit 3 of its F_LP_ERROR flag and is counted), plus the per-lock-step statistics (no atomics anywhere:
// 4096 adds on one word cost ~50 us per kernel on this chip).  Single workgroup of only 4 waves: beside another env
// group's rasteriser (28 of a CU's 32 wave slots taken) a 16-wave workgroup waited ~100 us for a CU to place it.
#define SCAN_THREADS 256
#define SCAN_WAVES (SCAN_THREADS / WAVE)
#define SCAN_BATCH 8        // envs per thread whose loads are in flight together (one round trip per 2048 envs)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(DevCtx c, int after_step) {
    __shared__ int wave_tot[SCAN_WAVES];
    __shared__ int carry_s;
    __shared__ unsigned long long red[SCAN_WAVES][8];
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // blocks, env-steps, reset-only, lp errors, if overflow, valid, warm re-solves, truncated candidate sets
    for (int base0 = 0; base0 < c.E; base0 += SCAN_THREADS * SCAN_BATCH) {
        // all loads of the batch first: beside a rasteriser every dependent round trip costs microseconds
        int v[SCAN_BATCH], nbk[SCAN_BATCH], nval[SCAN_BATCH];
        uint64_t fl8[SCAN_BATCH];
#pragma unroll
        for (int u = 0; u < SCAN_BATCH; ++u) {
            const int i = base0 + u * SCAN_THREADS + t;
            const bool in = i < c.E;
            v[u] = in ? c.b.n_cand[i] : 0;              // raw count of the producer
            nbk[u] = in ? c.b.n_blocks[i] : 0;
            nval[u] = (in && after_step) ? c.b.n_valid[i] : 0;
            fl8[u] = (in && after_step) ? *reinterpret_cast<const uint64_t*>(c.b.step_flags + (size_t)i * 8) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < SCAN_BATCH; ++u) {
            const int base = base0 + u * SCAN_THREADS;
            if (base >= c.E) break;
            const int i = base + t;
            if (i < c.E) {
                if (v[u] > c.a_max) {                        // more candidates than the env has room for: cut, flag, count
                    v[u] = c.a_max;
                    c.b.n_cand[i] = c.a_max;
                    c.b.step_flags[(size_t)i * 8 + F_LP_ERROR] |= 8;
                    acc[7] += 1;
                }
                acc[0] += (unsigned long long)nbk[u];
                if (after_step) {
                    const unsigned valid = (unsigned)(fl8[u] >> (8 * F_VALID)) & 0xffu;
                    const unsigned lperr = (unsigned)(fl8[u] >> (8 * F_LP_ERROR)) & 0xffu;
                    acc[1] += valid ? 1 : 0;
                    acc[2] += valid ? 0 : 1;
                    acc[3] += (lperr & 1) ? 1 : 0;
                    acc[4] += (lperr & 2) ? 1 : 0;
                    acc[5] += (unsigned long long)nval[u];             // valid candidates of the state just left
                    acc[6] += (lperr & 4) ? 1 : 0;                     // a warm 'unstable' was solved again from scratch
                }
            }
            int incl = v[u];                                 // inclusive scan inside the wave
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                int up = __shfl_up(incl, o, WAVE);
                if (lane >= o) incl += up;
            }
            if (lane == WAVE - 1) wave_tot[wv] = incl;
            __syncthreads();
            int wbase = 0;
            for (int k = 0; k < wv; ++k) wbase += wave_tot[k];
            const int carry = carry_s;
            if (i < c.E) c.b.cand_offset[i] = carry + wbase + incl - v[u];
            __syncthreads();
            if (t == SCAN_THREADS - 1) carry_s = carry + wbase + incl;
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned long long x = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
        if (lane == 0) red[wv][k] = x;
    }
    __syncthreads();
    if (t == 0) {
        unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < SCAN_WAVES; ++w)
            for (int k = 0; k < 8; ++k) tot[k] += red[w][k];
        c.b.cand_offset[c.E] = carry_s;
        if (c.h_total) *c.h_total = carry_s;
        c.b.stats[ST_SUM_CAND] += (uint64_t)carry_s;
        c.b.stats[ST_SUM_BLOCKS] += tot[0];
        c.b.stats[ST_ENV_STEPS] += tot[1];
        c.b.stats[ST_RESET_ONLY] += tot[2];
        c.b.stats[ST_LP_ERRORS] += tot[3];
        c.b.stats[ST_IF_OVERFLOW] += tot[4];
        c.b.stats[ST_SUM_VALID] += tot[5];
        c.b.stats[ST_WARM_RESOLVED] += tot[6];
        c.b.stats[ST_CAND_OVERFLOW] += tot[7];
        c.b.stats[ST_LOCKSTEPS] += 1;
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_enumerate(DevCtx c) {
    __shared__ uint8_t free_b[MAXK * MAXV], free_f[MAXK * MAXV];
    __shared__ bridges_shape sh_l[8];
    __shared__ double verts_l[MAXK * MAXV * 2];
    __shared__ int32_t shape_l[MAXK], occ_l[MAXK];
    __shared__ double xg_l[32], offs_l[8];
    __builtin_amdgcn_s_setprio(3);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int K = c.K;
    // everything the env id addresses is requested at once and staged in LDS (see k_step)
    if (lane < 32) xg_l[lane] = c.tt->x_ground[lane];
    if (lane < 8) offs_l[lane] = c.tt->offsets[lane];
    const int nb = c.b.n_blocks[e];
    const int ncand = c.b.n_cand[e];
    const size_t off = (size_t)c.b.cand_offset[e];
    {
        const double* src = reinterpret_cast<const double*>(c.tt->shapes);
        double* dst = reinterpret_cast<double*>(sh_l);
        const int nd = c.n_shapes * (int)(sizeof(bridges_shape) / 8);
        for (int i = lane; i < nd; i += WAVE) dst[i] = src[i];
        const double* vg = c.b.blk_verts + (size_t)e * K * MAXV * 2;
        for (int i = lane; i < nb * MAXV * 2; i += WAVE) verts_l[i] = vg[i];
        if (lane < K) { shape_l[lane] = c.b.blk_shape[(size_t)e * K + lane]; occ_l[lane] = c.b.blk_occ[(size_t)e * K + lane]; }
    }
    __syncthreads();
    const bridges_shape* shapes = sh_l;
    const int32_t* shape_id = shape_l;
    const double* verts = verts_l;
    // free (block, face) list in (block, face) order (actions.py:28-45)
    int nfree = 0;
    for (int i0 = 0; i0 < nb * MAXV; i0 += WAVE) {
        int i = i0 + lane;
        bool fr = false;
        int b = i / MAXV, f = i % MAXV;
        if (i < nb * MAXV) {
            int nv = shapes[shape_id[b]].nv;
            fr = f < nv && !((occ_l[b] >> f) & 1);
        }
        uint64_t bal = __ballot(fr);
        if (fr) {
            int idx = nfree + __popcll(bal & ((1ull << lane) - 1ull));
            free_b[idx] = (uint8_t)b;
            free_f[idx] = (uint8_t)f;
        }
        nfree += __popcll(bal);
    }
    __syncthreads();
    const int gsize = c.n_ground + nfree * c.n_offsets;
    for (int a = lane; a < ncand; a += WAVE) {
        int grp = a / gsize, slot = a % gsize;
        int sh = c.group_shape[grp], fc = c.group_face[grp];
        int tb = -1, tf = 0;
        double ox;
        Frame2 f1;
        if (slot < c.n_ground) {
            ox = xg_l[slot];
            f1.cx = 0.0; f1.cz = 0.0; f1.tx = 1.0; f1.tz = 0.0; f1.nx = 0.0; f1.nz = 1.0;   // assembly_env.py:339-340
        } else {
            int k = (slot - c.n_ground) / c.n_offsets;
            ox = offs_l[(slot - c.n_ground) % c.n_offsets];
            tb = free_b[k]; tf = free_f[k];
            const bridges_shape& st = shapes[shape_id[tb]];
            const double* v = verts + (size_t)tb * MAXV * 2;
            f1 = edge_frame(v[2 * st.fa[tf]], v[2 * st.fa[tf] + 1], v[2 * st.fb[tf]], v[2 * st.fb[tf] + 1]);
        }
        const bridges_shape& sn = shapes[sh];
        double px, pz, cs, sn_;
        align_place(f1, sn.fcx[fc], sn.fcz[fc], sn.fnx[fc], sn.fnz[fc], ox, 0.0, px, pz, cs, sn_);
        const size_t ci = off + a;
        bool inb = true;
        const double eps = 1e-6;
        double wx[MAXV], wz[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            wx[i] = 0.0; wz[i] = 0.0;
            if (i < sn.nv) {
                double rx, rz;
                rot2(sn.vx[i], sn.vz[i], cs, sn_, rx, rz);
                wx[i] = px + rx; wz[i] = pz + rz;
                if (wx[i] < c.xlim0 - eps || wx[i] > c.xlim1 + eps || wz[i] < c.ylim0 - eps || wz[i] > c.ylim1 + eps) inb = false;
                if (wz[i] < -eps) inb = false;
            }
            c.b.cand_verts[ci * MAXV * 2 + 2 * i] = wx[i];
            c.b.cand_verts[ci * MAXV * 2 + 2 * i + 1] = wz[i];
        }
        // world face frames (oracle: Block.frames) for the rasteriser
        {
#define PICK6(v, k) ((k) == 5 ? v[5] : (k) == 4 ? v[4] : (k) == 3 ? v[3] : (k) == 2 ? v[2] : (k) == 1 ? v[1] : v[0])
#pragma unroll
            for (int f = 0; f < MAXV; ++f) {
                double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
                if (f < sn.nv) {
                    const int ia = sn.fa[f], ib = sn.fb[f];
                    const double ax = PICK6(wx, ia), az = PICK6(wz, ia), bx = PICK6(wx, ib), bz = PICK6(wz, ib);
                    Frame2 fr = edge_frame(ax, az, bx, bz);
                    o0 = fr.cx; o1 = fr.cz; o2 = fr.nx; o3 = fr.nz;
                }
                double2* dst = reinterpret_cast<double2*>(c.b.cand_frames + (ci * MAXV + f) * 4);
                dst[0] = make_double2(o0, o1);
                dst[1] = make_double2(o2, o3);
            }
#undef PICK6
            c.b.cand_rows[ci * 2 + 0] = (sn.nv << 16) | ((inb ? 1 : 0) << 24);   // packed for the rasteriser (bits 0-15: unused)
            c.b.cand_rows[ci * 2 + 1] = e;
        }
        c.b.cand_pose[ci * 4 + 0] = px; c.b.cand_pose[ci * 4 + 1] = pz;
        c.b.cand_pose[ci * 4 + 2] = cs; c.b.cand_pose[ci * 4 + 3] = sn_;
        c.b.cand_desc[ci * 4 + 0] = tb; c.b.cand_desc[ci * 4 + 1] = tf;
        c.b.cand_desc[ci * 4 + 2] = sh; c.b.cand_desc[ci * 4 + 3] = fc;
        c.b.cand_ox[ci] = ox;
        c.b.cand_inb[ci] = inb;
        c.b.cand_env[ci] = e;
    }
}

// ---------------------------------------------------------------------------------------------
// f32 expansion of a 64x64 bit raster held one row per lane: 16 wave-wide float4 stores of 1 KiB.
// With nz != nullptr (sparse update) only the row groups that hold pixels now or held pixels the last time this slot
// was written are stored -- the rest of the slot is zero already -- and *nz is replaced by the new group mask.
__device__ __forceinline__ void write_f32_image(float* img, uint64_t rowbits, int lane, int32_t* nz = nullptr) {
    const int sub = lane >> 4, col4 = (lane & 15) * 4;
    const uint64_t nonzero_rows = __ballot(rowbits != 0ull);       // most of a raster is empty rows
    if (nz) {
        uint64_t t = nonzero_rows | (nonzero_rows >> 1);
        t = (t | (t >> 2)) & 0x1111111111111111ull;                // bit 4g = any of rows 4g..4g+3
        uint32_t now = 0u;
#pragma unroll
        for (int g = 0; g < IMG / 4; ++g) now |= (uint32_t)((t >> (4 * g)) & 1ull) << g;
        uint32_t need = (uint32_t)__builtin_amdgcn_readfirstlane(*nz) | now;
        if (lane == 0) *nz = (int32_t)now;
        while (need) {
            const int g = __builtin_ctz(need);
            need &= need - 1u;
            uint32_t nib = 0u;
            if ((now >> g) & 1u) {
                uint64_t m = shfl_u64(rowbits, 4 * g + sub);
                nib = (uint32_t)(m >> col4) & 0xFu;
            }
            float4 v;
            v.x = (nib & 1u) ? 1.f : 0.f;
            v.y = (nib & 2u) ? 1.f : 0.f;
            v.z = (nib & 4u) ? 1.f : 0.f;
            v.w = (nib & 8u) ? 1.f : 0.f;
            *reinterpret_cast<float4*>(img + (size_t)(4 * g + sub) * IMG + col4) = v;
        }
        return;
    }
#pragma unroll
    for (int r0 = 0; r0 < IMG; r0 += 4) {
        uint32_t nib = 0u;
        if ((nonzero_rows >> r0) & 0xFull) {
            uint64_t m = shfl_u64(rowbits, r0 + sub);
            nib = (uint32_t)(m >> col4) & 0xFu;
        }
        float4 v;
        v.x = (nib & 1u) ? 1.f : 0.f;
        v.y = (nib & 2u) ? 1.f : 0.f;
        v.z = (nib & 4u) ? 1.f : 0.f;
        v.w = (nib & 8u) ? 1.f : 0.f;
        *reinterpret_cast<float4*>(img + (size_t)(r0 + sub) * IMG + col4) = v;
    }
}

// Rasterise one convex outline from its world face frames (centre.xz, outward normal.xz per face) WITHOUT visiting
// pixels.  The reference's pixel test (assembly_env.py:126-137 through oracle/raster.py) is, per face,
//     d = ((X[x] - c.x) * n.x) + ((Y[r] - c.z) * n.z) <= 0
// with separately rounded binary64 operations.  With a[x] = fl(fl(X[x] - c.x) * n.x) and b[r] = fl(fl(Y[r] - c.z) * n.z)
// the rounded sum fl(a + b) is <= 0 exactly when the exact sum is (rounding is monotone and never turns a non-zero
// sum of two doubles into zero), i.e. when  a[x] <= -b[r]  -- one comparison, no addition.  X = np.linspace is strictly
// increasing, rounding is monotone, so a[x] is monotone in x (non-decreasing for n.x > 0, non-increasing for n.x < 0,
// constant +-0 for n.x = 0): the pixels of row r that pass face f form a prefix or a suffix of the row, whose length is
// found by a 7-probe binary search over a[.] with exactly the reference's comparison.  The row of the raster is the
// intersection of those runs.  Lane x holds a[x] (column role) and lane r holds -b[r] (row role); every lane searches
// for its own row, so all 64 rows of the image come out at once (row r in lane r) with ~50 wave instructions per face,
// independent of the block's size, instead of ~30 per face-row pair in a loop over the rows.  Bit for bit the same
// raster as the per-pixel test (tests/test_gpu_env_parity.py against oracle/raster.py).
// fr0..fr3: lane f < NF holds centre.x, centre.z, normal.x, normal.z of face f (all-zero frames pass every pixel).
// X / Y: this lane's column / row coordinate (lanes >= n_img repeat the last grid value); n_img = S <= 64.
__device__ __forceinline__ uint64_t low_mask64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

template <int NF, int F0 = 0>
__device__ __forceinline__ uint64_t raster_rows(double fr0, double fr1, double fr2, double fr3, int n_img, double X, double Y,
                                                int lane) {
    int alo[NF], ahi[NF], pos[NF];
    double t[NF];
    bool rev[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double cxf = readlane_d(fr0, F0 + f), czf = readlane_d(fr1, F0 + f), nxf = readlane_d(fr2, F0 + f), nzf = readlane_d(fr3, F0 + f);
        const double a = (X - cxf) * nxf;               // column role
        t[f] = -((Y - czf) * nzf);                      // row role: pixel (r, x) passes face f iff a[x] <= t[r]
        alo[f] = __double2loint(a);
        ahi[f] = __double2hiint(a);
        rev[f] = nxf < 0.0;                             // wave-uniform: a[.] non-increasing, the run is a suffix
        pos[f] = 0;
    }
    // probes at pos + s - 1 for s = 32 .. 1 (pos = the largest count <= 63 the run supports), then at pos itself (64)
#define RASTER_PROBE(S_, INC_)                                                                                          \
    _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                                    \
        const int j = pos[f] + (S_);                                                                                    \
        const int src = rev[f] ? 63 - j : j;                                                                            \
        const double av = __hiloint2double(__shfl(ahi[f], src, WAVE), __shfl(alo[f], src, WAVE));                       \
        if (av <= t[f]) pos[f] += (INC_);                                                                               \
    }
    RASTER_PROBE(31, 32) RASTER_PROBE(15, 16) RASTER_PROBE(7, 8) RASTER_PROBE(3, 4) RASTER_PROBE(1, 2) RASTER_PROBE(0, 1)
    RASTER_PROBE(0, 1)
#undef RASTER_PROBE
    uint64_t bits = low_mask64(n_img);                  // images narrower than the canvas: columns >= S stay empty
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const uint64_t run = low_mask64(pos[f]);
        bits &= rev[f] ? (pos[f] ? ~low_mask64(64 - pos[f]) : 0ull) : run;
    }
    return lane < n_img ? bits : 0ull;                  // rows >= S stay empty
}

// six faces as two batches of three (the searches of a batch run interleaved; one batch of six keeps 30 values per lane
// alive and costs the kernel two of its eight waves per SIMD)
__device__ __forceinline__ uint64_t raster_rows6(double fr0, double fr1, double fr2, double fr3, int n_img, double X, double Y, int lane) {
    return raster_rows<3, 0>(fr0, fr1, fr2, fr3, n_img, X, Y, lane) & raster_rows<3, 3>(fr0, fr1, fr2, fr3, n_img, X, Y, lane);
}

__device__ __forceinline__ uint64_t raster_rows_nv(double fr0, double fr1, double fr2, double fr3, int nv, int n_img, double X,
                                                   double Y, int lane) {
    return nv <= 4 ? raster_rows<4>(fr0, fr1, fr2, fr3, n_img, X, Y, lane) : raster_rows6(fr0, fr1, fr2, fr3, n_img, X, Y, lane);
}

// sum(raster * reward_map) from the row runs: the pixels of a row of a convex outline are one run [lo, hi), so the
// row's sum is prefix[r][hi] - prefix[r][lo] on the float64 row prefix sums of the reward map (reward_prefix, [64][65]).
// Two steps, so that the caller can put the image's 16 KiB of stores between the two gathers and their use (the loads
// are in flight under the stores instead of in front of them).
__device__ __forceinline__ void raster_reward_fetch(uint64_t rowbits, const double* prefix, int lane, double& p_hi, double& p_lo) {
    p_hi = 0.0; p_lo = 0.0;
    if (rowbits) {
        const int lo = __builtin_ctzll(rowbits), hi = 64 - __builtin_clzll(rowbits);
        const double* p = prefix + lane * (IMG + 1);
        p_hi = p[hi]; p_lo = p[lo];
    }
}
__device__ __forceinline__ double raster_reward_sum(double p_hi, double p_lo) { return wave_sum_d(p_hi - p_lo); }

// Same from world vertices (stand-alone operator): frames are derived first (oracle/raster.py contains_2d).
__device__ __forceinline__ uint64_t raster_outline(const double* v /*[6,2]*/, int nv, const int32_t* fa,
                                                   const int32_t* fb, const double* gx, const double* gy, int size, int lane) {
    double cx = 0.0, cz = 0.0, nx = 0.0, nz = 0.0;
    if (lane < nv) {
        Frame2 fr = edge_frame(v[2 * fa[lane]], v[2 * fa[lane] + 1], v[2 * fb[lane]], v[2 * fb[lane] + 1]);
        cx = fr.cx; cz = fr.cz; nx = fr.nx; nz = fr.nz;
    }
    const int g = lane < size ? lane : size - 1;        // lanes beyond the image repeat the last grid value (a[.] stays monotone)
    return raster_rows_nv(cx, cz, nx, nz, nv, size, gx[g], gy[g], lane);
}

// Rasteriser: work item i < total -> candidate i (compact index), else the state raster of env i - total.  One
// wave per image.  The grid is sized by the host from the previous lock-step's candidate count (kept in pinned
// memory) so that a wave sees about one item -- short-lived waves dispatched in item order keep the HBM write
// stream close to linear, which sustains ~14 % more bandwidth than a persistent grid (tools/store_bench.hip) -- and
// the grid-stride loop makes any count correct.  The row runs (raster_rows) cost ~250 wave instructions per image and
// hide behind its 16 KiB of stores.
// NF = 4 when every candidate shape of the task has at most 4 faces (64 VGPRs: 8 waves per SIMD), else MAXV.
template <int NF>
__global__ __launch_bounds__(256) void k_raster(DevCtx c) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nwaves = (gridDim.x * blockDim.x) / WAVE;
    const int total = c.b.cand_offset[c.E];
    const int items = total + (c.b.state_raster ? c.E : 0);
    const double X = c.tt->grid_x[lane], Y = c.tt->grid_y[lane];
    const uint64_t obst = c.b.obstacle_bits[lane];
    for (int itv = wave; itv < items; itv += nwaves) {
        const int it = __builtin_amdgcn_readfirstlane(itv);       // wave-uniform: metadata comes through scalar loads
        if (it < total) {
            const size_t ci = (size_t)it;
            const int meta = c.b.cand_rows[ci * 2];
            const int e = c.b.cand_rows[ci * 2 + 1];
            const int nv = (meta >> 16) & 0xff;
            const bool inb = (meta >> 24) & 1;
            (void)nv;
            // everything the descriptor addresses is requested before any of it is used
            double fr0 = 0.0, fr1 = 0.0, fr2 = 0.0, fr3 = 0.0;
            if (lane < MAXV) {
                const double2* p = reinterpret_cast<const double2*>(c.b.cand_frames + (ci * MAXV + lane) * 4);
                const double2 a = p[0], b = p[1];
                fr0 = a.x; fr1 = a.y; fr2 = b.x; fr3 = b.y;
            }
            const uint64_t occ = c.b.state_bits[(size_t)e * IMG + lane] | obst;
            uint64_t bits = 0ull;
            if (!DIAG(c, 2)) bits = (NF == 4 || nv <= 4) ? raster_rows<4>(fr0, fr1, fr2, fr3, c.img, X, Y, lane)
                                                         : raster_rows6(fr0, fr1, fr2, fr3, c.img, X, Y, lane);
            const bool overlap = __ballot((bits & occ) != 0ull) != 0ull;
            double p_hi, p_lo;
            raster_reward_fetch(bits, c.b.reward_prefix, lane, p_hi, p_lo);
            c.b.cand_bits[ci * IMG + lane] = bits;
            if (c.b.cand_raster && !DIAG(c, 4))
                write_f32_image(c.b.cand_raster + ci * IMG * IMG, bits, lane, c.b.cand_raster_nz ? c.b.cand_raster_nz + ci : nullptr);
            const double lin = raster_reward_sum(p_hi, p_lo);
            if (lane == 0) {
                c.b.cand_lin[ci] = (float)lin;
                c.b.cand_mask[ci] = (uint8_t)(inb && !overlap);
            }
        } else {
            const int e = it - total;
            write_f32_image(c.b.state_raster + (size_t)e * IMG * IMG, c.b.state_bits[(size_t)e * IMG + lane], lane,
                            c.b.state_raster_nz ? c.b.state_raster_nz + e : nullptr);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// draw == 0: n_valid + no-action detection (successor_dqn.py:409-411), part of every lock-step;
// draw == 1: the synthetic uniform-random policy over the valid candidates -> sel_index.
__global__ __launch_bounds__(WAVE) void k_select(DevCtx c, int draw) {
    const int e = blockIdx.x, lane = threadIdx.x;
    const int nc = c.b.n_cand[e];
    const size_t off = (size_t)c.b.cand_offset[e];
    int nvalid = 0;
    for (int a0 = 0; a0 < nc; a0 += WAVE) {
        int a = a0 + lane;
        nvalid += __popcll(__ballot(a < nc && c.b.cand_mask[off + a]));
    }
    int sel = 0;
    if (nvalid > 0 && draw) {
        uint64_t ctr = c.b.draw_counter[e];
        uint64_t r = splitmix64(splitmix64(((c.seed & 0xFFFFFFFFull) << 32) | (uint32_t)(c.env_id_base + e)) ^ ctr);
        int rank = (int)(r % (uint64_t)nvalid);
        int seen = 0;
        for (int a0 = 0; a0 < nc; a0 += WAVE) {
            int a = a0 + lane;
            uint64_t bal = __ballot(a < nc && c.b.cand_mask[off + a]);
            int cnt = __popcll(bal);
            if (rank < seen + cnt) {
                int want = rank - seen;            // want-th set bit of bal
                for (int k = 0; k < want; ++k) bal &= bal - 1;
                sel = a0 + (__ffsll((long long)bal) - 1);
                break;
            }
            seen += cnt;
        }
    }
    if (lane == 0) {
        if (draw) {
            c.b.sel_index[e] = sel;
            if (nvalid > 0) c.b.draw_counter[e] += 1;
        } else {
            c.b.n_valid[e] = nvalid;
            if (nvalid == 0) {
                c.b.needs_reset[e] = 1;
                c.b.step_flags[(size_t)e * 8 + F_NO_ACTIONS] = 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// is_action_stable_rbe (assembly_gym/assembly_gym/utils/stability.py:122-130) for every valid candidate of every env:
// the candidate block is appended to the assembly (free), the boundary conditions of the placed blocks stay (the last
// placed block is frozen, gym_env.py:238-240) and the RBE feasibility LP decides.  One wave per candidate.  Nothing is
// rebuilt: the env's persistent contact list (k_step) is read in place, only the face pairs between the candidate and
// the older bodies are tested (same pair order as append_interfaces, so the LP is column for column the one k_step
// would build after placing the candidate) and kept in LDS; the candidate's pose / vertices / frames come from the
// candidate arrays k_enumerate wrote.  Result: cand_stable[ci] = 1 stable, 0 unstable (or masked-out candidate),
// 2 = solver error / interface overflow (counts as unstable, stability.py:68 + gym_env.py:182).
//
// Two launches share the code.  QUEUE == false: grid over all raw candidates, a small LDS tableau (high occupancy);
// a candidate whose tableau does not fit is appended to cand_queue (one atomic per such candidate, they are rare).
// QUEUE == true: a small persistent grid drains that queue with the full-size LDS tableau and the env workspace
// lp_ws (one slot per workgroup) behind it; every wave leaves when the queue head passes the count.
#ifndef CS_NEW_IF
#define CS_NEW_IF 16
#endif
#ifndef CS_WAVES            // waves per SIMD the first pass is compiled for (register cap 512 / CS_WAVES); tools/cs_variants.sh
#define CS_WAVES 4
#endif
template <int TAB, int MAXCOLS, bool QUEUE>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(QUEUE ? 1 : CS_WAVES, QUEUE ? 2 : CS_WAVES))) void k_candidate_stability(DevCtx c) {
    __shared__ __attribute__((aligned(16))) double tab[TAB];
    __shared__ LpScratchT<MAXCOLS> S;
    __shared__ double new_geom[CS_NEW_IF * 8];
    __shared__ int32_t new_body[CS_NEW_IF * 2];
    const int lane = threadIdx.x;
    const int K = c.K;
    const bridges_shape* shapes = c.tt->shapes;
    const int total = c.b.cand_offset[c.E];
    int32_t* cnt = c.b.cand_counters;                 // [0] queue length, [1] queue head
    const int qlen = QUEUE ? cnt[0] : 0;
    for (int item = blockIdx.x;; item += gridDim.x) {
        int ci;
        if constexpr (QUEUE) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&cnt[1], 1);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i >= qlen) return;
            ci = c.b.cand_queue[i];
        } else {
            if (item >= total) return;
            ci = item;
            if (!c.b.cand_mask[ci] || DIAG(c, 16)) {        // debug bit4: every wave leaves here (cost of the empty grid)
                if (lane == 0) c.b.cand_stable[ci] = 0;
                continue;
            }
        }
        __syncthreads();                               // LDS lists of the previous item are dead
        const int e = c.b.cand_env[ci];
        const int nb = c.b.n_blocks[e];                // index the candidate would take
        const int32_t* shape_id = c.b.blk_shape + (size_t)e * K;
        const double* pose = c.b.blk_pose + (size_t)e * K * 4;
        const double* verts = c.b.blk_verts + (size_t)e * K * MAXV * 2;
        const int csh = c.b.cand_desc[(size_t)ci * 4 + 2];
        const bridges_shape& shn = shapes[csh];
        const double* cverts = c.b.cand_verts + (size_t)ci * MAXV * 2;
        const double* cframes = c.b.cand_frames + (size_t)ci * MAXV * 4;
        const int n_if0 = c.b.n_if[e];
        // ---- interfaces candidate <-> floor and blocks < nb ----
        const int totalp = (1 + nb * MAXV) * MAXV;
        int n_new = 0;
        for (int p0 = 0; p0 < totalp; p0 += WAVE) {
            const int p = p0 + lane;
            bool hit = false;
            double g[8];
            int bodyA = -1;
            if (p < totalp) {
                const int qa = p / MAXV, fn = p % MAXV;
                if (fn < shn.nv) {
                    const int ia = shn.fa[fn], ib = shn.fb[fn];
                    const double aBx = cverts[2 * ia], aBz = cverts[2 * ia + 1], bBx = cverts[2 * ib], bBz = cverts[2 * ib + 1];
                    const double cBx = cframes[4 * fn], cBz = cframes[4 * fn + 1], nBx = cframes[4 * fn + 2], nBz = cframes[4 * fn + 3];
                    if (qa == 0) {
                        hit = face_pair_contact_v(-c.floor_hw, 0.0, c.floor_hw, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, aBx, aBz, bBx,
                                                  bBz, cBx, cBz, nBx, nBz, fmin(c.floor_depth, shn.depth), g);
                    } else {
                        bodyA = (qa - 1) / MAXV;
                        const int f = (qa - 1) % MAXV;
                        const bridges_shape& sa = shapes[shape_id[bodyA]];
                        if (f < sa.nv) {
                            const double* v = verts + (size_t)bodyA * MAXV * 2;
                            const double ax = v[2 * sa.fa[f]], az = v[2 * sa.fa[f] + 1], bx = v[2 * sa.fb[f]], bz = v[2 * sa.fb[f] + 1];
                            const Frame2 fr = edge_frame(ax, az, bx, bz);
                            hit = face_pair_contact_v(ax, az, bx, bz, fr.cx, fr.cz, fr.tx, fr.tz, fr.nx, fr.nz, aBx, aBz, bBx, bBz,
                                                      cBx, cBz, nBx, nBz, fmin(sa.depth, shn.depth), g);
                        }
                    }
                }
            }
            const uint64_t bal = __ballot(hit);
            const int idx = n_new + __popcll(bal & ((1ull << lane) - 1ull));
            if (hit && idx < CS_NEW_IF) {
                new_body[2 * idx] = bodyA;
                new_body[2 * idx + 1] = nb;
#pragma unroll
                for (int k = 0; k < 8; ++k) new_geom[8 * idx + k] = g[k];
            }
            n_new += __popcll(bal);
        }
        __syncthreads();
        uint8_t res;
        if (n_new > CS_NEW_IF || n_if0 + n_new > MAXIF) {
            res = 2;                                   // contact list overflow (k_step would flag the same placement)
        } else {
            AsmView A;
            A.pose = pose; A.shape_id = shape_id; A.shapes = shapes; A.n_blocks = nb + 1;
            A.cand_b = nb; A.cand_pose = c.b.cand_pose + (size_t)ci * 4; A.cand_shape = csh; A.cen = nullptr; A.vol = nullptr; A.n_tens = 0; A.tens_coef = 1.0;
            A.n_if = n_if0 + n_new; A.n_if0 = n_if0;
            A.if_body0 = c.b.if_body + (size_t)e * MAXIF * 2; A.if_geom0 = c.b.if_geom + (size_t)e * MAXIF * 8;
            A.if_body1 = new_body; A.if_geom1 = new_geom;
            const uint32_t fixed = nb > 0 ? (1u << (nb - 1)) : 0u;
            bool err = false, too_big = false;
            double w = 0.0;
            int piv = 0;
            bool st = false, solved = false;
            // continue from the env's snapshot (k_step's "last block frozen" tableau of this very state): ~3 pivots
            // instead of ~2 per row from scratch
            if (c.b.lp_snap && nb > 0) {
                const double* snap = c.b.lp_snap + (size_t)e * c.b.lp_snap_stride;
                WarmPre W;
                warm_header(W, snap, lane);
                W.ok = warm_matches(W, nb, n_if0);
                W.n_pre = 0;
                if (W.ok) {
                    bool fits = false, werr = false;
                    st = rbe_candidate_warm(tab, TAB, MAXCOLS, S, A, n_if0, W, snap, c.mu, c.density, lane, &fits, &werr, &piv);
                    if (!fits) {
                        if constexpr (!QUEUE) {
                            if (lane == 0) c.b.cand_queue[atomicAdd(&cnt[0], 1)] = ci;
                            continue;                  // decided by the second launch
                        }
                    } else if (!werr) {
                        solved = true;
                    }
                    __syncthreads();
                }
            }
            if (!solved) {                             // no snapshot (first block, foreign state) or a failed check: from scratch
                double* ws = QUEUE ? c.b.cand_ws + (size_t)blockIdx.x * c.b.cand_ws_stride : nullptr;
                st = rbe_stable(tab, TAB, MAXCOLS, ws, QUEUE ? c.b.cand_ws_stride : (int64_t)0, S, A, fixed, c.mu, c.density,
                                lane, &w, &piv, &err, &too_big);
                if (too_big) {
                    if constexpr (!QUEUE) {
                        if (lane == 0) c.b.cand_queue[atomicAdd(&cnt[0], 1)] = ci;
                        continue;                      // decided by the second launch
                    }
                    err = true;
                }
            }
            res = err ? 2 : (st ? 1 : 0);
        }
        if (lane == 0) c.b.cand_stable[ci] = res;
    }
}

}  // namespace bridges

// Wave-level contact detection and rigid-block-equilibrium feasibility.
//
// Replaces AssemblyEnv._reset_cra_assembly -> assembly_interfaces_numpy(amin=0.001)
// (assembly_gym/assembly_gym/envs/assembly_env.py:281-304) and is_stable_rbe ->
// rbe_solve(mu, density, penalty=False) (assembly_gym/assembly_gym/utils/stability.py:49-71).
// One 64-lane wavefront (= one workgroup) owns one assembly: face frames are staged in LDS,
// face-pair hits are compacted with wave ballots, the phase-1 simplex tableau lives in LDS
// (global overflow for the rare large assemblies) and every pivot decision is a wave
// reduction, so the result is deterministic and identical on every GPU.
#pragma once
#include "bridges_device.h"

namespace bridges {

#ifdef LP_PROFILE   // diagnostic build only (tools/lp_microbench.py --profile): shader cycles per pivot phase
__device__ unsigned long long g_lp_prof[8];
#define LP_PROF_DECL long long lp_acc_[6] = {0, 0, 0, 0, 0, 0}
#define LP_STAMP(var) long long var = clock64()
#define LP_ACC(slot, a, b) lp_acc_[slot] += (b) - (a)
#define LP_PROF_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_lp_prof[k_], (unsigned long long)lp_acc_[k_]); } while (0)
#else
#define LP_PROF_DECL
#define LP_STAMP(var)
#define LP_ACC(slot, a, b)
#define LP_PROF_FLUSH
#endif

#define RBE_TOL_PARALLEL 1e-6
#define RBE_TOL_COPLANAR 1e-6
#define RBE_AMIN 0.001
#define RBE_FEAS_TOL 1e-5     // oracle: HiGHS optimum <= 1e-7; observed gap between the classes: 0 vs >= 3e-2
#define LP_EPS_COST 1e-9
#define LP_TAU 1e-5           // smallest admissible pivot element: above the ~1e-6 noise the float32 meshes put into the
                              // tableau (angles off by ~7e-8 x lever arms x mu), below every real coefficient (>= ~1e-3)
#define LP_VERIFY_TOL 1e-4    // L1 residual of the original rows accepted for a "feasible" verdict (rhs perturbation <= 2e-6)
#define LP_TIE 1e-9           // ratios within this (relative) band are ties
#define LP_STALL 40           // degenerate pivots before Bland's rule takes over
#define LP_MAX_PIVOTS 5000
#define LP_PERTURB 1e-8       // rhs perturbation unit (anti-stalling)
#define LP_MARGIN 1e3         // a continued (warm) tableau's "infeasible" optimum below LP_MARGIN x the threshold is solved again from
                              // scratch before it is reported (observed optima: <= 3.5e-7 feasible, >= 7.8e-4 infeasible; threshold 1e-5)
#define LP_MARGIN_LO 0.1      // ... and so is its "feasible" optimum above LP_MARGIN_LO x the threshold.  Phase 1 keeps artificials on one side
                              // of every row (the sign the right-hand side had when the row was activated), so on a system that misses an
                              // equilibrium by mesh noise the optimum depends on that sign pattern and on the rhs perturbations the rows were
                              // added with: a continued tableau ended below the threshold where the cold solve ends at 1.41e-5 (two-sided L1
                              // residual of the same rows: 2e-7; mixed trapezoid / hexagon bridge, mu = 2: tests/stress/stress_parity.py
                              // --task mixed --seed 99, lock-step 88, env 797); the check on the original rows accepts 1e-4
#define LP_S_MAX 1e4          // budget on the total contact force sum_j x_j (oracle/rbe.py S_MAX): equilibria that exist only
                              // through forces of 1e5..1e12 x the block weights along float32 mesh noise are not equilibria
#ifndef LP_TAB_LDS
#define LP_TAB_LDS 2048                       // doubles of LDS tableau per wave (16 KiB); larger tableaux live in global memory
#endif
#define MAXFACES (1 + MAXK * MAXV)            // floor + K blocks
#define LP_MAX_COLS (4 * MAXIF)
#define LP_MAX_CHUNKS ((LP_MAX_COLS + 2 + WAVE - 1) / WAVE)

struct FaceLds {            // staged face frames of one assembly
    double ax[MAXFACES], az[MAXFACES], bx[MAXFACES], bz[MAXFACES];
    double cx[MAXFACES], cz[MAXFACES], tx[MAXFACES], tz[MAXFACES], nx[MAXFACES], nz[MAXFACES];
};

// Stage frames of faces [q_begin, q_end) (q = 0 floor, q = 1 + b*MAXV + f otherwise).
__device__ inline void stage_faces(FaceLds& F, int q_begin, int q_end, const double* verts /*[K,6,2]*/,
                                   const int32_t* shape_id, const bridges_shape* shapes, double floor_hw, int lane) {
    for (int q = q_begin + lane; q < q_end; q += WAVE) {
        if (q == 0) {
            F.ax[0] = -floor_hw; F.az[0] = 0.0; F.bx[0] = floor_hw; F.bz[0] = 0.0;
            F.cx[0] = 0.0; F.cz[0] = 0.0; F.tx[0] = 1.0; F.tz[0] = 0.0; F.nx[0] = 0.0; F.nz[0] = 1.0;
        } else {
            int b = (q - 1) / MAXV, f = (q - 1) % MAXV;
            const bridges_shape& sh = shapes[shape_id[b]];
            if (f < sh.nv) {
                const double* v = verts + (size_t)b * MAXV * 2;
                double ax = v[2 * sh.fa[f]], az = v[2 * sh.fa[f] + 1];
                double bx = v[2 * sh.fb[f]], bz = v[2 * sh.fb[f] + 1];
                Frame2 fr = edge_frame(ax, az, bx, bz);
                F.ax[q] = ax; F.az[q] = az; F.bx[q] = bx; F.bz[q] = bz;
                F.cx[q] = fr.cx; F.cz[q] = fr.cz; F.tx[q] = fr.tx; F.tz[q] = fr.tz; F.nx[q] = fr.nx; F.nz[q] = fr.nz;
            }
        }
    }
}

// oracle/rbe.py face_pair_contact, same operation order.  Face A: end points a, b, centre c, tangent t, normal n;
// face B: end points, centre, normal.
__device__ __forceinline__ bool face_pair_contact_v(double aAx, double aAz, double bAx, double bAz, double cAx, double cAz,
                                                    double tAx, double tAz, double nAx, double nAz, double aBx, double aBz,
                                                    double bBx, double bBz, double cBx, double cBz, double nBx, double nBz,
                                                    double depth, double* out8) {
    double dotn = nAx * nBx + nAz * nBz;
    if (dotn > -1.0 + RBE_TOL_PARALLEL) return false;
    double gap = (cBx - cAx) * nAx + (cBz - cAz) * nAz;
    if (fabs(gap) > RBE_TOL_COPLANAR) return false;
    double a0 = (aAx - cAx) * tAx + (aAz - cAz) * tAz;
    double a1 = (bAx - cAx) * tAx + (bAz - cAz) * tAz;
    double b0 = (aBx - cAx) * tAx + (aBz - cAz) * tAz;
    double b1 = (bBx - cAx) * tAx + (bBz - cAz) * tAz;
    double lo = fmax(fmin(a0, a1), fmin(b0, b1));
    double hi = fmin(fmax(a0, a1), fmax(b0, b1));
    if ((hi - lo) * depth < RBE_AMIN) return false;
    out8[0] = cAx + lo * tAx; out8[1] = cAz + lo * tAz;
    out8[2] = cAx + hi * tAx; out8[3] = cAz + hi * tAz;
    out8[4] = nAx; out8[5] = nAz; out8[6] = tAx; out8[7] = tAz;
    return true;
}

__device__ __forceinline__ bool face_pair_contact(const FaceLds& F, int qa, int qb, double depth, double* out8) {
    return face_pair_contact_v(F.ax[qa], F.az[qa], F.bx[qa], F.bz[qa], F.cx[qa], F.cz[qa], F.tx[qa], F.tz[qa], F.nx[qa],
                               F.nz[qa], F.ax[qb], F.az[qb], F.bx[qb], F.bz[qb], F.cx[qb], F.cz[qb], F.nx[qb], F.nz[qb],
                               depth, out8);
}

// Append the interfaces between block `nb_new` and every earlier body (floor, blocks < nb_new).
// Faces must be staged for q in [0, 1 + (nb_new+1)*MAXV).  Returns the new interface count
// (uniform); sets *overflow if MAXIF was exceeded (extra contacts are dropped).
__device__ inline int append_interfaces(const FaceLds& F, int nb_new, const int32_t* shape_id,
                                        const bridges_shape* shapes, double floor_depth, int n_if,
                                        int32_t* if_body, double* if_geom, int lane, bool* overflow) {
    const bridges_shape& shn = shapes[shape_id[nb_new]];
    int n_old_q = 1 + nb_new * MAXV;
    int total = n_old_q * MAXV;
    for (int p0 = 0; p0 < total; p0 += WAVE) {
        int p = p0 + lane;
        bool hit = false;
        double g[8];
        int bodyA = -1;
        if (p < total) {
            int qa = p / MAXV, fn = p % MAXV;
            bool valid = fn < shn.nv;
            double depthA = floor_depth;
            if (qa > 0) {
                bodyA = (qa - 1) / MAXV;
                const bridges_shape& sa = shapes[shape_id[bodyA]];
                valid = valid && ((qa - 1) % MAXV) < sa.nv;
                depthA = sa.depth;
            }
            if (valid) {
                int qb = 1 + nb_new * MAXV + fn;
                hit = face_pair_contact(F, qa, qb, fmin(depthA, shn.depth), g);
            }
        }
        uint64_t bal = __ballot(hit);
        int idx = n_if + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit) {
            if (idx < MAXIF) {
                if_body[2 * idx] = bodyA;
                if_body[2 * idx + 1] = nb_new;
#pragma unroll
                for (int k = 0; k < 8; ++k) if_geom[8 * idx + k] = g[k];
            }
        }
        n_if += __popcll(bal);
    }
    if (n_if > MAXIF) { *overflow = true; n_if = MAXIF; }
    return n_if;
}

// What the LP reads of one assembly: the env's block arrays plus (optionally) one extra block that is not part of the
// state (a candidate placement, index cand_b), and the contact list as the env's persistent list followed by an
// appended part (the candidate's interfaces, held in LDS by the wave that found them).
struct AsmView {
    const double* pose;            // [K,4] env blocks
    const int32_t* shape_id;       // [K]
    const bridges_shape* shapes;
    int n_blocks;                  // incl. the extra block
    int cand_b;                    // index of the extra block, or -1
    const double* cand_pose;       // [4]
    int cand_shape;
    int n_if, n_if0;               // interfaces in total / in the first list
    const int32_t* if_body0;       // [n_if0,2]
    const double* if_geom0;        // [n_if0,8]
    const int32_t* if_body1;       // [n_if-n_if0,2]
    const double* if_geom1;
    const double* cen;             // optional [K,2]: world centroid of every block (then pose / shapes are not read)
    const double* vol;             // optional [K]
    int n_tens;                    // 0, or n_if: one tension column per contact point behind the generators (penalty
                                   // variant, stability.py:75-88): a force along -n, budget coefficient tens_coef
    double tens_coef;
    __device__ __forceinline__ const double* P(int b) const { return b == cand_b ? cand_pose : pose + 4 * b; }
    __device__ __forceinline__ const bridges_shape& S(int b) const { return shapes[b == cand_b ? cand_shape : shape_id[b]]; }
    __device__ __forceinline__ const int32_t* ib(int k) const { return k < n_if0 ? if_body0 + 2 * k : if_body1 + 2 * (k - n_if0); }
    __device__ __forceinline__ const double* ig(int k) const { return k < n_if0 ? if_geom0 + 8 * k : if_geom1 + 8 * (k - n_if0); }
    // world centroid of block b: pose position + rotated local centroid (the moment reference of its equilibrium rows)
    __device__ __forceinline__ void centroid(int b, double& gcx, double& gcz) const {
        if (cen) { gcx = cen[2 * b]; gcz = cen[2 * b + 1]; return; }
        const bridges_shape& sh = S(b);
        const double* Pb = P(b);
        double rgx, rgz;
        rot2(sh.gx, sh.gz, Pb[2], Pb[3], rgx, rgz);
        gcx = Pb[0] + rgx; gcz = Pb[1] + rgz;
    }
    __device__ __forceinline__ double volume(int b) const { return vol ? vol[b] : S(b).volume; }
};

__device__ __forceinline__ AsmView env_view(int n_blocks, const double* pose, const int32_t* shape_id, const bridges_shape* shapes,
                                            int n_if, const int32_t* if_body, const double* if_geom) {
    AsmView A;
    A.pose = pose; A.shape_id = shape_id; A.shapes = shapes; A.n_blocks = n_blocks;
    A.cand_b = -1; A.cand_pose = pose; A.cand_shape = 0; A.cen = nullptr; A.vol = nullptr; A.n_tens = 0; A.tens_coef = 1.0;
    A.n_if = n_if; A.n_if0 = n_if; A.if_body0 = if_body; A.if_geom0 = if_geom; A.if_body1 = if_body; A.if_geom1 = if_geom;
    return A;
}

// Build the phase-1 tableau.  Rows 3*b..3*b+2 = (Fx, Fz, My) of free block b (< n_free), row m = the force budget
// sum_j x_j + s = LP_S_MAX, row m+1 = cost.  Columns [0, n) = cone generators, column n = the budget slack s,
// column n+1 = right-hand side.
// Blocks >= n_free are fixed (only the last block is ever frozen, gym_env.py:235-240).
// Every absolute tolerance of the solve (budget, rhs perturbation, feasibility and verification thresholds) is a
// force and is therefore scaled by `density` (the right-hand side is linear in it, the matrix does not depend on it);
// the pivot floor LP_TAU and the reduced-cost threshold are matrix quantities and stay.  At density = 1 the products
// are exact, i.e. the arithmetic is the one the tolerances were calibrated with.
// The right-hand side carries a tiny deterministic perturbation (<= 2e-7 per row, far below RBE_FEAS_TOL): these
// equilibrium systems are massively degenerate (most rhs entries are exactly 0) and the perturbation is what keeps
// the simplex from stalling.
// ncarr = m adds one "carrier" column per equilibrium row behind the rhs (column nn + 1 + i = e_i at the start, i.e. the
// explicit artificial of row i).  Row operations keep  tableau = R * [M | slack | w | I], so carrier i always holds
// R e_i: what a column that arrives later (a new contact of a later block) must be multiplied with to enter the tableau
// in its current coordinates (lp_warm_prepare).  Carriers are never priced.
template <typename TP>
__device__ inline void lp_build(TP T, int stride, int m, int m_act, int n, const AsmView& A, const int* row_of /*LDS [K]*/,
                                double mu, double density, int lane, int ncarr = 0) {
    const int nn = n + 1;                              // structural columns incl. the budget slack; rhs column index
    int cells = (m + 2) * stride;
    for (int i = lane; i < cells; i += WAVE) T[i] = 0.0;
    __syncthreads();
    const int n4 = 4 * A.n_if;                         // cone generators; columns [n4, n) are tension columns
    for (int j = lane; j < n; j += WAVE) {
        const bool tens = j >= n4;
        int k = tens ? (j - n4) >> 1 : j >> 2, ip = tens ? (j - n4) & 1 : (j >> 1) & 1, ig = j & 1;
        const double* g = A.ig(k);
        const int32_t* bd = A.ib(k);
        double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
        double nx = g[4], nz = g[5], tx = g[6], tz = g[7];
        double gx = tens ? -nx : (ig ? nx - mu * tx : nx + mu * tx);
        double gz = tens ? -nz : (ig ? nz - mu * tz : nz + mu * tz);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            int body = bd[side == 0 ? 1 : 0];                  // side 0: body B (+), side 1: body A (-)
            const int r = body >= 0 ? row_of[body] : -1;
            if (r >= 0) {
                double sgx = side == 0 ? gx : -gx;
                double sgz = side == 0 ? gz : -gz;
                double gcx, gcz;
                A.centroid(body, gcx, gcz);
                double rx = px - gcx, rz = pz - gcz;
                T[(r + 0) * stride + j] = sgx;
                T[(r + 1) * stride + j] = sgz;
                T[(r + 2) * stride + j] = rx * sgz - rz * sgx;
            }
        }
    }
    for (int i = lane; i < m; i += WAVE)
        T[i * stride + nn] = (LP_PERTURB * density) * (1.0 + 0.37 * (double)(i % 7) + 0.0618 * (double)(i % 11));
    for (int q = lane; q <= nn; q += WAVE) T[m * stride + q] = q < nn ? ((q >= n4 && q < n) ? A.tens_coef : 1.0) : LP_S_MAX * density;
    for (int i = lane; i < ncarr; i += WAVE) T[i * stride + nn + 1 + i] = 1.0;
    __syncthreads();
    for (int b = lane; b < A.n_blocks; b += WAVE)
        if (row_of[b] >= 0) T[(row_of[b] + 1) * stride + nn] += density * A.volume(b);
    __syncthreads();
    for (int q = lane; q <= nn; q += WAVE) {         // phase-1 cost row over the ACTIVE rows (rows >= m_act are passive)
        double s = 0.0;
        for (int i = 0; i < m_act; ++i) s += T[i * stride + q];
        T[(m + 1) * stride + q] = -s;
    }
    __syncthreads();
}

template <int MAXCOLS>
struct LpScratchT {                // LDS scratch of one wave's simplex (MAXCOLS = generator columns it can hold)
    double col[WAVE];              // entering column (row i in slot i, cost entry in slot m)
    double rowr[MAXCOLS + 4 + 3 * MAXK];   // normalised pivot row (generators, slack, rhs, carriers)
    int basis[WAVE];
    int row_of[MAXK];              // first tableau row of block b, or -1 if the block is fixed (is_static)
    short rows_nz[WAVE];           // rows touched by the current pivot (entering column entry != 0)
    short cols_nz[MAXCOLS + 4 + 3 * MAXK]; // columns touched by the current pivot (pivot row entry != 0)
};
typedef LpScratchT<LP_MAX_COLS> LpScratch;

// Ordering point between the lanes of the ONE wave that owns a tableau.  LDS operations of a wave execute in
// program order, so for an LDS tableau a compiler fence is enough; the global overflow path needs the real
// barrier (s_waitcnt vmcnt(0)) before other lanes re-read what was stored.
template <bool IN_LDS>
__device__ __forceinline__ void wave_sync() {
    if constexpr (IN_LDS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <typename TP>
__device__ __forceinline__ double artificial_sum(TP T, int stride, int m, int n, const int* basis, int lane) {
    double v = 0.0;
    if (lane < m && basis[lane] >= n) {
        double rhs = T[lane * stride + n];
        v = rhs > 0.0 ? rhs : 0.0;
    }
    return wave_sum_d(v);
}

// 1/x to ~1e-16 relative: hardware v_rcp_f64 seed + one Newton step (the LP needs no correctly rounded quotient;
// a full IEEE f64 division is ~40 dependent instructions and sat on the critical path of every pivot).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return r + r * (1.0 - x * r);
}

// Phase-1 simplex on the dense tableau (artificial columns implicit: an artificial that leaves never returns).
// Pricing: most negative reduced cost (ties -> lowest lane).  Ratio test: rows with a pivot candidate > LP_TAU,
// minimum ratio; among ratios tied within LP_TIE the LARGEST pivot element wins (ties -> lowest row) -- the
// float32 block meshes make faces parallel only to ~1e-7, which creates legitimate 1e-8 tableau entries that must
// never be pivoted on.  After LP_STALL pivots without progress the rules switch to Bland's (lowest entering column,
// lowest leaving variable) until the objective moves again.  Every decision is a wave reduction with a total
// order, so the pivot sequence -- and the boolean -- is identical on every GPU.
// A pivot is one dependent chain (price -> ratio -> stage -> sweep), so its latency is what bounds the kernel:
// reductions run on DPP, the cells of the elimination sweep are batched 4 per lane, and the running objective is
// read from the cost row (the exact artificial sum is recomputed only to confirm a "feasible" exit).
// `feas` = the feasibility threshold (RBE_FEAS_TOL x density).
// Returns w = sum of the artificial basics (<= feas <=> stable).  All lanes return the same value.
template <bool IN_LDS, typename TP, typename SC>
__device__ inline double lp_phase1(TP T, int stride, int m, int m_act, int n_gen, SC& S, int lane, int* pivots_out,
                                   bool* error, bool init_basis, double feas, int ncarr = 0) {
    // m equilibrium rows are stored and swept; only ACTIVE rows take part in the ratio test and carry artificials
    // (the others are "passive": equality rows that are transformed along but not enforced, see rbe_both).  A row is
    // passive iff its basis entry is -1: with init_basis that is rows >= m_act, otherwise whatever the caller set up
    // (the candidate LPs keep the frozen block's rows passive in the middle of the tableau).
    // Row m is the force-budget row (always enforced, basic variable = its slack, column n_gen), row m+1 the cost.
    int* basis = S.basis;
    const int n = n_gen + 1;                           // structural columns incl. the slack; also the rhs column index
    const int mb = m, mc = m + 1;                      // budget row, cost row
    if (init_basis) {
        for (int i = lane; i <= mb; i += WAVE) basis[i] = (i < m_act) ? n + i : (i == mb ? n_gen : -1);
        wave_sync<IN_LDS>();
    }
    int pivots = *pivots_out, stall = 0;
    bool bland = false;
    const int nchunk = (n + WAVE - 1) / WAVE;
    const int ncols = n + 1 + ncarr;                   // swept columns: structural, rhs, carriers
    const double progress = 1e-7 * feas;               // 1e-12 at density 1
    LP_PROF_DECL;
    (void)m_act;                                       // from here on activity is read off the basis
    double w = artificial_sum(T, stride, m, n, basis, lane);
    for (;;) {
        if (w <= feas) {                                           // confirm with the exact artificial sum
            w = artificial_sum(T, stride, m, n, basis, lane);
            if (w <= feas) break;
        }
        LP_STAMP(t_a);
        // ---- entering column ----
        int jin = -1;
        if (bland) {
            for (int c = 0; c < nchunk && jin < 0; ++c) {
                int j = c * WAVE + lane;
                bool neg = (j < n) && (T[mc * stride + j] < -LP_EPS_COST);
                uint64_t bal = __ballot(neg);
                if (bal) jin = c * WAVE + (__ffsll((long long)bal) - 1);
            }
        } else {
            double dbest = 0.0;
            int jbest = 0;
            for (int j = lane; j < n; j += WAVE) {
                double d = T[mc * stride + j];
                if (d < dbest) { dbest = d; jbest = j; }           // strict: keeps the lane's lowest column of a tie
            }
            double dmin = wave_min_d(dbest);
            if (dmin < -LP_EPS_COST) {
                int src = __ffsll((long long)__ballot(dbest == dmin)) - 1;
                jin = __builtin_amdgcn_readlane(jbest, src);
            }
        }
        if (jin < 0) {                                             // optimal: w is the true minimum
            w = artificial_sum(T, stride, m, n, basis, lane);
            break;
        }
        LP_STAMP(t_b);
        LP_ACC(0, t_a, t_b);
        // ---- ratio test, lanes over rows (m + 2 <= 50 < 64) ----
        double col = (lane <= mc) ? T[lane * stride + jin] : 0.0;  // lane m: budget row, lane m+1: cost entry
        double ratio = 1e300;
        if (((lane < m && basis[lane] >= 0) || lane == mb) && col > LP_TAU) {
            double rhs = T[lane * stride + n];
            ratio = (rhs > 0.0 ? rhs : 0.0) * fast_rcp(col);
        }
        const double rmin = wave_min_d(ratio);
        if (rmin >= 1e300) {                                       // no usable pivot in this column: retire it
            if (lane == 0) T[mc * stride + jin] = 0.0;
            wave_sync<IN_LDS>();
            continue;
        }
        const bool tie = ratio <= rmin + LP_TIE * (1.0 + rmin);
        const uint64_t tbal = __ballot(tie);
        int r;
        if ((tbal & (tbal - 1ull)) == 0ull) {                      // a single candidate row (the common case): no tie to break
            r = __ffsll((long long)tbal) - 1;
        } else if (bland) {
            int var = tie ? basis[lane] : 0x7fffffff;
            int vmin = wave_min_i(var);
            r = __ffsll((long long)__ballot(tie && var == vmin)) - 1;
        } else {
            double cmax = wave_max_d(tie ? col : -1e300);
            r = __ffsll((long long)__ballot(tie && col == cmax)) - 1;
        }
        const double ipiv = fast_rcp(readlane_d(col, r));
        LP_STAMP(t_c);
        LP_ACC(1, t_b, t_c);
        // ---- stage the entering column, the normalised pivot row and the lists of rows / columns the rank-1
        //      update actually touches (equilibrium tableaux are sparse: typically a fraction of the cells) ----
        S.col[lane] = col;
        const uint64_t rbal = __ballot(lane <= mc && (col != 0.0 || lane == r));
        if ((rbal >> lane) & 1ull) S.rows_nz[__popcll(rbal & ((1ull << lane) - 1ull))] = (short)lane;
        const int nr = __popcll(rbal);
        int nc = 0;
        for (int c = 0; c * WAVE < ncols; ++c) {
            const int q = c * WAVE + lane;
            double v = 0.0;
            if (q < ncols) {
                v = (q == jin) ? 1.0 : T[r * stride + q] * ipiv;
                S.rowr[q] = v;
            }
            const uint64_t cbal = __ballot(q < ncols && v != 0.0);
            if ((cbal >> lane) & 1ull) S.cols_nz[nc + __popcll(cbal & ((1ull << lane) - 1ull))] = (short)q;
            nc += __popcll(cbal);
        }
        wave_sync<IN_LDS>();
        LP_STAMP(t_d);
        LP_ACC(2, t_c, t_d);
        // ---- elimination over the touched cells only, 4 independent cells per lane per trip (loads first, then
        //      stores: the cells are distinct, which the compiler cannot prove, so the batching is explicit) ----
        {
            const int cells = nr * nc;
            const int di = WAVE / nc, dq = WAVE - di * nc;
            int a = lane / nc, b = lane - a * nc;
            for (int idx = lane; idx < cells; idx += 4 * WAVE) {
                int ii[4], qq[4];
                double tv[4], cv[4], rv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = idx + u * WAVE < cells;
                    const int i = ok ? S.rows_nz[a] : 0, q = ok ? S.cols_nz[b] : 0;
                    ii[u] = i; qq[u] = q;
                    tv[u] = ok ? T[i * stride + q] : 0.0;
                    cv[u] = ok ? S.col[i] : 0.0;
                    rv[u] = ok ? S.rowr[q] : 0.0;
                    a += di; b += dq;
                    if (b >= nc) { b -= nc; ++a; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (idx + u * WAVE < cells) {
                        double v;
                        if (ii[u] == r) v = rv[u];
                        else if (qq[u] == jin) v = 0.0;
                        else v = tv[u] - cv[u] * rv[u];
                        T[ii[u] * stride + qq[u]] = v;
                    }
                }
            }
        }
        if (lane == 0) basis[r] = jin;
        wave_sync<IN_LDS>();
        LP_STAMP(t_e);
        LP_ACC(3, t_d, t_e);
        LP_ACC(5, t_a, t_a + 1);       // pivot count
        const double wn = -T[mc * stride + n];
        if (wn < w - progress) { stall = 0; bland = false; }
        else if (++stall > LP_STALL) bland = true;
        w = wn;
        if (++pivots >= LP_MAX_PIVOTS) { *error = true; w = artificial_sum(T, stride, m, n, basis, lane); break; }
    }
    *pivots_out = pivots;
    LP_PROF_FLUSH;
    return w;
}

// Independent check of a "feasible" verdict: read the basic solution x off the tableau and evaluate the ORIGINAL
// equilibrium rows  sum_j M_ij x_j - w_i  again from the contact list (nothing of the pivoted tableau is reused).
// A tableau damaged by an ill-conditioned pivot cannot pass this.  Returns the L1 residual over rows < m_chk.
template <typename TP, typename SC>
__device__ inline double lp_verify(TP T, int stride, int m, int m_chk, int n, SC& S, const AsmView& A, double mu,
                                   double density, int lane) {
    for (int q = lane; q < n; q += WAVE) S.rowr[q] = 0.0;
    __syncthreads();
    if (lane <= m && S.basis[lane] >= 0 && S.basis[lane] < n) {     // rows incl. the budget row; generators only
        double v = T[lane * stride + n + 1];
        S.rowr[S.basis[lane]] = v > 0.0 ? v : 0.0;
    }
    __syncthreads();
    double res = 0.0;
    if (lane < m_chk && S.basis[lane] != -1) {                      // enforced rows only (passive rows have basis -1)
        // row -> (block, component): rows are 3 per free block in block order (row_of)
        int b = -1;
        for (int k = 0; k < A.n_blocks; ++k) if (S.row_of[k] >= 0 && S.row_of[k] <= lane && lane < S.row_of[k] + 3) b = k;
        const int comp = lane - S.row_of[b];
        double gcx, gcz;
        A.centroid(b, gcx, gcz);
        double acc = 0.0;
        for (int k = 0; k < A.n_if; ++k) {
            const int32_t* bd = A.ib(k);
            const int bA = bd[0], bB = bd[1];
            if (bA != b && bB != b) continue;
            const double sign = (bB == b) ? 1.0 : -1.0;
            const double* g = A.ig(k);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int ip = c4 >> 1, ig = c4 & 1;
                const double x = S.rowr[4 * k + c4];
                if (x == 0.0) continue;
                const double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
                const double gx = sign * (ig ? g[4] - mu * g[6] : g[4] + mu * g[6]);
                const double gz = sign * (ig ? g[5] - mu * g[7] : g[5] + mu * g[7]);
                const double coef = comp == 0 ? gx : (comp == 1 ? gz : (px - gcx) * gz - (pz - gcz) * gx);
                acc += coef * x;
            }
            if (A.n_tens) {
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const double x = S.rowr[4 * A.n_if + 2 * k + ip];
                    if (x == 0.0) continue;
                    const double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
                    const double gx = -sign * g[4], gz = -sign * g[5];
                    const double coef = comp == 0 ? gx : (comp == 1 ? gz : (px - gcx) * gz - (pz - gcz) * gx);
                    acc += coef * x;
                }
            }
        }
        const double rhs = comp == 1 ? density * A.volume(b) : 0.0;
        res = fabs(acc - rhs);
    }
    return wave_sum_d(res);
}

// Enforce the passive rows [m_act, m): give each an artificial (negating the row if its rhs went negative) and
// price it into the cost row.  The tableau then continues from the basis reached so far (warm start).
template <bool IN_LDS, typename TP, typename SC>
__device__ inline void lp_activate_rows(TP T, int stride, int m, int m_act, int n_gen, SC& S, int lane, int ncarr = 0) {
    const int n = n_gen + 1;
    for (int i = m_act; i < m; ++i) {
        const bool neg = T[i * stride + n] < 0.0;                  // uniform
        wave_sync<IN_LDS>();
        for (int q = lane; q <= n + ncarr; q += WAVE) {
            double v = T[i * stride + q];
            if (neg) { v = -v; T[i * stride + q] = v; }
            T[(m + 1) * stride + q] -= v;
        }
        if (lane == 0) S.basis[i] = n + i;
    }
    wave_sync<IN_LDS>();
}

// Shared setup of a solve: row map of the free blocks.
template <typename SC>
__device__ inline void lp_row_map(SC& S, uint32_t free_mask, int lane) {
    __syncthreads();
    if (lane < MAXK)
        S.row_of[lane] = ((free_mask >> lane) & 1u) ? 3 * __popc(free_mask & ((1u << lane) - 1u)) : -1;
    __syncthreads();
}

// Tableau geometry of an assembly with n_free free blocks and n_if interfaces: rows m = 3 n_free (+ budget + cost),
// generator columns n = 4 n_if, odd row stride (conflict-free column reads).
__device__ __forceinline__ void lp_dims(int n_free, int n_if, int& m, int& n, int& stride, int64_t& cells, bool carriers = false,
                                        int n_tens = 0) {
    m = 3 * n_free;
    n = 4 * n_if + 2 * n_tens;
    stride = n + 2 + (carriers ? m : 0);
    if ((stride & 1) == 0) stride += 1;
    cells = (int64_t)(m + 2) * stride;
}

// Stability of one assembly variant.  fixed_mask bit b = block b is_static (fixed).  tab_lds holds lds_cap doubles
// and the scratch MAXCOLS generator columns; a tableau that does not fit goes to tab_ws (ws_cap doubles, may be 0).
// *too_big is set (and false returned, without an error) when neither fits -- the caller decides what that means.
template <typename SC>
__device__ inline bool rbe_stable(double* tab_lds, int lds_cap, int max_cols, double* tab_ws, int64_t ws_cap, SC& S,
                                  const AsmView& A, uint32_t fixed_mask, double mu, double density, int lane,
                                  double* w_out, int* pivots_out, bool* error, bool* too_big) {
    *w_out = 0.0;
    *pivots_out = 0;
    const int n_blocks = A.n_blocks;
    const uint32_t all = n_blocks >= 32 ? 0xffffffffu : ((1u << n_blocks) - 1u);
    const uint32_t free_mask = all & ~fixed_mask;
    const int n_free = __popc(free_mask);
    if (A.n_if == 0) return n_free == 0;               // stability.py:53-56
    if (n_free == 0) return true;
    int m, n, stride;
    int64_t cells;
    lp_dims(n_free, A.n_if, m, n, stride, cells, false, A.n_tens);
    const double feas = RBE_FEAS_TOL * density, vtol = LP_VERIFY_TOL * density;
    double w;
    if (cells <= lds_cap && n <= max_cols) {          // LDS path: address space known at compile time (ds_read/ds_write)
        lp_row_map(S, free_mask, lane);
        lp_build(tab_lds, stride, m, m, n, A, S.row_of, mu, density, lane);
        w = lp_phase1<true>(tab_lds, stride, m, m, n, S, lane, pivots_out, error, true, feas);
        if (w <= feas && lp_verify(tab_lds, stride, m, m, n, S, A, mu, density, lane) > vtol) { *error = true; w = 1e300; }
    } else {
        if (cells > ws_cap || n > max_cols) { *too_big = true; return false; }
        lp_row_map(S, free_mask, lane);
        lp_build(tab_ws, stride, m, m, n, A, S.row_of, mu, density, lane);
        w = lp_phase1<false>(tab_ws, stride, m, m, n, S, lane, pivots_out, error, true, feas);
        if (w <= feas && lp_verify(tab_ws, stride, m, m, n, S, A, mu, density, lane) > vtol) { *error = true; w = 1e300; }
    }
    *w_out = w;
    return w <= feas;
}

// ---- persistent tableau of an environment (incremental simplex) --------------------------------------------------
// Between two lock-steps the assembly only GROWS: one block, its contacts.  The equilibrium LP of the next step is
// therefore the LP just solved plus three rows (the new block) and four columns per new contact, and "last block
// frozen" of the next step enforces exactly the rows "nothing frozen" enforced in this one.  So k_step keeps the final
// tableau of every live environment in lp_ws (header + basis + cells, two halves for tableaux that are worked on in
// global memory) and the next step continues from it: old cells are copied over (column / row indices remapped, the
// layout is always  generators | slack | rhs | carriers), the new columns enter as  sum_i a_i * carrier_i + slack
// (their budget coefficient is 1), the new rows are appended untouched (no old column reaches into them), the phase-1
// cost row is rebuilt from the rows whose artificial is still basic.  Stage 1 then needs no pivot at all when the
// previous assembly was stable unfrozen, stage 2 about three -- instead of ~2 per row from an all-artificial start
// (measured before: 14 pivots per step on average, 80 for 12 blocks; a pivot is ~3750 cycles whatever the size).
// Every verdict is still re-checked against the original rows (lp_verify); a warm tableau that fails the check is
// rebuilt from scratch (cold path) before anything is reported.
#define WARM_HDR_DOUBLES 64                                  // header (16 int32) + basis (64 int32) + padding
#define WARM_MAX_STRIDE (LP_MAX_COLS + 2 + 3 * MAXK + 1)
#define WARM_HALF ((int64_t)(3 * MAXK + 2) * WARM_MAX_STRIDE)
#define WARM_WS_DOUBLES (WARM_HDR_DOUBLES + 2 * WARM_HALF)   // bridges_env_buffers.lp_ws_stride must be >= this
#define WARM_MAGIC 0x57A7B1E5

struct WarmHdr {
    int32_t magic, n_blocks, n_if, stride, half, m, pad_[10];
    int32_t basis[64];
};

// What k_step requests of the persisted tableau before it knows anything else about the step: the header with the
// first batch of loads, the cells (up to WARM_PRE per lane, compact index lane + 64 u over rows 0..m_o x the old
// columns) with the second -- so that continuing costs no further global-memory round trip for all but the largest
// tableaux.
#define WARM_PRE 16
struct WarmPre {
    int32_t magic, n_blocks, n_if, stride, half, m;
    int32_t basis_lane;                 // hdr.basis[lane]
    bool ok;                            // the header describes exactly the assembly without the new block
    int n_pre;                          // cells prefetched per lane: WARM_PRE, or 0 (lp_warm_prepare reads them in place)
    double cell[WARM_PRE];
};

__device__ __forceinline__ void warm_header(WarmPre& W, const double* ws, int lane) {
    const WarmHdr* h = reinterpret_cast<const WarmHdr*>(ws);
    W.magic = h->magic; W.n_blocks = h->n_blocks; W.n_if = h->n_if; W.stride = h->stride; W.half = h->half; W.m = h->m;
    W.basis_lane = h->basis[lane];
}

__device__ __forceinline__ bool warm_matches(const WarmPre& W, int n_old_blocks, int n_if_old) {
    return W.magic == WARM_MAGIC && W.n_blocks == n_old_blocks && W.n_if == n_if_old && n_old_blocks >= 1 &&
           W.m == 3 * n_old_blocks;
}

__device__ __forceinline__ void warm_prefetch(WarmPre& W, const double* ws, int n_old_blocks, int n_if_old, int lane) {
    W.ok = warm_matches(W, n_old_blocks, n_if_old);
    W.n_pre = WARM_PRE;
    const int m_o = 3 * n_old_blocks, ncols_o = 4 * n_if_old + 2 + m_o;
    const int cells_o = W.ok ? (m_o + 1) * ncols_o : 0;
    const double* src = ws + WARM_HDR_DOUBLES + (int64_t)(W.half & 1) * WARM_HALF;
    const int di = WAVE / ncols_o, dq = WAVE - di * ncols_o;     // same stepping as lp_warm_prepare
    int i = lane / ncols_o, q = lane - i * ncols_o;
#pragma unroll
    for (int u = 0; u < WARM_PRE; ++u) {
        const int idx = lane + WAVE * u;
        double v = 0.0;
        if (idx < cells_o) v = src[(size_t)i * W.stride + q];
        W.cell[u] = v;
        i += di; q += dq;
        if (q >= ncols_o) { q -= ncols_o; ++i; }
    }
}

// Continue from the persisted tableau `src` (stride_o, m_o rows + budget, n_gen_o generators, carriers for its m_o
// rows): fill T (stride, m = m_o + 3 rows, n_gen generators) and S.basis, rebuild the cost row for stage 1 (active
// rows = the old ones).  A.n_blocks - 1 is the new block, interfaces >= n_gen_o / 4 are its contacts.
template <bool IN_LDS, typename TP, typename SC>
__device__ inline void lp_warm_prepare(TP T, int stride, int m, int n_gen, const double* src, int stride_o, int m_o,
                                       int n_gen_o, const WarmPre& W, SC& S, const AsmView& A, double mu,
                                       double density, int lane) {
    const int n = n_gen + 1, n_o = n_gen_o + 1;                 // rhs column index (new / old)
    const int ncols_o = n_o + 1 + m_o;
    const int cells = (m + 2) * stride;
    for (int i = lane; i < cells; i += WAVE) T[i] = 0.0;
    wave_sync<IN_LDS>();
    // old rows (equilibrium rows keep their index, the budget row moves from m_o to m), old columns remapped
    const int cells_o = (m_o + 1) * ncols_o;
    // cell idx = lane + 64 u of the old tableau <-> (row i, column q) = (idx / ncols_o, idx % ncols_o), stepped from one
    // division per lane (a division per cell was ~25 instructions each)
    const int di_o = WAVE / ncols_o, dq_o = WAVE - di_o * ncols_o;
    int ci_o = lane / ncols_o, cq_o = lane - ci_o * ncols_o;
#pragma unroll
    for (int u = 0; u < WARM_PRE; ++u) {                         // the prefetched cells
        const int idx = lane + WAVE * u;
        if (u < W.n_pre && idx < cells_o) {
            const int i = ci_o, q = cq_o;
            const int qn = q < n_gen_o ? q : (q == n_gen_o ? n_gen : (q == n_o ? n : n + 1 + (q - n_o - 1)));
            const int in = i < m_o ? i : m;
            T[in * stride + qn] = W.cell[u];
        }
        if (u < W.n_pre) {
            ci_o += di_o; cq_o += dq_o;
            if (cq_o >= ncols_o) { cq_o -= ncols_o; ++ci_o; }
        }
    }
    for (int idx = lane + WAVE * W.n_pre; idx < cells_o; idx += WAVE) {     // the rest (everything without a prefetch)
        const int i = ci_o, q = cq_o;
        const double v = src[(size_t)i * stride_o + q];
        const int qn = q < n_gen_o ? q : (q == n_gen_o ? n_gen : (q == n_o ? n : n + 1 + (q - n_o - 1)));
        const int in = i < m_o ? i : m;
        T[in * stride + qn] = v;
        ci_o += di_o; cq_o += dq_o;
        if (cq_o >= ncols_o) { cq_o -= ncols_o; ++ci_o; }
    }
    if (lane <= m_o) {
        const int b = W.basis_lane;
        const int bn = b < n_gen_o ? b : (b == n_gen_o ? n_gen : n + (b - n_o));
        S.basis[lane < m_o ? lane : m] = bn;
    }
    if (lane >= m_o && lane < m) S.basis[lane] = -1;            // the new block's rows are passive in stage 1
    wave_sync<IN_LDS>();
    // the new block's rows: rhs (perturbation + weight on the Fz row), carrier = identity
    const int nbn = A.n_blocks - 1;
    if (lane < 3) {
        const int i = m_o + lane;
        double rhs = (LP_PERTURB * density) * (1.0 + 0.37 * (double)(i % 7) + 0.0618 * (double)(i % 11));
        if (lane == 1) rhs += density * A.volume(nbn);
        T[i * stride + n] = rhs;
        T[i * stride + n + 1 + i] = 1.0;
    }
    // new generator columns: R * a for the old rows and the budget row, raw entries in the new rows.  16 columns at a time:
    // lanes over COLUMNS build each column's descriptor once (what it adds to the carriers of body A, its three entries in
    // the new block's rows) into LDS scratch, then lanes over ROWS apply the descriptors -- the per-column geometry
    // (contact point, cone generator, two centroids) was evaluated by every lane for every column before (~60 wave
    // instructions per column, a third of a continued solve's set-up).  Same expressions, same order: the tableau is
    // bit for bit the one the column-by-column loop built.
    {
        const int n_newc = n_gen - n_gen_o;
        double ncx = 0.0, ncz = 0.0;
        A.centroid(nbn, ncx, ncz);
        for (int j0 = 0; j0 < n_newc; j0 += 16) {
            wave_sync<IN_LDS>();
            if (lane < 16 && j0 + lane < n_newc) {
                const int j = n_gen_o + j0 + lane;
                const int k = j >> 2, ip = (j >> 1) & 1, ig = j & 1;
                const double* g = A.ig(k);
                const int32_t* bd = A.ib(k);
                const double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
                const double nx = g[4], nz = g[5], tx = g[6], tz = g[7];
                const double gx = ig ? nx - mu * tx : nx + mu * tx;
                const double gz = ig ? nz - mu * tz : nz + mu * tz;
                const int bodyA = bd[0];                             // floor (-1) or an older block; body B is the new block
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
                if (bodyA >= 0) {
                    double gcx, gcz;
                    A.centroid(bodyA, gcx, gcz);
                    const double rx = px - gcx, rz = pz - gcz;
                    a0 = -gx; a1 = -gz; a2 = rx * a1 - rz * a0;
                }
                const double rxn = px - ncx, rzn = pz - ncz;
                double* d = S.rowr + 7 * lane;
                d[0] = a0; d[1] = a1; d[2] = a2; d[3] = (double)bodyA;
                d[4] = gx; d[5] = gz; d[6] = rxn * gz - rzn * gx;
            }
            wave_sync<IN_LDS>();
            const int nb16 = (n_newc - j0) < 16 ? (n_newc - j0) : 16;
            if (lane <= m_o) {
                const int row = lane < m_o ? lane : m;
                const double slack = T[row * stride + n_gen];        // budget coefficient 1 times the slack column
                for (int u = 0; u < nb16; ++u) {
                    const double* d = S.rowr + 7 * u;
                    const int bodyA = (int)d[3];
                    double v = slack;
                    if (bodyA >= 0) {
                        const int c0 = n + 1 + 3 * bodyA;
                        v += d[0] * T[row * stride + c0] + d[1] * T[row * stride + c0 + 1] + d[2] * T[row * stride + c0 + 2];
                    }
                    T[row * stride + n_gen_o + j0 + u] = v;
                }
            } else if (lane < m_o + 4) {
                const int cmp = lane - m_o - 1;                      // 0, 1, 2: Fx, Fz, My of the new block
                for (int u = 0; u < nb16; ++u) T[(m_o + cmp) * stride + n_gen_o + j0 + u] = S.rowr[7 * u + 4 + cmp];
            }
        }
    }
    wave_sync<IN_LDS>();
    // phase-1 cost row over the structural columns and the rhs: minus the rows whose artificial is still basic
    for (int q = lane; q <= n; q += WAVE) {
        double sacc = 0.0;
        for (int i = 0; i < m_o; ++i)
            if (S.basis[i] >= n) sacc += T[i * stride + q];
        T[(m + 1) * stride + q] = -sacc;
    }
    wave_sync<IN_LDS>();
}

// Persist rows 0..m (equilibrium + budget) of the LDS tableau.
__device__ inline void lp_warm_store(const double* T, int stride, int m, int n_gen, double* dst, int lane) {
    const int ncols = n_gen + 2 + m;
    const int cells = (m + 1) * ncols;
    const int di = WAVE / ncols, dq = WAVE - di * ncols;         // (row, column) of cell idx stepped, not divided per cell
    int i = lane / ncols, q = lane - i * ncols;
    for (int idx = lane; idx < cells; idx += WAVE) {
        dst[(size_t)i * stride + q] = T[i * stride + q];
        i += di; q += dq;
        if (q >= ncols) { q -= ncols; ++i; }
    }
}

// Stage 1 (last block frozen: its rows passive) and stage 2 (nothing frozen) of gym_env.py:325-333 on one tableau
// that is either freshly built (warm == false) or continued from the persisted one (prepared by lp_warm_prepare).
template <bool IN_LDS, typename TP>
__device__ inline void rbe_both_run(TP T, int stride, int m, int n, LpScratch& S, const AsmView& A, double mu, double density,
                                    int lane, bool warm, bool* st_frozen, bool* st_free, bool* error, int* pivots,
                                    double* snap, bool* marginal) {
    const int m_act = m - 3;
    const double feas = RBE_FEAS_TOL * density, vtol = LP_VERIFY_TOL * density;
    // an optimum within (LP_MARGIN_LO, LP_MARGIN) x the threshold is the verdict a continued tableau could owe to accumulated
    // round-off and to the rhs perturbations its rows were added with: the caller re-solves it from scratch
    const double near = LP_MARGIN * feas;
    *marginal = false;
    if (!warm) lp_build(T, stride, m, m_act, n, A, S.row_of, mu, density, lane, m);
    int& piv = *pivots;
    double w = lp_phase1<IN_LDS>(T, stride, m, m_act, n, S, lane, &piv, error, !warm, feas, m);
    *st_frozen = w <= feas;
    if (w > LP_MARGIN_LO * feas && w < near) *marginal = true;
    if (*st_frozen && m_act > 0 && lp_verify(T, stride, m, m_act, n, S, A, mu, density, lane) > vtol) {
        *st_frozen = false;                 // the verdict does not survive the check on the original rows
        *error = true;
    }
    if (!*st_frozen) { *st_free = false; return; }
    if (snap) {
        // the tableau of "last block frozen" is what the candidate-stability LPs of the NEXT state continue from (the
        // frozen block's rows are in it, passive: basis -1)
        wave_sync<IN_LDS>();
        lp_warm_store(T, stride, m, n, snap + WARM_HDR_DOUBLES, lane);
        WarmHdr* sh = reinterpret_cast<WarmHdr*>(snap);
        if (lane <= m) sh->basis[lane] = S.basis[lane];
        if (lane == 0) {
            sh->n_blocks = A.n_blocks; sh->n_if = A.n_if; sh->stride = stride; sh->half = 0; sh->m = m;
            sh->magic = WARM_MAGIC;
        }
    }
    lp_activate_rows<IN_LDS>(T, stride, m, m_act, n, S, lane, m);
    w = lp_phase1<IN_LDS>(T, stride, m, m, n, S, lane, &piv, error, false, feas, m);
    *st_free = w <= feas;
    if (w > LP_MARGIN_LO * feas && w < near) *marginal = true;
    if (*st_free && lp_verify(T, stride, m, m, n, S, A, mu, density, lane) > vtol) {
        *st_free = false;
        *error = true;
    }
}

// k_step's solve: both stability variants of the assembly A (A.n_blocks blocks, the last one new), continuing from the
// environment's persisted tableau in `ws` when it matches the state (n_blocks - 1 blocks, n_if_old interfaces), and
// persisting the result there when `keep`.  *warm_used reports which path produced the verdict.
__device__ inline void rbe_both(double* tab_lds, int lds_cap, double* ws, int64_t ws_cap, LpScratch& S, const AsmView& A, int n_if_old,
                                const WarmPre& W, double mu, double density, int lane, bool* st_frozen, bool* st_free,
                                bool* error, bool* warm_used, int* diag = nullptr, double* snap = nullptr, bool* resolved = nullptr) {
    WarmHdr* hdr = reinterpret_cast<WarmHdr*>(ws);
    double* halves = ws + WARM_HDR_DOUBLES;
    *warm_used = false;
    const int nb = A.n_blocks;
    if (A.n_if == 0) {                                 // stability.py:53-56: no edges -> stable iff no free node
        *st_frozen = nb == 1;
        *st_free = false;
        if (lane == 0) { hdr->magic = 0; if (snap) reinterpret_cast<WarmHdr*>(snap)->magic = 0; }
        return;
    }
    if (snap && lane == 0) reinterpret_cast<WarmHdr*>(snap)->magic = 0;     // rewritten below when stage 1 succeeds
    const uint32_t all = nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u);
    lp_row_map(S, all, lane);
    int m, n, stride;
    int64_t cells;
    lp_dims(nb, A.n_if, m, n, stride, cells, true);
    if (ws_cap < WARM_WS_DOUBLES || cells > WARM_HALF) {
        *error = true; *st_frozen = false; *st_free = false;
        if (lane == 0) hdr->magic = 0;
        return;
    }
    // the persisted tableau continues this state iff it was written for exactly the assembly without the new block
    bool warm = W.ok;
    const int half_o = W.half & 1, stride_o = W.stride;
    const bool in_lds = cells <= lds_cap;
    const int half_n = in_lds ? 0 : 1 - half_o;        // a tableau worked on in global memory alternates halves
    double* Tg = halves + (int64_t)half_n * WARM_HALF;
    int pivots = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool err = false;
        if (warm) {
            const double* src = halves + (int64_t)half_o * WARM_HALF;
            if (in_lds) lp_warm_prepare<true>(tab_lds, stride, m, n, src, stride_o, m - 3, 4 * n_if_old, W, S, A, mu, density, lane);
            else lp_warm_prepare<false>(Tg, stride, m, n, src, stride_o, m - 3, 4 * n_if_old, W, S, A, mu, density, lane);
        }
        bool marginal = false;
        if (in_lds) rbe_both_run<true>(tab_lds, stride, m, n, S, A, mu, density, lane, warm, st_frozen, st_free, &err, &pivots, snap, &marginal);
        else rbe_both_run<false>(Tg, stride, m, n, S, A, mu, density, lane, warm, st_frozen, st_free, &err, &pivots, snap, &marginal);
        if (diag) *diag = pivots * 4 + (in_lds ? 0 : 2) + attempt;     // pivots, global-memory tableau, second attempt
        if (!(warm && (err || marginal))) { *error = err; *warm_used = warm; break; }
        if (resolved) *resolved = true;
        warm = false;                                  // the continued tableau failed its check, or reported "unstable" by a
                                                       // margin round-off could explain: solve from scratch
        lp_row_map(S, all, lane);
    }
    __syncthreads();
    // persist for the next step (pointless when the episode ends here; the caller invalidates on reset)
    if (*st_frozen && !*error) {
        if (in_lds) lp_warm_store(tab_lds, stride, m, n, halves, lane);
        if (lane <= m) hdr->basis[lane] = S.basis[lane];
        if (lane == 0) {
            hdr->n_blocks = nb; hdr->n_if = A.n_if; hdr->stride = stride; hdr->half = half_n; hdr->m = m;
            hdr->magic = WARM_MAGIC;
        }
    } else if (lane == 0) {
        hdr->magic = 0;
    }
    __syncthreads();
}

// is_action_stable_rbe continued from the env's snapshot (the stage-1 tableau k_step persisted for the current state):
// the candidate is block A.n_blocks - 1, its contacts are the interfaces >= n_if_old.  The frozen block's rows stay
// passive, the candidate's three rows are activated right away.  Returns false in *fits when the tableau (with
// carriers) exceeds lds_cap doubles or the scratch's columns; *error when the verdict fails its check on the original
// rows (the caller then solves from scratch).
template <typename SC>
__device__ inline bool rbe_candidate_warm(double* tab_lds, int lds_cap, int max_cols, SC& S, const AsmView& A, int n_if_old,
                                          const WarmPre& W, const double* snap, double mu, double density, int lane,
                                          bool* fits, bool* error, int* pivots) {
    const int nb = A.n_blocks;                          // incl. the candidate
    int m, n, stride;
    int64_t cells;
    lp_dims(nb, A.n_if, m, n, stride, cells, true);
    *fits = cells <= lds_cap && n + 2 + m <= max_cols + 4 + 3 * MAXK;
    if (!*fits) return false;
    const uint32_t all = (1u << nb) - 1u;
    lp_row_map(S, all, lane);
    const double feas = RBE_FEAS_TOL * density, vtol = LP_VERIFY_TOL * density;
    lp_warm_prepare<true>(tab_lds, stride, m, n, snap + WARM_HDR_DOUBLES, W.stride, m - 3, 4 * n_if_old, W, S, A, mu, density, lane);
    lp_activate_rows<true>(tab_lds, stride, m, m - 3, n, S, lane, m);
    const double w = lp_phase1<true>(tab_lds, stride, m, m, n, S, lane, pivots, error, false, feas, m);
    bool stable = w <= feas;
    if (w > LP_MARGIN_LO * feas && w < LP_MARGIN * feas) *error = true;   // marginal verdict of a continued tableau: the caller re-solves cold
    if (stable && lp_verify(tab_lds, stride, m, m, n, S, A, mu, density, lane) > vtol) { *error = true; stable = false; }
    return stable;
}

}  // namespace bridges

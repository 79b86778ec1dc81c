// Register-resident phase-1 simplex for the small tableaux that make up almost all of the work.
//
// Same LP, same pivot rules, same arithmetic as lp_build / lp_phase1 / lp_verify in rbe_device.h (so the pivot
// sequence and therefore every intermediate value is identical), but the tableau never touches LDS: lane q of the
// wave owns COLUMN q (cone generators 0..n_gen-1, the budget slack n_gen) as RM + 2 doubles in registers, the
// right-hand side and the basis live in "row form" (lane i holds the entry of row i).  A pivot is then
//   price   : one DPP min over the cost entries (one per lane),
//   ratio   : the entering column is broadcast with v_readlane into row form, one DPP min + one DPP max,
//   update  : every lane updates its own column with uniform multipliers -- no staging, no barriers, no LDS traffic.
// Measured on MI355X the LDS tableau spends ~3750 shader cycles per pivot whatever its size (price 570, ratio 1045,
// stage 630, sweep 1500: LDS round trips and wave barriers, tools/lp_microbench.py); the loops here are unrolled over
// the RM row slots and leave at row m, so a pivot costs in proportion to the rows the LP really has.
// Limits: n_gen + 1 <= 64 columns (<= 15 interfaces), m <= RM equilibrium rows; anything larger takes the LDS path.
#pragma once
#include "rbe_device.h"

namespace bridges {

#define REG_MAX_GEN 63

__device__ __forceinline__ int reg_row_of(uint32_t free_mask, int b) {
    return ((free_mask >> b) & 1u) ? 3 * __popc(free_mask & ((1u << b) - 1u)) : -1;
}

// Tableau of lp_build: this lane's column (t[0..m), budget entry tb, cost entry tc) and the row-form right-hand
// side `brow` (lane i < m: rhs of row i, lane m: budget, lane m + 1: cost row rhs = -w).
template <int RM>
__device__ __forceinline__ void reg_build(double (&t)[RM], double& tb, double& tc, double& brow, int m, int m_act, int n_gen,
                                          const AsmView& A, uint32_t free_mask, double mu, double density, int lane) {
    int rB = -1, rA = -1;
    double gxB = 0.0, gzB = 0.0, moB = 0.0, gxA = 0.0, gzA = 0.0, moA = 0.0;
    if (lane < n_gen) {
        const int k = lane >> 2, ip = (lane >> 1) & 1, ig = lane & 1;
        const double* g = A.ig(k);
        const int32_t* bd = A.ib(k);
        const double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
        const double nx = g[4], nz = g[5], tx = g[6], tz = g[7];
        const double gx = ig ? nx - mu * tx : nx + mu * tx;
        const double gz = ig ? nz - mu * tz : nz + mu * tz;
        {   // body B (+)
            const int body = bd[1];
            rB = body >= 0 ? reg_row_of(free_mask, body) : -1;
            if (rB >= 0) {
                const bridges_shape& sh = A.S(body);
                const double* P = A.P(body);
                double rgx, rgz;
                rot2(sh.gx, sh.gz, P[2], P[3], rgx, rgz);
                const double rx = px - (P[0] + rgx), rz = pz - (P[1] + rgz);
                gxB = gx; gzB = gz; moB = rx * gz - rz * gx;
            }
        }
        {   // body A (-)
            const int body = bd[0];
            rA = body >= 0 ? reg_row_of(free_mask, body) : -1;
            if (rA >= 0) {
                const bridges_shape& sh = A.S(body);
                const double* P = A.P(body);
                double rgx, rgz;
                rot2(sh.gx, sh.gz, P[2], P[3], rgx, rgz);
                const double rx = px - (P[0] + rgx), rz = pz - (P[1] + rgz);
                const double sgx = -gx, sgz = -gz;
                gxA = sgx; gzA = sgz; moA = rx * sgz - rz * sgx;
            }
        }
    }
    double s = 0.0;                                   // cost entry: -(sum over the active rows, ascending)
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        double v = 0.0;
        v = (i == rB) ? gxB : v;
        v = (i == rB + 1 && rB >= 0) ? gzB : v;
        v = (i == rB + 2 && rB >= 0) ? moB : v;
        v = (i == rA) ? gxA : v;
        v = (i == rA + 1 && rA >= 0) ? gzA : v;
        v = (i == rA + 2 && rA >= 0) ? moA : v;
        t[i] = v;
        if (i < m_act) s += v;
    }
    tb = lane <= n_gen ? 1.0 : 0.0;                   // budget row: generators and the slack
    tc = -s;
    // right-hand side in row form
    double b = 0.0;
    if (lane < m) {
        b = (LP_PERTURB * density) * (1.0 + 0.37 * (double)(lane % 7) + 0.0618 * (double)(lane % 11));
        for (int blk = 0; blk < A.n_blocks; ++blk) {
            const int r = reg_row_of(free_mask, blk);
            if (r >= 0 && r + 1 == lane) b += density * A.S(blk).volume;
        }
    } else if (lane == m) {
        b = LP_S_MAX * density;
    }
    double sr = 0.0;                                  // cost rhs: -(sum of the active rows' rhs, ascending)
    for (int i = 0; i < m_act; ++i) sr += readlane_d(b, i);
    if (lane == m + 1) b = -sr;
    brow = b;
}

__device__ __forceinline__ double reg_art_sum(double brow, int basis, int m_act, int n, int lane) {
    const double v = (lane < m_act && basis >= n && brow > 0.0) ? brow : 0.0;
    return wave_sum_d(v);
}

// lp_phase1 on the register tableau.  basis: row form (lane i = basic variable of row i; lane m = the budget row).
template <int RM>
__device__ inline double reg_phase1(double (&t)[RM], double& tb, double& tc, double& brow, int& basis, int m, int m_act,
                                    int n_gen, int lane, int* pivots_io, bool* error, bool init_basis, double feas) {
    const int n = n_gen + 1;                          // structural columns incl. the slack
    const int mb = m, mc = m + 1;
    if (init_basis) basis = (lane < m_act) ? n + lane : (lane == mb ? n_gen : -1);
    int pivots = *pivots_io, stall = 0;
    bool bland = false;
    const double progress = 1e-7 * feas;
    double w = reg_art_sum(brow, basis, m_act, n, lane);
    for (;;) {
        if (w <= feas) {
            w = reg_art_sum(brow, basis, m_act, n, lane);
            if (w <= feas) break;
        }
        // ---- entering column ----
        int jin = -1;
        if (bland) {
            const uint64_t bal = __ballot(lane < n && tc < -LP_EPS_COST);
            if (bal) jin = __ffsll((long long)bal) - 1;
        } else {
            const double dbest = (lane < n && tc < 0.0) ? tc : 0.0;
            const double dmin = wave_min_d(dbest);
            if (dmin < -LP_EPS_COST) jin = __ffsll((long long)__ballot(dbest == dmin)) - 1;
        }
        if (jin < 0) {
            w = reg_art_sum(brow, basis, m_act, n, lane);
            break;
        }
        // ---- entering column into row form (lane i = T[i][jin]; lane m budget, lane m+1 cost) ----
        double col = 0.0;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            if (i < m) {
                const double ci = readlane_d(t[i], jin);
                col = (lane == i) ? ci : col;
            }
        }
        {
            const double cb = readlane_d(tb, jin), cc = readlane_d(tc, jin);
            col = (lane == mb) ? cb : col;
            col = (lane == mc) ? cc : col;
        }
        // ---- ratio test ----
        double ratio = 1e300;
        if ((lane < m_act || lane == mb) && col > LP_TAU) ratio = (brow > 0.0 ? brow : 0.0) * fast_rcp(col);
        const double rmin = wave_min_d(ratio);
        if (rmin >= 1e300) {                          // no usable pivot in this column: retire it
            if (lane == jin) tc = 0.0;
            continue;
        }
        const bool tie = ratio <= rmin + LP_TIE * (1.0 + rmin);
        int r;
        if (bland) {
            const int var = tie ? basis : 0x7fffffff;
            const int vmin = wave_min_i(var);
            r = __ffsll((long long)__ballot(tie && var == vmin)) - 1;
        } else {
            const double cmax = wave_max_d(tie ? col : -1e300);
            r = __ffsll((long long)__ballot(tie && col == cmax)) - 1;
        }
        const double ipiv = fast_rcp(readlane_d(col, r));
        // ---- normalised pivot row entry of this lane's column ----
        double tr = tb;                               // r == m: the budget row
#pragma unroll
        for (int i = 0; i < RM; ++i)
            if (i < m) tr = (i == r) ? t[i] : tr;
        const double rv = (lane == jin) ? 1.0 : tr * ipiv;
        const bool act = rv != 0.0;                   // columns with a zero pivot-row entry are not touched
        // ---- elimination: rows with a non-zero entering-column entry ----
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            if (i < m) {
                if (i == r) {
                    t[i] = act ? rv : t[i];
                } else {
                    const double ci = readlane_d(col, i);
                    if (ci != 0.0) {
                        double nv = t[i] - ci * rv;
                        nv = (lane == jin) ? 0.0 : nv;
                        t[i] = act ? nv : t[i];
                    }
                }
            }
        }
        {
            const double cb = readlane_d(col, mb), cc = readlane_d(col, mc);
            if (r == mb) {
                tb = act ? rv : tb;
            } else if (cb != 0.0) {
                double nv = tb - cb * rv;
                nv = (lane == jin) ? 0.0 : nv;
                tb = act ? nv : tb;
            }
            if (cc != 0.0) {
                double nv = tc - cc * rv;
                nv = (lane == jin) ? 0.0 : nv;
                tc = act ? nv : tc;
            }
        }
        // ---- right-hand side (row form) and basis ----
        {
            const double br = readlane_d(brow, r) * ipiv;
            if (br != 0.0) {
                if (lane == r) brow = br;
                else if (lane <= mc && col != 0.0) brow = brow - col * br;
            }
        }
        if (lane == r) basis = jin;
        const double wn = -readlane_d(brow, mc);
        if (wn < w - progress) { stall = 0; bland = false; }
        else if (++stall > LP_STALL) bland = true;
        w = wn;
        if (++pivots >= LP_MAX_PIVOTS) { *error = true; w = reg_art_sum(brow, basis, m_act, n, lane); break; }
    }
    *pivots_io = pivots;
    return w;
}

// lp_activate_rows: enforce the passive rows [m_act, m).
template <int RM>
__device__ __forceinline__ void reg_activate_rows(double (&t)[RM], double& tc, double& brow, int& basis, int m, int m_act,
                                                  int n_gen, int lane) {
    const int n = n_gen + 1, mc = m + 1;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        if (i >= m_act && i < m) {
            double bi = readlane_d(brow, i);
            const bool neg = bi < 0.0;                // uniform
            double v = t[i];
            if (neg) { v = -v; t[i] = v; bi = -bi; if (lane == i) brow = bi; }
            tc -= v;
            if (lane == mc) brow -= bi;
            if (lane == i) basis = n + i;
        }
    }
}

// lp_verify: residual of the ORIGINAL equilibrium rows at the basic solution.  xsol = LDS scratch of >= 64 doubles.
__device__ inline double reg_verify(double* xsol, double brow, int basis, int m, int m_chk, int n_gen, const AsmView& A,
                                    uint32_t free_mask, double mu, double density, int lane) {
    __syncthreads();
    xsol[lane] = 0.0;
    __syncthreads();
    if (lane <= m && basis >= 0 && basis < n_gen) xsol[basis] = brow > 0.0 ? brow : 0.0;
    __syncthreads();
    double res = 0.0;
    if (lane < m_chk) {
        int b = -1, rb = 0;
        for (int k = 0; k < A.n_blocks; ++k) {
            const int r = reg_row_of(free_mask, k);
            if (r >= 0 && r <= lane && lane < r + 3) { b = k; rb = r; }
        }
        const int comp = lane - rb;
        const bridges_shape& sh = A.S(b);
        const double* P = A.P(b);
        double rgx, rgz;
        rot2(sh.gx, sh.gz, P[2], P[3], rgx, rgz);
        const double gcx = P[0] + rgx, gcz = P[1] + rgz;
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
                const double x = xsol[4 * k + c4];
                if (x == 0.0) continue;
                const double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
                const double gx = sign * (ig ? g[4] - mu * g[6] : g[4] + mu * g[6]);
                const double gz = sign * (ig ? g[5] - mu * g[7] : g[5] + mu * g[7]);
                const double coef = comp == 0 ? gx : (comp == 1 ? gz : (px - gcx) * gz - (pz - gcz) * gx);
                acc += coef * x;
            }
        }
        const double rhs = comp == 1 ? density * sh.volume : 0.0;
        res = fabs(acc - rhs);
    }
    return wave_sum_d(res);
}

__device__ __forceinline__ bool reg_fits(int m, int n_if, int rm) { return 4 * n_if <= REG_MAX_GEN && m <= rm; }

// rbe_stable on the register tableau (one variant, fixed_mask given).  Caller checked reg_fits.
template <int RM>
__device__ inline bool reg_stable(double* xsol, const AsmView& A, uint32_t free_mask, int n_free, double mu, double density,
                                  int lane, double* w_out, int* pivots_out, bool* error) {
    const int m = 3 * n_free, n_gen = 4 * A.n_if;
    const double feas = RBE_FEAS_TOL * density, vtol = LP_VERIFY_TOL * density;
    double t[RM], tb, tc, brow;
    int basis;
    reg_build<RM>(t, tb, tc, brow, m, m, n_gen, A, free_mask, mu, density, lane);
    double w = reg_phase1<RM>(t, tb, tc, brow, basis, m, m, n_gen, lane, pivots_out, error, true, feas);
    if (w <= feas && reg_verify(xsol, brow, basis, m, m, n_gen, A, free_mask, mu, density, lane) > vtol) { *error = true; w = 1e300; }
    *w_out = w;
    return w <= feas;
}

// rbe_both_in on the register tableau: stage 1 = last block frozen (its rows passive), stage 2 = nothing frozen.
template <int RM>
__device__ inline void reg_both(double* xsol, const AsmView& A, double mu, double density, int lane, bool* st_frozen,
                                bool* st_free, bool* error) {
    const int nb = A.n_blocks;
    const uint32_t all = (1u << nb) - 1u;
    const int m = 3 * nb, m_act = m - 3, n_gen = 4 * A.n_if;
    const double feas = RBE_FEAS_TOL * density, vtol = LP_VERIFY_TOL * density;
    double t[RM], tb, tc, brow;
    int basis, piv = 0;
    reg_build<RM>(t, tb, tc, brow, m, m_act, n_gen, A, all, mu, density, lane);
    double w = reg_phase1<RM>(t, tb, tc, brow, basis, m, m_act, n_gen, lane, &piv, error, true, feas);
    *st_frozen = w <= feas;
    if (*st_frozen && m_act > 0 && reg_verify(xsol, brow, basis, m, m_act, n_gen, A, all, mu, density, lane) > vtol) {
        *st_frozen = false;
        *error = true;
    }
    if (!*st_frozen) { *st_free = false; return; }
    reg_activate_rows<RM>(t, tc, brow, basis, m, m_act, n_gen, lane);
    w = reg_phase1<RM>(t, tb, tc, brow, basis, m, m, n_gen, lane, &piv, error, false, feas);
    *st_free = w <= feas;
    if (*st_free && reg_verify(xsol, brow, basis, m, m, n_gen, A, all, mu, density, lane) > vtol) {
        *st_free = false;
        *error = true;
    }
}

// ---- dispatch: register tableau when it fits, the LDS / global tableau of rbe_device.h otherwise ----
template <int RM>
__device__ inline void rbe_both_auto(double* tab_lds, double* tab_ws, int64_t ws_cap, LpScratch& S, const AsmView& A,
                                     double mu, double density, int lane, bool* st_frozen, bool* st_free, bool* error) {
    if (A.n_if > 0 && reg_fits(3 * A.n_blocks, A.n_if, RM)) {
        reg_both<RM>(S.rowr, A, mu, density, lane, st_frozen, st_free, error);
        __syncthreads();
        return;
    }
    rbe_both(tab_lds, tab_ws, ws_cap, S, A, mu, density, lane, st_frozen, st_free, error);
}

template <int RM, typename SC>
__device__ inline bool rbe_stable_auto(double* tab_lds, int lds_cap, int max_cols, double* tab_ws, int64_t ws_cap, SC& S,
                                       const AsmView& A, uint32_t fixed_mask, double mu, double density, int lane,
                                       double* w_out, int* pivots_out, bool* error, bool* too_big) {
    const int n_blocks = A.n_blocks;
    const uint32_t all = n_blocks >= 32 ? 0xffffffffu : ((1u << n_blocks) - 1u);
    const uint32_t free_mask = all & ~fixed_mask;
    const int n_free = __popc(free_mask);
    if (A.n_if > 0 && n_free > 0 && reg_fits(3 * n_free, A.n_if, RM)) {
        *w_out = 0.0;
        *pivots_out = 0;
        return reg_stable<RM>(S.rowr, A, free_mask, n_free, mu, density, lane, w_out, pivots_out, error);
    }
    return rbe_stable(tab_lds, lds_cap, max_cols, tab_ws, ws_cap, S, A, fixed_mask, mu, density, lane, w_out, pivots_out, error,
                      too_big);
}

}  // namespace bridges

/*
 * oracle_env.c -- plain-C restatement of the hot path (TEST INFRASTRUCTURE / CPU baseline only).
 *
 * One environment, one thread, float64, no SIMD intrinsics, compiled -O2 -ffp-contract=off.
 * It follows the same reference code as the Python oracle (the oracle/ python modules, which it is tested against bit for bit):
 *   placement      AssemblyGym.create_block gym_env.py:204-216, align_frames_2d geometry.py:39-50
 *   interfaces     AssemblyEnv._reset_cra_assembly assembly_env.py:281-304 (compas_cra assembly_interfaces_numpy)
 *   stability      is_stable_rbe stability.py:49-71 (compas_cra rbe_solve; here: phase-1 simplex, see oracle/rbe.py)
 *   step/reward    AssemblyGym.step / terminated / sparse_reward / stabilities_freezing gym_env.py:11-22,141-253,325-333
 *   candidates     generate_actions actions.py:7-52, filter_actions actions.py:71-82, collision_on_action gym_env.py:304-323
 *   rasters        render_blocks_2d rendering.py:105-113, Shape.contains_2d assembly_env.py:126-137
 *   lin_reward     rollout_episode successor_dqn.py:397-401
 * Nothing in the product tree links or loads this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXV 6
#define MAXK 16
#define MAXIF 64
#define IMG 64
#define MAXG 24
#define MAXT 8

typedef struct {
    int32_t nv, pad_;
    double vx[MAXV], vz[MAXV];
    int32_t fa[MAXV], fb[MAXV];
    double fcx[MAXV], fcz[MAXV], fnx[MAXV], fnz[MAXV];
    double depth, volume, gx, gz;
} orc_shape;

typedef struct {
    int32_t max_steps, a_max, n_shapes, n_groups;
    int32_t group_shape[MAXG], group_face[MAXG];
    int32_t n_ground, n_offsets, n_targets, pad_;
    double mu, density, floor_half_width, floor_depth;
    double xlim[2], ylim[2];
    double targets[MAXT][3];
    double x_ground[32], offsets[8], grid_x[IMG], grid_y[IMG];
    uint64_t obstacle_bits[IMG];
    float reward_map[IMG * IMG];
    orc_shape shapes[8];
} orc_cfg;

typedef struct { double cx, cz, tx, tz, nx, nz; } frame2;

typedef struct {
    int32_t tb, tf, sh, fc;
    double ox, pose[4], verts[MAXV][2];
    uint64_t bits[IMG];
    float lin;
    uint8_t inb, mask;
} orc_cand;

typedef struct {
    int32_t valid_step, action_index, stable_frozen, stable_unfrozen, terminated, truncated, done, no_actions;
    int32_t n_blocks, n_reached, lp_pivots, pad_;
    double reward, lin_reward, pose[4];
} orc_out;

typedef struct {
    orc_cfg c;
    int32_t nb, shape[MAXK];
    double pose[MAXK][4], verts[MAXK][MAXV][2];
    uint8_t occ[MAXK];
    uint32_t targets_left;
    uint64_t state_bits[IMG];
    int32_t n_if, if_body[MAXIF][2];
    double if_geom[MAXIF][8];
    uint64_t draw_counter;
    int32_t needs_reset, n_cand, n_valid;
    orc_cand* cand;
    float* f32;           /* optional [a_max+1][64][64] f32 rasters (what the reference surfaces to the networks) */
    double* tab;          /* simplex tableau (3K+2) x (4*MAXIF+3) */
    long total_pivots;
} orc_env;

/* ---- arithmetic contract (oracle/shapes.py edge_frame, oracle/geometry.py rot/align) ---- */
static frame2 edge_frame(const double* a, const double* b) {
    frame2 f;
    f.cx = (a[0] + b[0]) * 0.5;
    f.cz = (a[1] + b[1]) * 0.5;
    double dx = b[0] - a[0], dz = b[1] - a[1];
    double L = sqrt(dx * dx + dz * dz);
    f.tx = dx / L;
    f.tz = dz / L;
    f.nx = -f.tz;
    f.nz = f.tx;
    return f;
}
static void rot2(double vx, double vz, double c, double s, double* ox, double* oz) {
    *ox = vx * c + vz * s;
    *oz = vz * c - vx * s;
}
static void align_place(const frame2* f1, double c2x, double c2z, double n2x, double n2z, double ox, double oy, double* pose) {
    double dot = f1->nx * n2x + f1->nz * n2z;
    double c = -dot;
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    double cy = f1->nz * n2x - f1->nx * n2z;
    double s = (cy + 1e-6 >= 0) ? fabs(cy) : -fabs(cy);
    double r2x, r2z;
    rot2(c2x, c2z, c, s, &r2x, &r2z);
    pose[0] = ((f1->cx + ox * f1->tx) + oy * f1->nx) - r2x;
    pose[1] = ((f1->cz + ox * f1->tz) + oy * f1->nz) - r2z;
    pose[2] = c;
    pose[3] = s;
}
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* ---- rasteriser (oracle/raster.py contains_2d) ---- */
static void raster_block(const orc_cfg* c, const orc_shape* sh, const double verts[MAXV][2], uint64_t* bits) {
    frame2 fr[MAXV];
    double zmin = 1e300, zmax = -1e300;
    for (int f = 0; f < sh->nv; ++f) {
        fr[f] = edge_frame(verts[sh->fa[f]], verts[sh->fb[f]]);
        if (verts[f][1] < zmin) zmin = verts[f][1];
        if (verts[f][1] > zmax) zmax = verts[f][1];
    }
    memset(bits, 0, IMG * sizeof(uint64_t));
    /* conservative row window: rows outside it fail some half-plane by a wide margin */
    double ytop = c->grid_y[0], dy = (c->grid_y[0] - c->grid_y[IMG - 1]) / (double)(IMG - 1);
    int r_lo = (int)floor((ytop - zmax) / dy) - 1, r_hi = (int)ceil((ytop - zmin) / dy) + 1;
    if (r_lo < 0) r_lo = 0;
    if (r_hi > IMG - 1) r_hi = IMG - 1;
    double tx[MAXV][IMG];
    for (int f = 0; f < sh->nv; ++f)
        for (int q = 0; q < IMG; ++q) tx[f][q] = (c->grid_x[q] - fr[f].cx) * fr[f].nx;
    for (int r = r_lo; r <= r_hi; ++r) {
        uint64_t m = ~0ull;
        for (int f = 0; f < sh->nv && m; ++f) {
            double tz = (c->grid_y[r] - fr[f].cz) * fr[f].nz;
            uint64_t mf = 0;
            for (int q = 0; q < IMG; ++q) mf |= (uint64_t)((tx[f][q] + tz) <= 0.0) << q;
            m &= mf;
        }
        bits[r] = m;
    }
}

/* ---- contact interfaces (oracle/rbe.py face_pair_contact) ---- */
typedef struct { double a[2], b[2]; frame2 fr; } face_t;
static int face_pair_contact(const face_t* A, const face_t* B, double depth, double* g) {
    double dotn = A->fr.nx * B->fr.nx + A->fr.nz * B->fr.nz;
    if (dotn > -1.0 + 1e-6) return 0;
    double cAx = A->fr.cx, cAz = A->fr.cz, tAx = A->fr.tx, tAz = A->fr.tz, nAx = A->fr.nx, nAz = A->fr.nz;
    double gap = (B->fr.cx - cAx) * nAx + (B->fr.cz - cAz) * nAz;
    if (fabs(gap) > 1e-6) return 0;
    double a0 = (A->a[0] - cAx) * tAx + (A->a[1] - cAz) * tAz, a1 = (A->b[0] - cAx) * tAx + (A->b[1] - cAz) * tAz;
    double b0 = (B->a[0] - cAx) * tAx + (B->a[1] - cAz) * tAz, b1 = (B->b[0] - cAx) * tAx + (B->b[1] - cAz) * tAz;
    double lo = fmax(fmin(a0, a1), fmin(b0, b1)), hi = fmin(fmax(a0, a1), fmax(b0, b1));
    if ((hi - lo) * depth < 0.001) return 0;
    g[0] = cAx + lo * tAx; g[1] = cAz + lo * tAz; g[2] = cAx + hi * tAx; g[3] = cAz + hi * tAz;
    g[4] = nAx; g[5] = nAz; g[6] = tAx; g[7] = tAz;
    return 1;
}
static void block_faces(const orc_env* e, int b, face_t* out) {
    const orc_shape* sh = &e->c.shapes[e->shape[b]];
    for (int f = 0; f < sh->nv; ++f) {
        memcpy(out[f].a, e->verts[b][sh->fa[f]], 16);
        memcpy(out[f].b, e->verts[b][sh->fb[f]], 16);
        out[f].fr = edge_frame(out[f].a, out[f].b);
    }
}
static void append_interfaces(orc_env* e, int nbn) {
    face_t fn[MAXV], fo[MAXV], floor_f;
    const orc_shape* shn = &e->c.shapes[e->shape[nbn]];
    block_faces(e, nbn, fn);
    floor_f.a[0] = -e->c.floor_half_width; floor_f.a[1] = 0.0; floor_f.b[0] = e->c.floor_half_width; floor_f.b[1] = 0.0;
    floor_f.fr.cx = 0; floor_f.fr.cz = 0; floor_f.fr.tx = 1; floor_f.fr.tz = 0; floor_f.fr.nx = 0; floor_f.fr.nz = 1;
    double g[8];
    for (int body = -1; body < nbn; ++body) {
        int nfo = 1;
        double depthA = e->c.floor_depth;
        if (body >= 0) { block_faces(e, body, fo); nfo = e->c.shapes[e->shape[body]].nv; depthA = e->c.shapes[e->shape[body]].depth; }
        for (int fa = 0; fa < nfo; ++fa)
            for (int fb = 0; fb < shn->nv; ++fb)
                if (face_pair_contact(body < 0 ? &floor_f : &fo[fa], &fn[fb], fmin(depthA, shn->depth), g) && e->n_if < MAXIF) {
                    e->if_body[e->n_if][0] = body; e->if_body[e->n_if][1] = nbn;
                    memcpy(e->if_geom[e->n_if], g, sizeof(g));
                    e->n_if++;
                }
    }
}

/* ---- phase-1 simplex: exists x >= 0 with M x = w and sum(x) <= S_MAX ?  (oracle/rbe.py; same rules as the device
 * kernel).  Tableau layout: rows [0, m) equilibrium, row m the force budget  sum_j x_j + s = S_MAX, row m+1 the
 * phase-1 cost; columns [0, n) cone generators, column n the budget slack s, column n+1 the right-hand side. ---- */
#define FEAS_TOL 1e-5
#define EPS_COST 1e-9
#define TAU 1e-5
#define TIE 1e-9
#define STALL 40
#define PERTURB 1e-8
#define S_MAX 1e4
static double art_sum(const double* T, int stride, int m_act, int nn, const int* basis) {
    double s = 0.0;
    for (int i = 0; i < m_act; ++i) if (basis[i] >= nn) { double r = T[i * stride + nn]; s += r > 0 ? r : 0; }
    return s;
}
/* m equilibrium rows are stored (rows >= m_act passive), the budget row m is always enforced; nn = n + 1 columns */
/* every absolute tolerance is a force and scales with the density (the rhs is linear in it, the matrix is not) */
static double phase1(double* T, int stride, int m, int m_act, int nn, int* basis, int init, long* pivots, double density) {
    const int cost = m + 1;
    const double feas = FEAS_TOL * density, progress = 1e-7 * feas;
    if (init) { for (int i = 0; i < m; ++i) basis[i] = i < m_act ? nn + i : -1; basis[m] = nn - 1; }
    int stall = 0, bland = 0, guard = 0;
    double w = art_sum(T, stride, m_act, nn, basis);
    while (w > feas) {
        int jin = -1;
        if (bland) { for (int j = 0; j < nn; ++j) if (T[cost * stride + j] < -EPS_COST) { jin = j; break; } }
        else { double best = -EPS_COST; for (int j = 0; j < nn; ++j) if (T[cost * stride + j] < best) { best = T[cost * stride + j]; jin = j; } }
        if (jin < 0) break;
        double rmin = 1e300;
        for (int i = 0; i <= m; ++i) {
            if (i >= m_act && i != m) continue;
            double a = T[i * stride + jin];
            if (a > TAU) { double r = T[i * stride + nn]; r = (r > 0 ? r : 0) / a; if (r < rmin) rmin = r; }
        }
        if (rmin >= 1e300) { T[cost * stride + jin] = 0.0; continue; }
        int r = -1; double cbest = -1e300; int vbest = 0x7fffffff;
        for (int i = 0; i <= m; ++i) {
            if (i >= m_act && i != m) continue;
            double a = T[i * stride + jin];
            if (a <= TAU) continue;
            double rr = T[i * stride + nn]; rr = (rr > 0 ? rr : 0) / a;
            if (rr > rmin + TIE * (1.0 + rmin)) continue;
            if (bland) { if (basis[i] < vbest) { vbest = basis[i]; r = i; } }
            else if (a > cbest) { cbest = a; r = i; }
        }
        double piv = T[r * stride + jin];
        for (int q = 0; q <= nn; ++q) T[r * stride + q] /= piv;
        T[r * stride + jin] = 1.0;
        for (int i = 0; i <= cost; ++i) {
            if (i == r) continue;
            double f = T[i * stride + jin];
            if (f == 0.0) continue;
            for (int q = 0; q <= nn; ++q) T[i * stride + q] -= f * T[r * stride + q];
            T[i * stride + jin] = 0.0;
        }
        basis[r] = jin;
        ++*pivots;
        double wn = art_sum(T, stride, m_act, nn, basis);
        if (wn < w - progress) { stall = 0; bland = 0; } else if (++stall > STALL) bland = 1;
        w = wn;
        if (++guard > 5000) break;
    }
    return w;
}
/* both variants of gym_env.py:325-333: stage 1 = last block frozen, stage 2 = nothing frozen (adds its 3 rows) */
static void rbe_both(orc_env* e, int* st_frozen, int* st_free) {
    int nb = e->nb, n_if = e->n_if;
    if (n_if == 0) { *st_frozen = nb == 1; *st_free = 0; return; }
    int m = 3 * nb, n = 4 * n_if, nn = n + 1, stride = nn + 1, m_act = m - 3, cost = m + 1;
    double* T = e->tab;
    int basis[3 * MAXK + 1];
    memset(T, 0, sizeof(double) * (size_t)(m + 2) * stride);
    for (int k = 0; k < n_if; ++k) {
        const double* g = e->if_geom[k];
        for (int ip = 0; ip < 2; ++ip) for (int ig = 0; ig < 2; ++ig) {
            int j = 4 * k + 2 * ip + ig;
            double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
            double gx = ig ? g[4] - e->c.mu * g[6] : g[4] + e->c.mu * g[6];
            double gz = ig ? g[5] - e->c.mu * g[7] : g[5] + e->c.mu * g[7];
            for (int side = 0; side < 2; ++side) {
                int body = e->if_body[k][side == 0 ? 1 : 0];
                if (body < 0) continue;
                double sgx = side == 0 ? gx : -gx, sgz = side == 0 ? gz : -gz;
                const orc_shape* sh = &e->c.shapes[e->shape[body]];
                double rgx, rgz;
                rot2(sh->gx, sh->gz, e->pose[body][2], e->pose[body][3], &rgx, &rgz);
                double rx = px - (e->pose[body][0] + rgx), rz = pz - (e->pose[body][1] + rgz);
                T[(3 * body + 0) * stride + j] = sgx;
                T[(3 * body + 1) * stride + j] = sgz;
                T[(3 * body + 2) * stride + j] = rx * sgz - rz * sgx;
            }
        }
    }
    for (int i = 0; i < m; ++i) T[i * stride + nn] = (PERTURB * e->c.density) * (1.0 + 0.37 * (double)(i % 7) + 0.0618 * (double)(i % 11));
    for (int b = 0; b < nb; ++b) T[(3 * b + 1) * stride + nn] += e->c.density * e->c.shapes[e->shape[b]].volume;
    for (int q = 0; q < nn; ++q) T[m * stride + q] = 1.0;
    T[m * stride + nn] = S_MAX * e->c.density;
    for (int q = 0; q <= nn; ++q) { double s = 0; for (int i = 0; i < m_act; ++i) s += T[i * stride + q]; T[cost * stride + q] = -s; }
    const double feas = FEAS_TOL * e->c.density;
    double w = phase1(T, stride, m, m_act, nn, basis, 1, &e->total_pivots, e->c.density);
    *st_frozen = w <= feas;
    if (!*st_frozen) { *st_free = 0; return; }
    for (int i = m_act; i < m; ++i) {
        int neg = T[i * stride + nn] < 0.0;
        for (int q = 0; q <= nn; ++q) { double v = T[i * stride + q]; if (neg) { v = -v; T[i * stride + q] = v; } T[cost * stride + q] -= v; }
        basis[i] = nn + i;
    }
    w = phase1(T, stride, m, m, nn, basis, 0, &e->total_pivots, e->c.density);
    *st_free = w <= feas;
}

/* is_stable_rbe with an arbitrary set of fixed blocks (bit b of fixed = block b is_static): one phase-1 solve over the
 * rows of the free blocks (oracle/rbe.py equilibrium_system). */
static int rbe_fixed(orc_env* e, uint32_t fixed) {
    int nb = e->nb, n_if = e->n_if, row_of[MAXK], n_free = 0;
    for (int b = 0; b < nb; ++b) row_of[b] = ((fixed >> b) & 1u) ? -1 : 3 * n_free++;
    if (n_if == 0) return n_free == 0;
    if (n_free == 0) return 1;
    int m = 3 * n_free, n = 4 * n_if, nn = n + 1, stride = nn + 1, cost = m + 1;
    double* T = e->tab;
    int basis[3 * MAXK + 1];
    memset(T, 0, sizeof(double) * (size_t)(m + 2) * stride);
    for (int k = 0; k < n_if; ++k) {
        const double* g = e->if_geom[k];
        for (int ip = 0; ip < 2; ++ip) for (int ig = 0; ig < 2; ++ig) {
            int j = 4 * k + 2 * ip + ig;
            double px = ip ? g[2] : g[0], pz = ip ? g[3] : g[1];
            double gx = ig ? g[4] - e->c.mu * g[6] : g[4] + e->c.mu * g[6];
            double gz = ig ? g[5] - e->c.mu * g[7] : g[5] + e->c.mu * g[7];
            for (int side = 0; side < 2; ++side) {
                int body = e->if_body[k][side == 0 ? 1 : 0];
                if (body < 0 || row_of[body] < 0) continue;
                double sgx = side == 0 ? gx : -gx, sgz = side == 0 ? gz : -gz;
                const orc_shape* sh = &e->c.shapes[e->shape[body]];
                double rgx, rgz;
                rot2(sh->gx, sh->gz, e->pose[body][2], e->pose[body][3], &rgx, &rgz);
                double rx = px - (e->pose[body][0] + rgx), rz = pz - (e->pose[body][1] + rgz);
                T[(row_of[body] + 0) * stride + j] = sgx;
                T[(row_of[body] + 1) * stride + j] = sgz;
                T[(row_of[body] + 2) * stride + j] = rx * sgz - rz * sgx;
            }
        }
    }
    for (int i = 0; i < m; ++i) T[i * stride + nn] = (PERTURB * e->c.density) * (1.0 + 0.37 * (double)(i % 7) + 0.0618 * (double)(i % 11));
    for (int b = 0; b < nb; ++b) if (row_of[b] >= 0) T[(row_of[b] + 1) * stride + nn] += e->c.density * e->c.shapes[e->shape[b]].volume;
    for (int q = 0; q < nn; ++q) T[m * stride + q] = 1.0;
    T[m * stride + nn] = S_MAX * e->c.density;
    for (int q = 0; q <= nn; ++q) { double s = 0; for (int i = 0; i < m; ++i) s += T[i * stride + q]; T[cost * stride + q] = -s; }
    double w = phase1(T, stride, m, m, nn, basis, 1, &e->total_pivots, e->c.density);
    return w <= FEAS_TOL * e->c.density;
}

/* is_action_stable_rbe (stability.py:122-130): candidate a appended as a free block, the last placed block stays
 * frozen (gym_env.py:238-240); the state is restored afterwards. */
static int action_stable(orc_env* e, int a) {
    const orc_cand* cd = &e->cand[a];
    int nb = e->nb, n_if = e->n_if;
    if (nb >= MAXK) return 0;
    e->shape[nb] = cd->sh;
    memcpy(e->pose[nb], cd->pose, sizeof(cd->pose));
    memcpy(e->verts[nb], cd->verts, sizeof(cd->verts));
    e->nb = nb + 1;
    append_interfaces(e, nb);
    int st = rbe_fixed(e, nb > 0 ? (1u << (nb - 1)) : 0u);
    e->nb = nb;
    e->n_if = n_if;
    return st;
}

/* ---- candidates of the current state (generate_actions + create_block + rasters + filter + lin) ---- */
static void reset_state(orc_env* e) {
    e->nb = 0; e->n_if = 0; e->needs_reset = 0;
    memset(e->occ, 0, sizeof(e->occ));
    memset(e->state_bits, 0, sizeof(e->state_bits));
    e->targets_left = e->c.n_targets >= 32 ? 0xffffffffu : ((1u << e->c.n_targets) - 1u);
}
static void refresh(orc_env* e) {
    const orc_cfg* c = &e->c;
    int fb_[MAXK * MAXV], ff_[MAXK * MAXV], nfree = 0;
    for (int b = 0; b < e->nb; ++b)
        for (int f = 0; f < c->shapes[e->shape[b]].nv; ++f)
            if (!((e->occ[b] >> f) & 1)) { fb_[nfree] = b; ff_[nfree] = f; nfree++; }
    int gsize = c->n_ground + nfree * c->n_offsets, a = 0;
    e->n_valid = 0;
    for (int grp = 0; grp < c->n_groups; ++grp) {
        const orc_shape* sn = &c->shapes[c->group_shape[grp]];
        int fc = c->group_face[grp];
        for (int slot = 0; slot < gsize && a < c->a_max; ++slot, ++a) {
            orc_cand* cd = &e->cand[a];
            frame2 f1;
            if (slot < c->n_ground) {
                cd->tb = -1; cd->tf = 0; cd->ox = c->x_ground[slot];
                f1.cx = 0; f1.cz = 0; f1.tx = 1; f1.tz = 0; f1.nx = 0; f1.nz = 1;
            } else {
                int k = (slot - c->n_ground) / c->n_offsets;
                cd->tb = fb_[k]; cd->tf = ff_[k]; cd->ox = c->offsets[(slot - c->n_ground) % c->n_offsets];
                const orc_shape* st = &c->shapes[e->shape[cd->tb]];
                f1 = edge_frame(e->verts[cd->tb][st->fa[cd->tf]], e->verts[cd->tb][st->fb[cd->tf]]);
            }
            cd->sh = c->group_shape[grp]; cd->fc = fc;
            align_place(&f1, sn->fcx[fc], sn->fcz[fc], sn->fnx[fc], sn->fnz[fc], cd->ox, 0.0, cd->pose);
            int inb = 1;
            const double eps = 1e-6;
            memset(cd->verts, 0, sizeof(cd->verts));
            for (int i = 0; i < sn->nv; ++i) {
                double rx, rz;
                rot2(sn->vx[i], sn->vz[i], cd->pose[2], cd->pose[3], &rx, &rz);
                double wx = cd->pose[0] + rx, wz = cd->pose[1] + rz;
                cd->verts[i][0] = wx; cd->verts[i][1] = wz;
                if (wx < c->xlim[0] - eps || wx > c->xlim[1] + eps || wz < c->ylim[0] - eps || wz > c->ylim[1] + eps) inb = 0;
                if (wz < -eps) inb = 0;
            }
            cd->inb = (uint8_t)inb;
            raster_block(c, sn, (const double(*)[2])cd->verts, cd->bits);
            int overlap = 0;
            double lin = 0.0;
            for (int r = 0; r < IMG; ++r) {
                uint64_t m = cd->bits[r];
                if (m & (e->state_bits[r] | c->obstacle_bits[r])) overlap = 1;
                while (m) { int q = __builtin_ctzll(m); lin += (double)c->reward_map[r * IMG + q]; m &= m - 1; }
            }
            cd->lin = (float)lin;
            cd->mask = (uint8_t)(inb && !overlap);
            e->n_valid += cd->mask;
        }
    }
    e->n_cand = a;
    e->needs_reset = e->n_valid == 0;
    if (e->f32) {                       /* torch.Tensor(render_blocks_2d(...)) for every raw candidate + the state */
        for (int i = 0; i <= a; ++i) {
            const uint64_t* b = i < a ? e->cand[i].bits : e->state_bits;
            float* img = e->f32 + (size_t)i * IMG * IMG;
            for (int r = 0; r < IMG; ++r) {
                uint64_t m = b[r];
                if (!m) { memset(img + r * IMG, 0, IMG * sizeof(float)); continue; }
                for (int q = 0; q < IMG; ++q) img[r * IMG + q] = (float)((m >> q) & 1ull);
            }
        }
    }
}

/* ---- public API ---- */
orc_env* orc_create(const orc_cfg* cfg) {
    orc_env* e = (orc_env*)calloc(1, sizeof(orc_env));
    if (!e) return 0;
    e->c = *cfg;
    e->cand = (orc_cand*)calloc((size_t)cfg->a_max, sizeof(orc_cand));
    e->tab = (double*)malloc(sizeof(double) * (3 * MAXK + 2) * (4 * MAXIF + 3));
    reset_state(e);
    refresh(e);
    return e;
}
void orc_destroy(orc_env* e) { if (e) { free(e->cand); free(e->tab); free(e->f32); free(e); } }
/* Also materialise the f32 rasters every lock-step (the unit of work of the benchmark). */
int orc_enable_f32(orc_env* e) {
    if (!e->f32) e->f32 = (float*)malloc(sizeof(float) * (size_t)(e->c.a_max + 1) * IMG * IMG);
    return e->f32 != 0;
}
const float* orc_f32(const orc_env* e) { return e->f32; }
void orc_reset(orc_env* e) { reset_state(e); e->draw_counter = 0; refresh(e); }
const orc_cand* orc_candidates(const orc_env* e, int32_t* n_cand, int32_t* n_valid) { *n_cand = e->n_cand; *n_valid = e->n_valid; return e->cand; }
const uint64_t* orc_state_bits(const orc_env* e) { return e->state_bits; }
long orc_total_pivots(const orc_env* e) { return e->total_pivots; }
/* current assembly: shape ids and poses (x, z, cos, sin) of the placed blocks */
int orc_blocks(const orc_env* e, int32_t* shape, double* pose) {
    for (int b = 0; b < e->nb; ++b) { shape[b] = e->shape[b]; memcpy(pose + 4 * b, e->pose[b], 4 * sizeof(double)); }
    return e->nb;
}

/* out[a] = is_action_stable_rbe of candidate a for the valid (mask) candidates of the current state, 0 otherwise. */
int orc_candidate_stability(orc_env* e, uint8_t* out) {
    int n = 0;
    for (int a = 0; a < e->n_cand; ++a) { out[a] = e->cand[a].mask ? (uint8_t)action_stable(e, a) : 0; n += e->cand[a].mask; }
    return n;
}

/* One lock-step of the protocol of DESIGN.md: place a uniformly drawn valid candidate (or reset-only), both
 * stability variants, reward / termination, auto-reset, candidates of the new state. */
void orc_lockstep(orc_env* e, uint64_t seed, int32_t env_id, orc_out* out) {
    memset(out, 0, sizeof(*out));
    const orc_cfg* c = &e->c;
    if (e->needs_reset) {
        reset_state(e);
    } else {
        uint64_t r = splitmix64(splitmix64(((seed & 0xFFFFFFFFull) << 32) | (uint32_t)env_id) ^ e->draw_counter);
        e->draw_counter++;
        int rank = (int)(r % (uint64_t)e->n_valid), a = 0;
        for (int i = 0, seen = 0; i < e->n_cand; ++i) if (e->cand[i].mask) { if (seen == rank) { a = i; break; } seen++; }
        const orc_cand* cd = &e->cand[a];
        int nb = e->nb;
        e->shape[nb] = cd->sh;
        memcpy(e->pose[nb], cd->pose, sizeof(cd->pose));
        memcpy(e->verts[nb], cd->verts, sizeof(cd->verts));
        for (int r2 = 0; r2 < IMG; ++r2) e->state_bits[r2] |= cd->bits[r2];
        e->occ[nb] = (uint8_t)(1u << cd->fc);
        if (cd->tb >= 0) e->occ[cd->tb] |= (uint8_t)(1u << cd->tf);
        e->nb = nb + 1;
        const orc_shape* sh = &c->shapes[cd->sh];
        double x0 = 1e300, x1 = -1e300, z0 = 1e300, z1 = -1e300;
        for (int i = 0; i < sh->nv; ++i) {
            x0 = fmin(x0, cd->verts[i][0]); x1 = fmax(x1, cd->verts[i][0]);
            z0 = fmin(z0, cd->verts[i][1]); z1 = fmax(z1, cd->verts[i][1]);
        }
        double cx = (x0 + x1) * 0.5, cz = (z0 + z1) * 0.5, hx = (x1 - x0) * 0.5, hz = (z1 - z0) * 0.5, hy = sh->depth * 0.5;
        /* gym_env.py:163-169 removes from the list it iterates over: the open target after each reached one is skipped */
        for (int t = 0, skip = 0; t < c->n_targets; ++t) {
            if (!((e->targets_left >> t) & 1u)) continue;
            if (skip) { skip = 0; continue; }
            if (fabs(c->targets[t][0] - cx) < hx + 1e-6 && fabs(c->targets[t][1]) < hy + 1e-6 && fabs(c->targets[t][2] - cz) < hz + 1e-6) {
                e->targets_left &= ~(1u << t);
                skip = 1;
            }
        }
        int n_reached = c->n_targets - __builtin_popcount(e->targets_left);
        append_interfaces(e, nb);
        int sf, su;
        long p0 = e->total_pivots;
        rbe_both(e, &sf, &su);
        int all = e->targets_left == 0;
        /* a state that fills the K = (max_steps or MAXK) block slots of the device layout is truncated too */
        int terminated = !sf || all, truncated = (c->max_steps > 0 && e->nb >= c->max_steps) || e->nb >= MAXK;
        out->valid_step = 1; out->action_index = a; out->stable_frozen = sf; out->stable_unfrozen = su;
        out->terminated = terminated; out->truncated = truncated; out->done = terminated || truncated;
        out->n_blocks = e->nb; out->n_reached = n_reached; out->lp_pivots = (int32_t)(e->total_pivots - p0);
        out->reward = !sf ? -1.0 : (all ? (double)n_reached : (double)(-1 + n_reached));
        float base = cd->lin;
        out->lin_reward = su ? (double)base : (sf ? (double)(base / 100.f) : 0.0);
        memcpy(out->pose, cd->pose, sizeof(out->pose));
        if (out->done) reset_state(e);
    }
    refresh(e);
    out->no_actions = e->needs_reset;
}

/* Timed loop for the CPU baseline: n lock-steps, returns the number of real env-steps. */
long orc_run(orc_env* e, uint64_t seed, int32_t env_id, long n_locksteps) {
    orc_out o;
    long steps = 0;
    for (long i = 0; i < n_locksteps; ++i) { orc_lockstep(e, seed, env_id, &o); steps += o.valid_step; }
    return steps;
}

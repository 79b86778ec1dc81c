// Device-side helpers shared by the gfx950 kernels.
//
// ARITHMETIC CONTRACT: every geometric quantity is computed with
// separately rounded IEEE-754 binary64 operations in exactly the order written
// here; the translation units are compiled with -ffp-contract=off so hipcc
// cannot fuse a*b+c.  The CPU oracle (oracle/geometry.py, oracle/shapes.py,
// oracle/raster.py, oracle/rbe.py) states the same order, which is what makes
// poses, masks and rasters comparable bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bridges_hip.h"

#define WAVE 64
#define MAXV BRIDGES_MAX_VERTS
#define MAXK BRIDGES_MAX_BLOCKS
#define MAXIF BRIDGES_MAX_INTERFACES
#define IMG BRIDGES_IMG

// Diagnostic switches (bridges_task.debug) exist only in builds made with -DBRIDGES_DIAG (tools/build_diag.sh); the
// product library compiles every such branch away and bridges_env_create refuses a non-zero debug word.
#ifdef BRIDGES_DIAG
#define DIAG(c, bit) (((c).debug & (bit)) != 0)
#else
#define DIAG(c, bit) false
#endif

namespace bridges {

struct Frame2 {            // face frame: centre, tangent (x-axis), outward normal
    double cx, cz, tx, tz, nx, nz;
};

// assembly_env.py:118-124 on a directed edge va->vb (oracle/shapes.py edge_frame)
__device__ __forceinline__ Frame2 edge_frame(double ax, double az, double bx, double bz) {
    Frame2 f;
    f.cx = (ax + bx) * 0.5;
    f.cz = (az + bz) * 0.5;
    double dx = bx - ax;
    double dz = bz - az;
    double L = sqrt(dx * dx + dz * dz);
    f.tx = dx / L;
    f.tz = dz / L;
    f.nx = -f.tz;
    f.nz = f.tx;
    return f;
}

__device__ __forceinline__ void rot2(double vx, double vz, double c, double s, double& ox, double& oz) {
    ox = vx * c + vz * s;
    oz = vz * c - vx * s;
}

// geometry.py:39-50 align_frames_2d + gym_env.py:204-216 create_block (oracle/geometry.py align)
__device__ __forceinline__ void align_place(const Frame2& f1, double c2x, double c2z, double n2x, double n2z,
                                            double ox, double oy, double& px, double& pz, double& c, double& s) {
    double dot = f1.nx * n2x + f1.nz * n2z;
    c = -dot;
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    double cy = f1.nz * n2x - f1.nx * n2z;
    s = (cy + 1e-6 >= 0) ? fabs(cy) : -fabs(cy);
    double r2x, r2z;
    rot2(c2x, c2z, c, s, r2x, r2z);
    px = ((f1.cx + ox * f1.tx) + oy * f1.nx) - r2x;
    pz = ((f1.cz + ox * f1.tz) + oy * f1.nz) - r2z;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, WAVE); }

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, WAVE);
    hi = __shfl(hi, src, WAVE);
    return ((uint64_t)hi << 32) | lo;
}

// ---- wave-wide reductions on the DPP cross-lane path (VALU speed; __shfl_xor lowers to ds_bpermute, which costs
// ~100+ cycles per step and made every simplex pivot ~10k cycles).  gfx9 DPP controls:
#define DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142   // lane 15 of each row -> next row   (row_mask 0xA)
#define DPP_ROW_BCAST31 0x143   // lane 31 -> rows 2,3               (row_mask 0xC)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int x) {
    return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double old, double x) {
    int lo = dpp_i<CTRL, ROW_MASK>(__double2loint(old), __double2loint(x));
    int hi = dpp_i<CTRL, ROW_MASK>(__double2hiint(old), __double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane_uniform) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_uniform);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_uniform);
    return __hiloint2double(hi, lo);
}
#define WAVE_REDUCE_D(OP, IDENT_OLD)                                                   \
    v = OP(v, dpp_d<DPP_QUAD_XOR1, 0xF>(v, v));                                        \
    v = OP(v, dpp_d<DPP_QUAD_XOR2, 0xF>(v, v));                                        \
    v = OP(v, dpp_d<DPP_ROW_HALF_MIRROR, 0xF>(v, v));                                  \
    v = OP(v, dpp_d<DPP_ROW_MIRROR, 0xF>(v, v));                                       \
    v = OP(v, dpp_d<DPP_ROW_BCAST15, 0xA>(IDENT_OLD, v));                              \
    v = OP(v, dpp_d<DPP_ROW_BCAST31, 0xC>(IDENT_OLD, v));                              \
    return readlane_d(v, 63);

__device__ __forceinline__ double add_d_(double a, double b) { return a + b; }
// v_min_f64 / v_max_f64 as they are: fmin / fmax make the compiler quiet both operands first (two v_max_f64 x, x per call --
// the operands come out of integer DPP moves, so it cannot know they are not signalling NaNs), which tripled every step of
// the reductions the simplex pivots on.  The reduced values are never NaN.  The s_nop covers the two wait states a DPP
// read of the result needs (the next reduction step): the hazard pass does not look into inline assembly.
__device__ __forceinline__ double min_raw_d(double a, double b) { double r; asm volatile("v_min_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_raw_d(double a, double b) { double r; asm volatile("v_max_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double wave_sum_d(double v) { WAVE_REDUCE_D(add_d_, 0.0) }
__device__ __forceinline__ double wave_min_d(double v) { WAVE_REDUCE_D(min_raw_d, v) }
__device__ __forceinline__ double wave_max_d(double v) { WAVE_REDUCE_D(max_raw_d, v) }
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<DPP_QUAD_XOR1, 0xF>(v, v));
    v = min(v, dpp_i<DPP_QUAD_XOR2, 0xF>(v, v));
    v = min(v, dpp_i<DPP_ROW_HALF_MIRROR, 0xF>(v, v));
    v = min(v, dpp_i<DPP_ROW_MIRROR, 0xF>(v, v));
    v = min(v, dpp_i<DPP_ROW_BCAST15, 0xA>(v, v));
    v = min(v, dpp_i<DPP_ROW_BCAST31, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Small constant tables of a task, resident in device memory.
struct TaskTable {
    bridges_shape shapes[8];
    double x_ground[32];
    double offsets[8];
    double grid_x[IMG];
    double grid_y[IMG];
};

// Everything a kernel needs, passed by value (kernarg segment, scalar loads).
struct DevCtx {
    bridges_env_buffers b;
    const TaskTable* tt;
    int32_t* h_total;               // mapped host word: total raw candidates of the last scan (sizes the next raster grid)
    int32_t E, K, max_steps, a_max, n_groups, n_ground, n_offsets, n_targets;
    int32_t debug, env_id_base;     // debug: BRIDGES_DIAG builds only (bit0 skip the LPs, bit1 / bit2 skip the half-plane runs / the f32
                                    // stores of the rasteriser, bit3 per-env phase stamps, bit4 empty candidate-stability grid)
    int32_t n_shapes, img;            // img: image width = height S <= 64 (64 = the reference's default)
    int32_t group_shape[BRIDGES_MAX_GROUPS];
    int32_t group_face[BRIDGES_MAX_GROUPS];
    double mu, density, floor_hw, floor_depth;
    double xlim0, xlim1, ylim0, ylim1;
    double targets[BRIDGES_MAX_TARGETS][3];
    uint64_t seed;
};

enum { F_VALID = 0, F_STABLE_FROZEN, F_STABLE_UNFROZEN, F_TERMINATED, F_TRUNCATED, F_DONE, F_NO_ACTIONS, F_LP_ERROR };
enum { ST_SUM_CAND = 0, ST_SUM_BLOCKS, ST_ENV_STEPS, ST_RESET_ONLY, ST_LP_ERRORS, ST_IF_OVERFLOW, ST_LOCKSTEPS, ST_SUM_VALID, ST_WARM_RESOLVED, ST_CAND_OVERFLOW };

}  // namespace bridges

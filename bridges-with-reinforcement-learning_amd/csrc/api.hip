// extern "C" entry points of libbridges_hip.so (see include/bridges_hip.h).  Unity build: the kernel
// translation units are included here so one hipcc invocation produces the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "env_kernels.hip"
#include "ops_kernels.hip"
#include "dqn_kernels.hip"
#include "mlp_kernels.hip"
#include "conv_kernels.hip"
#include "conv_train_kernels.hip"

using namespace bridges;

static thread_local char g_err[512] = "";

static int fail_hip(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return BRIDGES_E_HIP;
}
static int fail_arg(const char* what) {
    snprintf(g_err, sizeof(g_err), "bad argument: %s", what);
    return BRIDGES_E_ARG;
}
#define HIP_TRY(x)                                        \
    do {                                                  \
        hipError_t e_ = (x);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #x);    \
    } while (0)
#define LAUNCH_CHECK(name)                                       \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) return fail_hip(e_, name);         \
    } while (0)

struct bridges_gate {
    hipEvent_t last;           // completion of the most recent rasteriser launch attached to this gate (or null)
};

struct bridges_env {
    bridges_gate* gate;
    hipEvent_t raster_done;
    DevCtx ctx;
    TaskTable* tt_dev;
    int32_t* h_total;          // pinned: candidate count of the previous lock-step (sizes the raster / expand grids)
    int max_blocks;            // upper bound of a useful grid
    int max_faces;             // most 2-D faces among the task's candidate shapes (picks the rasteriser instantiation)
    // optional per-launch timing of the dominant kernel (k_raster) with HIP events on the launch stream
    hipEvent_t* ev_start;
    hipEvent_t* ev_stop;
    int ev_cap, ev_used;
};

extern "C" {

const char* bridges_last_error(void) { return g_err; }

#ifndef BRIDGES_SRC_HASH
#define BRIDGES_SRC_HASH "unstamped"
#endif
const char* bridges_source_hash(void) { return "BRIDGES_SRC_HASH=" BRIDGES_SRC_HASH; }

int bridges_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bridges_env_create(const bridges_task* t, const bridges_env_buffers* buf, bridges_env** out) {
    if (!t || !buf || !out) return fail_arg("null");
    if (t->n_envs <= 0) return fail_arg("n_envs");
    if (t->max_blocks <= 0 || t->max_blocks > BRIDGES_MAX_BLOCKS) return fail_arg("max_blocks > BRIDGES_MAX_BLOCKS");
    if (t->n_shapes <= 0 || t->n_shapes > 8) return fail_arg("n_shapes");
    if (t->n_groups <= 0 || t->n_groups > BRIDGES_MAX_GROUPS) return fail_arg("n_groups");
    if (t->n_ground < 0 || t->n_ground > 32 || t->n_offsets <= 0 || t->n_offsets > 8) return fail_arg("n_ground/n_offsets");
    if (t->n_targets < 0 || t->n_targets > BRIDGES_MAX_TARGETS) return fail_arg("n_targets");
    if (t->a_max <= 0) return fail_arg("a_max");
    if (t->img_size != 0 && (t->img_size < 2 || t->img_size > BRIDGES_IMG)) return fail_arg("img_size must be 0 (= 64) or 2..64");
#ifndef BRIDGES_DIAG
    if (t->debug != 0) return fail_arg("debug switches exist only in a diagnostic build (-DBRIDGES_DIAG, tools/build_diag.sh)");
#endif
    if (!buf->reward_prefix) return fail_arg("reward_prefix (float64 row prefix sums of reward_map) not given");
    if (buf->lp_ws_stride < (int64_t)BRIDGES_LP_WS_DOUBLES) return fail_arg("lp_ws_stride < BRIDGES_LP_WS_DOUBLES");
    static_assert(BRIDGES_LP_WS_DOUBLES == WARM_WS_DOUBLES, "header and device code disagree on the persistent tableau size");
    static_assert(BRIDGES_LP_SNAP_DOUBLES == WARM_HDR_DOUBLES + WARM_HALF, "header and device code disagree on the snapshot size");
    if (buf->lp_snap && buf->lp_snap_stride < (int64_t)BRIDGES_LP_SNAP_DOUBLES) return fail_arg("lp_snap_stride < BRIDGES_LP_SNAP_DOUBLES");
    for (int g = 0; g < t->n_groups; ++g) {
        if (t->group_shape[g] < 0 || t->group_shape[g] >= t->n_shapes) return fail_arg("group_shape");
        if (t->group_face[g] < 0 || t->group_face[g] >= t->shapes[t->group_shape[g]].nv) return fail_arg("group_face");
    }
    for (int s = 0; s < t->n_shapes; ++s)
        if (t->shapes[s].nv < 3 || t->shapes[s].nv > BRIDGES_MAX_VERTS) return fail_arg("shape nv");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device");
        return BRIDGES_E_NODEV;
    }
    bridges_env* env = new (std::nothrow) bridges_env();
    if (!env) return fail_arg("oom");
    TaskTable* host = new (std::nothrow) TaskTable();
    if (!host) { delete env; return fail_arg("oom"); }
    memset(host, 0, sizeof(TaskTable));
    memcpy(host->shapes, t->shapes, sizeof(bridges_shape) * t->n_shapes);
    memcpy(host->x_ground, t->x_ground, sizeof(double) * t->n_ground);
    memcpy(host->offsets, t->offsets, sizeof(double) * t->n_offsets);
    const int img = t->img_size ? t->img_size : IMG;
    for (int i = 0; i < IMG; ++i) {                      // lanes / rows >= img repeat the last grid value (never inside)
        host->grid_x[i] = t->grid_x[i < img ? i : img - 1];
        host->grid_y[i] = t->grid_y[i < img ? i : img - 1];
    }
    hipError_t e = hipMalloc((void**)&env->tt_dev, sizeof(TaskTable));
    if (e == hipSuccess) e = hipMemcpy(env->tt_dev, host, sizeof(TaskTable), hipMemcpyHostToDevice);
    delete host;
    if (e != hipSuccess) { delete env; return fail_hip(e, "task table upload"); }
    DevCtx& c = env->ctx;
    memset(&c, 0, sizeof(c));
    c.b = *buf;
    c.tt = env->tt_dev;
    c.E = t->n_envs; c.K = t->max_blocks; c.max_steps = t->max_steps; c.a_max = t->a_max;
    c.n_groups = t->n_groups; c.n_ground = t->n_ground; c.n_offsets = t->n_offsets; c.n_targets = t->n_targets;
    memcpy(c.group_shape, t->group_shape, sizeof(c.group_shape));
    memcpy(c.group_face, t->group_face, sizeof(c.group_face));
    c.mu = t->mu; c.density = t->density; c.floor_hw = t->floor_half_width; c.floor_depth = t->floor_depth;
    c.xlim0 = t->xlim[0]; c.xlim1 = t->xlim[1]; c.ylim0 = t->ylim[0]; c.ylim1 = t->ylim[1];
    memcpy(c.targets, t->targets, sizeof(c.targets));
    c.seed = t->seed;
    c.debug = t->debug;
    c.env_id_base = t->env_id_base;
    c.n_shapes = t->n_shapes;
    c.img = img;
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    env->ev_start = env->ev_stop = nullptr;
    env->ev_cap = env->ev_used = 0;
    env->gate = nullptr;
    env->raster_done = nullptr;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    (void)cus;
    // mapped, coherent host word: k_scan stores the candidate count of the lock-step straight into it (no copy command
    // in the stream); the host reads it as a hint for the next grid, so a late value costs time, never correctness
    e = hipHostMalloc((void**)&env->h_total, sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c.h_total, env->h_total, 0);
    if (e != hipSuccess) { (void)hipFree(env->tt_dev); delete env; return fail_hip(e, "hipHostMalloc"); }
    *env->h_total = t->n_envs * 64;        // first guess; replaced after every scan
    env->max_blocks = 1 << 22;
    env->max_faces = 0;
    for (int g = 0; g < t->n_groups; ++g) {
        const int nv = t->shapes[t->group_shape[g]].nv;
        if (nv > env->max_faces) env->max_faces = nv;
    }
    *out = env;
    return BRIDGES_OK;
}

static void free_events(bridges_env* env) {
    for (int i = 0; i < env->ev_cap; ++i) {
        (void)hipEventDestroy(env->ev_start[i]);
        (void)hipEventDestroy(env->ev_stop[i]);
    }
    delete[] env->ev_start;
    delete[] env->ev_stop;
    env->ev_start = env->ev_stop = nullptr;
    env->ev_cap = env->ev_used = 0;
}

int bridges_env_destroy(bridges_env* env) {
    if (!env) return BRIDGES_OK;
    free_events(env);
    if (env->raster_done) (void)hipEventDestroy(env->raster_done);
    (void)hipFree(env->tt_dev);
    (void)hipHostFree(env->h_total);
    delete env;
    return BRIDGES_OK;
}

int bridges_gate_create(bridges_gate** out) {
    if (!out) return fail_arg("gate_create");
    bridges_gate* g = new (std::nothrow) bridges_gate();
    if (!g) return fail_arg("oom");
    g->last = nullptr;
    *out = g;
    return BRIDGES_OK;
}

int bridges_gate_destroy(bridges_gate* gate) {
    delete gate;
    return BRIDGES_OK;
}

int bridges_env_set_gate(bridges_env* env, bridges_gate* gate) {
    if (!env) return fail_arg("set_gate");
    if (gate && !env->raster_done) HIP_TRY(hipEventCreateWithFlags(&env->raster_done, hipEventDisableTiming));
    env->gate = gate;
    return BRIDGES_OK;
}

int bridges_env_timing_begin(bridges_env* env, int32_t max_launches) {
    if (!env || max_launches <= 0 || max_launches > (1 << 16)) return fail_arg("timing_begin");
    free_events(env);
    env->ev_start = new (std::nothrow) hipEvent_t[max_launches];
    env->ev_stop = new (std::nothrow) hipEvent_t[max_launches];
    if (!env->ev_start || !env->ev_stop) {
        delete[] env->ev_start;
        delete[] env->ev_stop;
        env->ev_start = env->ev_stop = nullptr;
        return fail_arg("oom");
    }
    for (int i = 0; i < max_launches; ++i) {
        HIP_TRY(hipEventCreate(&env->ev_start[i]));
        HIP_TRY(hipEventCreate(&env->ev_stop[i]));
        env->ev_cap = i + 1;
    }
    env->ev_used = 0;
    return BRIDGES_OK;
}

int bridges_env_timing_end(bridges_env* env, double* raster_ms_total, int32_t* n_launches) {
    if (!env || !raster_ms_total || !n_launches) return fail_arg("timing_end");
    double total = 0.0;
    for (int i = 0; i < env->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(env->ev_stop[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, env->ev_start[i], env->ev_stop[i]));
        total += ms;
    }
    *raster_ms_total = total;
    *n_launches = env->ev_used;
    free_events(env);
    return BRIDGES_OK;
}

// Unused dynamic LDS per rasteriser workgroup (4 waves): caps how many of them a CU holds (160 KB of LDS) and so leaves
// registers / wave slots to the latency-bound task kernels of the other env groups.  0 = no cap: 8 workgroups per CU fill every
// wave slot (tools/raster_occupancy.sh measures the alternatives).
#ifndef RASTER_DYN_LDS
#define RASTER_DYN_LDS 0
#endif
static int refresh(bridges_env* env, hipStream_t s, int after_step) {
    const DevCtx& c = env->ctx;
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_THREADS), 0, s, c, after_step);
    LAUNCH_CHECK("k_scan");
    hipLaunchKernelGGL(k_enumerate, dim3(c.E), dim3(WAVE), 0, s, c);
    LAUNCH_CHECK("k_enumerate");
    // grids from the previous lock-step's candidate count (+3 %); the kernels grid-stride, so a stale or low
    // estimate costs time, never correctness.  k_scan stores the fresh count into the mapped host word for the next call.
    long long est = (long long)(*(volatile int32_t*)env->h_total);
    if (est < c.E) est = c.E;
    est += est / 32 + 64;
    const long long items_est = est + c.E;            // one wave per image (candidates + state rasters)
    long long rblocks = (items_est + 3) / 4;
    if (rblocks > env->max_blocks) rblocks = env->max_blocks;
    const bool timed = env->ev_cap > 0 && env->ev_used < env->ev_cap;
    if (env->gate && env->gate->last) HIP_TRY(hipStreamWaitEvent(s, env->gate->last, 0));
    if (timed) HIP_TRY(hipEventRecord(env->ev_start[env->ev_used], s));
    if (env->max_faces <= 4) hipLaunchKernelGGL(k_raster<4>, dim3((unsigned)rblocks), dim3(256), RASTER_DYN_LDS, s, c);
    else hipLaunchKernelGGL(k_raster<MAXV>, dim3((unsigned)rblocks), dim3(256), RASTER_DYN_LDS, s, c);
    LAUNCH_CHECK("k_raster");
    if (timed) { HIP_TRY(hipEventRecord(env->ev_stop[env->ev_used], s)); env->ev_used++; }
    if (env->gate) {
        HIP_TRY(hipEventRecord(env->raster_done, s));
        env->gate->last = env->raster_done;
    }
    hipLaunchKernelGGL(k_select, dim3(c.E), dim3(WAVE), 0, s, c, 0);
    LAUNCH_CHECK("k_select");
    return BRIDGES_OK;
}

int bridges_env_reset(bridges_env* env, void* stream) {
    if (!env) return fail_arg("null env");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_reset, dim3(env->ctx.E), dim3(WAVE), 0, s, env->ctx);
    LAUNCH_CHECK("k_reset");
    return refresh(env, s, 0);
}

int bridges_env_step(bridges_env* env, void* stream) {
    if (!env) return fail_arg("null env");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_step, dim3(env->ctx.E), dim3(WAVE), 0, s, env->ctx);
    LAUNCH_CHECK("k_step");
    return refresh(env, s, 1);
}

int bridges_env_refresh(bridges_env* env, void* stream) {
    if (!env) return fail_arg("null env");
    return refresh(env, (hipStream_t)stream, 0);
}

#ifndef CS_TAB_SMALL
#define CS_TAB_SMALL 768     // 6 KiB: with carriers, candidates on up to ~4 placed blocks; 9.5 KB of LDS and <= 128 VGPRs per wave: 16 waves per CU
#endif
#ifndef CS_COLS_SMALL
#define CS_COLS_SMALL 92
#endif
#define CS_TAB_LARGE 4096
int bridges_env_candidate_stability(bridges_env* env, void* stream) {
    if (!env) return fail_arg("null env");
    const DevCtx& c = env->ctx;
    if (!c.b.cand_stable || !c.b.cand_queue || !c.b.cand_counters || !c.b.cand_ws) return fail_arg("cand_stable / cand_queue / cand_counters / cand_ws not given");
    if (c.b.cand_ws_stride < (int64_t)(3 * c.K + 2) * (4 * BRIDGES_MAX_INTERFACES + 3)) return fail_arg("cand_ws_stride");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(c.b.cand_counters, 0, 2 * sizeof(int32_t), s));
    // one wave per raw candidate (masked-out ones leave at once); grid from the last known candidate count, grid-stride beyond
    long long est = (long long)(*(volatile int32_t*)env->h_total);
    if (est < c.E) est = c.E;
    est += est / 32 + 64;
    if (est > env->max_blocks) est = env->max_blocks;
    hipLaunchKernelGGL((k_candidate_stability<CS_TAB_SMALL, CS_COLS_SMALL, false>), dim3((unsigned)est), dim3(WAVE), 0, s, c);
    LAUNCH_CHECK("k_candidate_stability");
    const int drain = BRIDGES_CAND_WS_SLOTS;          // one cand_ws slot per workgroup
    hipLaunchKernelGGL((k_candidate_stability<CS_TAB_LARGE, LP_MAX_COLS, true>), dim3(drain), dim3(WAVE), 0, s, c);
    LAUNCH_CHECK("k_candidate_stability (queue)");
    return BRIDGES_OK;
}

int bridges_env_select_random(bridges_env* env, void* stream) {
    if (!env) return fail_arg("null env");
    hipLaunchKernelGGL(k_select, dim3(env->ctx.E), dim3(WAVE), 0, (hipStream_t)stream, env->ctx, 1);
    LAUNCH_CHECK("k_select");
    return BRIDGES_OK;
}

int bridges_env_lockstep_random(bridges_env* env, void* stream) {
    int rc = bridges_env_select_random(env, stream);
    if (rc != BRIDGES_OK) return rc;
    return bridges_env_step(env, stream);
}

int bridges_shapes_upload(const bridges_shape* host, int32_t n, bridges_shape** out_dev) {
    if (!host || n <= 0 || !out_dev) return fail_arg("shapes_upload");
    HIP_TRY(hipMalloc((void**)out_dev, sizeof(bridges_shape) * n));
    HIP_TRY(hipMemcpy(*out_dev, host, sizeof(bridges_shape) * n, hipMemcpyHostToDevice));
    return BRIDGES_OK;
}

int bridges_shapes_free(bridges_shape* dev) {
    if (dev) HIP_TRY(hipFree(dev));
    return BRIDGES_OK;
}

int bridges_place(const bridges_shape* shapes_dev, int32_t n, const double* frame1, const int32_t* shape_id,
                  const int32_t* face, const double* ox, const double* oy, double* pose, double* verts, void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_place");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_place, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shapes_dev, n, frame1,
                       shape_id, face, ox, oy, pose, verts);
    LAUNCH_CHECK("k_place");
    return BRIDGES_OK;
}

int bridges_create_block(const bridges_shape* shapes_dev, int32_t n, const double* target_verts,
                         const int32_t* target_shape, const int32_t* target_face, const int32_t* shape_id,
                         const int32_t* face, const double* ox, const double* oy, double* pose, double* verts,
                         double* target_frame_out, void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_create_block");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_create_block, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shapes_dev, n,
                       target_verts, target_shape, target_face, shape_id, face, ox, oy, pose, verts, target_frame_out);
    LAUNCH_CHECK("k_create_block");
    return BRIDGES_OK;
}

int bridges_pose_block(const bridges_shape* shapes_dev, int32_t n, const int32_t* shape_id, const double* pose,
                       double* verts, void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_pose_block");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_pose_block, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shapes_dev, n, shape_id,
                       pose, verts);
    LAUNCH_CHECK("k_pose_block");
    return BRIDGES_OK;
}

int bridges_face_frames(const bridges_shape* shapes_dev, int32_t n, const int32_t* shape_id, const double* verts,
                        double* frames, void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_face_frames");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_face_frames, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shapes_dev, n, shape_id,
                       verts, frames);
    LAUNCH_CHECK("k_face_frames");
    return BRIDGES_OK;
}

int bridges_contains_points(const bridges_shape* shapes_dev, int32_t shape_id, const double* verts, int32_t n,
                            const double* points, uint8_t* inside, void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_contains_points");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_contains_points, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shapes_dev, shape_id,
                       verts, n, points, inside);
    LAUNCH_CHECK("k_contains_points");
    return BRIDGES_OK;
}

static int grid_for_waves(int64_t n_items) {
    int64_t blocks = (n_items + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int bridges_render_blocks(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                          const double* grid_x, int32_t W, const double* grid_y, int32_t H, uint8_t* out, void* stream) {
    if (!shapes_dev || n < 0 || W < 1 || H < 1 || !grid_x || !grid_y || !out || (n > 0 && (!verts || !shape_id)))
        return fail_arg("bridges_render_blocks");
    const int64_t px = (int64_t)W * H;
    hipLaunchKernelGGL(k_render_blocks, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, (hipStream_t)stream, shapes_dev, n, verts,
                       shape_id, grid_x, W, grid_y, H, out);
    LAUNCH_CHECK("k_render_blocks");
    return BRIDGES_OK;
}

int bridges_raster_sized(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                         const double* grid_x, const double* grid_y, int32_t size, uint64_t* bits, float* img,
                         void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_raster");
    if (size < 2 || size > IMG) return fail_arg("bridges_raster: image size must be 2..64");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_raster_generic, dim3(grid_for_waves(n)), dim3(256), 0, (hipStream_t)stream, shapes_dev, n,
                       verts, shape_id, grid_x, grid_y, (int)size, bits, img);
    LAUNCH_CHECK("k_raster_generic");
    return BRIDGES_OK;
}

int bridges_raster(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                   const double* grid_x, const double* grid_y, uint64_t* bits, float* img, void* stream) {
    return bridges_raster_sized(shapes_dev, n, verts, shape_id, grid_x, grid_y, IMG, bits, img, stream);
}

int bridges_action_features(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                            const double* grid_x, const double* grid_y, int32_t size, double xlim0, double xlim1, double ylim0,
                            double ylim1, const uint64_t* state_bits, const uint64_t* obstacle_bits, const double* reward_prefix,
                            uint64_t* bits, float* img, uint8_t* mask, float* lin_reward, void* stream) {
    if (n < 0 || !shapes_dev || !verts || !shape_id || !grid_x || !grid_y) return fail_arg("bridges_action_features");
    if (size < 2 || size > IMG) return fail_arg("bridges_action_features: image size must be 2..64");
    if (lin_reward && !reward_prefix) return fail_arg("bridges_action_features: lin_reward needs reward_prefix");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_action_features, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, shapes_dev, n, verts,
                       shape_id, grid_x, grid_y, (int)size, xlim0, xlim1, ylim0, ylim1, state_bits, obstacle_bits, reward_prefix, bits,
                       img, mask, lin_reward);
    LAUNCH_CHECK("k_action_features");
    return BRIDGES_OK;
}

int bridges_bits_or(int32_t n_groups, const int32_t* ranges, const uint64_t* bits, uint64_t* out, void* stream) {
    if (n_groups < 0) return fail_arg("bridges_bits_or");
    if (n_groups == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_bits_or, dim3(n_groups), dim3(WAVE), 0, (hipStream_t)stream, n_groups, ranges, bits, out);
    LAUNCH_CHECK("k_bits_or");
    return BRIDGES_OK;
}

int bridges_bits_to_f32(int32_t n, const uint64_t* bits, float* img, void* stream) {
    if (n < 0) return fail_arg("bridges_bits_to_f32");
    if (n == 0) return BRIDGES_OK;
    // one short-lived wave per image, dispatched in image order (the store structure of k_raster)
    hipLaunchKernelGGL(k_bits_to_f32, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n, bits, img);
    LAUNCH_CHECK("k_bits_to_f32");
    return BRIDGES_OK;
}

static int stability_launch(const bridges_shape* shapes_dev, int32_t n, int32_t K, const double* pose, const double* verts,
                            const int32_t* shape_id, const int32_t* n_blocks, const uint32_t* fixed_mask, double mu,
                            double density, double floor_half_width, double floor_depth, uint8_t* stable, double* info,
                            double* lp_ws, int64_t lp_ws_stride, double tension_tol, double* forces, void* stream) {
    if (n < 0 || !shapes_dev) return fail_arg("bridges_stability");
    if (K <= 0 || K > BRIDGES_MAX_BLOCKS) return fail_arg("K > BRIDGES_MAX_BLOCKS");
    if (lp_ws_stride < 9 * BRIDGES_MAX_INTERFACES + (int64_t)(3 * K + 2) * (4 * BRIDGES_MAX_INTERFACES + 3))
        return fail_arg("lp_ws_stride");
    if (n == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_stability, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, shapes_dev, n, K, pose, verts, shape_id,
                       n_blocks, fixed_mask, mu, density, floor_half_width, floor_depth, stable, info, lp_ws, lp_ws_stride,
                       tension_tol, forces);
    LAUNCH_CHECK("k_stability");
    return BRIDGES_OK;
}

int bridges_stability(const bridges_shape* shapes_dev, int32_t n, int32_t K, const double* pose, const double* verts,
                      const int32_t* shape_id, const int32_t* n_blocks, const uint32_t* fixed_mask, double mu,
                      double density, double floor_half_width, double floor_depth, uint8_t* stable, double* info,
                      double* lp_ws, int64_t lp_ws_stride, void* stream) {
    return stability_launch(shapes_dev, n, K, pose, verts, shape_id, n_blocks, fixed_mask, mu, density, floor_half_width,
                            floor_depth, stable, info, lp_ws, lp_ws_stride, 0.0, nullptr, stream);
}

int bridges_stability_penalty(const bridges_shape* shapes_dev, int32_t n, int32_t K, const double* pose, const double* verts,
                              const int32_t* shape_id, const int32_t* n_blocks, const uint32_t* fixed_mask, double mu,
                              double density, double floor_half_width, double floor_depth, double tension_tol,
                              uint8_t* stable, double* info, double* forces, double* lp_ws, int64_t lp_ws_stride,
                              void* stream) {
    if (!(tension_tol >= 0.0)) return fail_arg("tension_tol");
    return stability_launch(shapes_dev, n, K, pose, verts, shape_id, n_blocks, fixed_mask, mu, density, floor_half_width,
                            floor_depth, stable, info, lp_ws, lp_ws_stride, tension_tol, forces, stream);
}

#ifdef LP_PROFILE
int bridges_debug_lp_profile(unsigned long long* out8, int reset) {
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lp_prof), 8 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lp_prof), z, sizeof(z)));
    }
    return BRIDGES_OK;
}
#endif

int bridges_soft_update(float* target, const float* policy, int64_t n, float tau, float one_minus_tau, void* stream) {
    if (n < 0) return fail_arg("bridges_soft_update");
    if (n == 0) return BRIDGES_OK;
    if ((((uintptr_t)target) | ((uintptr_t)policy)) & 15) return fail_arg("soft_update pointers must be 16-byte aligned");
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_soft_update, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, target, policy, n, tau, one_minus_tau);
    LAUNCH_CHECK("k_soft_update");
    return BRIDGES_OK;
}

int bridges_td_target(int32_t n_trans, const int32_t* seg_lo, const int32_t* seg_hi, const float* next_q, const float* next_sf,
                      int64_t next_sf_row_stride, const float* action_raster, const float* lin_reward,
                      const uint8_t* done, float gamma, int32_t sf_dim, float* q_target, float* sf_target,
                      int32_t* argmax_row, void* stream) {
    if (n_trans < 0 || sf_dim < 0 || (n_trans > 0 && (!seg_lo || !seg_hi))) return fail_arg("bridges_td_target");
    if (n_trans == 0) return BRIDGES_OK;
    if (sf_dim > 0 && ((next_sf_row_stride & 3) || (sf_dim & 3) ||
                       ((((uintptr_t)next_sf) | ((uintptr_t)action_raster) | ((uintptr_t)sf_target)) & 15)))
        return fail_arg("td_target: sf rows must be 16-byte aligned");
    hipLaunchKernelGGL(k_td_target, dim3(n_trans), dim3(256), 0, (hipStream_t)stream, n_trans, seg_lo, seg_hi, next_q,
                       next_sf, next_sf_row_stride, action_raster, lin_reward, done, gamma, sf_dim, q_target, sf_target,
                       argmax_row);
    LAUNCH_CHECK("k_td_target");
    return BRIDGES_OK;
}

int bridges_bits_linear(int32_t n_rows, const uint64_t* bits, const int64_t* bits_row, const float* wt, int32_t d,
                        const float* base, const int64_t* base_row, float* out, void* stream) {
    if (n_rows < 0 || d <= 0 || (d & 3) || !bits || !wt || !out) return fail_arg("bridges_bits_linear");
    if ((((uintptr_t)wt) | ((uintptr_t)out) | ((uintptr_t)base)) & 15) return fail_arg("bits_linear: rows must be 16-byte aligned");
    if (n_rows == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_bits_linear, dim3(grid_for_waves(n_rows)), dim3(256), 0, (hipStream_t)stream, n_rows, bits,
                       bits_row, wt, d, base, base_row, out);
    LAUNCH_CHECK("k_bits_linear");
    return BRIDGES_OK;
}

int bridges_eps_greedy_select(int32_t E, int32_t n_rows, const int32_t* seg_lo, const int32_t* seg_hi, const float* q, const float* join,
                              const float* u, float eps, int32_t greedy, const int64_t* idx, const int32_t* cand_offset,
                              const int32_t* rep, int64_t* sel_compact, int32_t* sel_index, float* q_sel, float* explore_w, void* stream) {
    if (E < 1 || n_rows < 1 || !seg_lo || !seg_hi || !q || !join || !u || !idx || !cand_offset || !sel_compact || !sel_index || !q_sel || !explore_w)
        return fail_arg("bridges_eps_greedy_select");
    hipLaunchKernelGGL(k_eps_greedy_select, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, (hipStream_t)stream, E, n_rows, seg_lo, seg_hi, q, join, u, eps,
                       (int)greedy, idx, cand_offset, rep, sel_compact, sel_index, q_sel, explore_w);
    LAUNCH_CHECK("k_eps_greedy_select");
    return BRIDGES_OK;
}

int bridges_record_state(int32_t E, int32_t K, const int32_t* n_blocks, const int32_t* blk_shape, const double* blk_pose,
                         const uint8_t* blk_occ, const uint8_t* step_flags, const int64_t* sel_row, const int32_t* cand_desc,
                         const double* cand_pose, double* rec, void* stream) {
    if (E < 0 || K < 1 || K > BRIDGES_REC_K || !n_blocks || !blk_shape || !blk_pose || !blk_occ || !step_flags || !sel_row ||
        !cand_desc || !cand_pose || !rec)
        return fail_arg("bridges_record_state");
    if (E == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_record_state, dim3((unsigned)E), dim3(64), 0, (hipStream_t)stream, E, K, n_blocks, blk_shape, blk_pose, blk_occ,
                       step_flags, sel_row, cand_desc, cand_pose, rec);
    LAUNCH_CHECK("k_record_state");
    return BRIDGES_OK;
}

int bridges_record_result(int32_t E, const float* reward, const float* lin_reward, const uint8_t* step_flags, double* rec,
                          uint8_t* valid, void* stream) {
    if (E < 0 || !reward || !lin_reward || !step_flags || !rec || !valid) return fail_arg("bridges_record_result");
    if (E == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_record_result, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, E, reward, lin_reward,
                       step_flags, rec, valid);
    LAUNCH_CHECK("k_record_result");
    return BRIDGES_OK;
}

int bridges_replay_unpack(int32_t E, int32_t n_rec, int32_t K, const double* rec, const int32_t* shape_faces, int32_t n_shapes,
                          int32_t n_groups, int32_t n_ground, int32_t n_off, int32_t* n_blocks, int32_t* blk_shape,
                          double* blk_pose, uint8_t* blk_occ, int32_t* n_cand, int32_t* ranges_next, int32_t* ranges_prev,
                          float* lin, float* stable_s, uint8_t* done, uint8_t* stable_n, void* stream) {
    if (E < 0 || n_rec < 1 || n_rec > E || K < 1 || K > BRIDGES_REC_K || !rec || !shape_faces || n_shapes < 1 || n_groups < 0 ||
        n_ground < 0 || n_off < 0 || !n_blocks || !blk_shape || !blk_pose || !blk_occ || !n_cand || !ranges_next ||
        !ranges_prev || !lin || !stable_s || !done || !stable_n)
        return fail_arg("bridges_replay_unpack");
    hipLaunchKernelGGL(k_replay_unpack, dim3((unsigned)E), dim3(64), 0, (hipStream_t)stream, E, n_rec, K, rec, shape_faces, n_shapes, n_groups,
                       n_ground, n_off, n_blocks, blk_shape, blk_pose, blk_occ, n_cand, ranges_next, ranges_prev, lin, stable_s,
                       done, stable_n);
    LAUNCH_CHECK("k_replay_unpack");
    return BRIDGES_OK;
}

int bridges_valid_rows(int32_t E, const int32_t* cand_offset, const int32_t* n_cand, const int32_t* n_valid, const uint8_t* cand_mask,
                       const int32_t* rep, int32_t* seg, int32_t* seg_lo, int32_t* seg_hi, int64_t* idx, int64_t* row_env,
                       int32_t* h_total, void* stream) {
    if (E < 1 || !cand_offset || !n_cand || !n_valid || !cand_mask || !seg || !idx || !row_env || !h_total || (rep && (!seg_lo || !seg_hi)) ||
        ((seg_lo == nullptr) != (seg_hi == nullptr)))
        return fail_arg("bridges_valid_rows");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_valid_scan, dim3(1), dim3(1024), 0, st, E, n_valid, rep, seg, h_total);
    LAUNCH_CHECK("k_valid_scan");
    hipLaunchKernelGGL(k_valid_fill, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, st, E, cand_offset, n_cand, cand_mask, (const int32_t*)seg, rep,
                       seg_lo, seg_hi, idx, row_env);
    LAUNCH_CHECK("k_valid_fill");
    return BRIDGES_OK;
}

int bridges_env_groups(int32_t E, int32_t K, const int32_t* n_blocks, const int32_t* blk_shape, const double* blk_pose,
                       const uint8_t* blk_occ, const uint8_t* flag, uint64_t* hkey, int32_t* rep, void* stream) {
    if (E < 1 || K < 1 || K > 64 || !n_blocks || !blk_shape || !blk_pose || !blk_occ || !hkey || !rep) return fail_arg("bridges_env_groups");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_env_hash, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, st, E, K, n_blocks, blk_shape, blk_pose, blk_occ, flag, hkey);
    LAUNCH_CHECK("k_env_hash");
    hipLaunchKernelGGL(k_env_match, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, st, E, K, n_blocks, blk_shape, blk_pose, blk_occ, flag,
                       (const uint64_t*)hkey, rep);
    LAUNCH_CHECK("k_env_match");
    return BRIDGES_OK;
}

int bridges_head_sigmoid_dot(int32_t n_rows, int32_t K, int32_t N, const float* h, int64_t h_stride, const float* Wd,
                             const float* bd, const float* w, float* out, float* part, int32_t splits, void* stream) {
    if (n_rows < 0 || N <= 0 || !h || !Wd || !bd || !w || !out || splits < 1 || (splits > 1 && !part))
        return fail_arg("bridges_head_sigmoid_dot");
    if (K != HEAD_K) return fail_arg("bridges_head_sigmoid_dot: the hidden width must be 256");
    if ((h_stride & 3) || h_stride < K || ((((uintptr_t)h) | ((uintptr_t)Wd)) & 15)) return fail_arg("bridges_head_sigmoid_dot: rows must be 16-byte aligned");
    if (n_rows == 0) return BRIDGES_OK;
    const int tiles = (N + HEAD_BN - 1) / HEAD_BN;
    const int per = (tiles + splits - 1) / splits;
    const int used = (tiles + per - 1) / per;                             // ranges that hold at least one tile
    hipLaunchKernelGGL(k_head_sigmoid_dot, dim3((unsigned)((n_rows + 127) / 128), (unsigned)used), dim3(256), 0, (hipStream_t)stream,
                       n_rows, N, h, h_stride, Wd, bd, w, used > 1 ? part : out, per);
    LAUNCH_CHECK("k_head_sigmoid_dot");
    if (used > 1) {
        hipLaunchKernelGGL(k_head_sum, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_rows, used, part, out);
        LAUNCH_CHECK("k_head_sum");
    }
    return BRIDGES_OK;
}

int bridges_bits_dot(int32_t n_rows, const uint64_t* bits, const int64_t* bits_row, const float* img, const int64_t* slot,
                     float* out, void* stream) {
    if (n_rows < 0 || !bits || !img || !slot || !out) return fail_arg("bridges_bits_dot");
    if (n_rows == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_bits_dot, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n_rows, bits, bits_row,
                       img, slot, out);
    LAUNCH_CHECK("k_bits_dot");
    return BRIDGES_OK;
}

int bridges_bits_accumulate(int32_t n_rows, const uint64_t* bits, const int64_t* bits_row, const float* weight,
                            const int64_t* slot, float* img, void* stream) {
    if (n_rows < 0 || !bits || !img || !slot) return fail_arg("bridges_bits_accumulate");
    if (n_rows == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_bits_accumulate, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n_rows, bits,
                       bits_row, weight, slot, img);
    LAUNCH_CHECK("k_bits_accumulate");
    return BRIDGES_OK;
}

int bridges_sigmoid_dot(int32_t n_rows, const float* d, int64_t row_stride, const float* w, int32_t k, float* out,
                        void* stream) {
    if (n_rows < 0 || k <= 0 || (k & 3) || (row_stride & 3) || !d || !w || !out) return fail_arg("bridges_sigmoid_dot");
    if ((((uintptr_t)d) | ((uintptr_t)w)) & 15) return fail_arg("sigmoid_dot: rows must be 16-byte aligned");
    if (n_rows == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_sigmoid_dot, dim3(grid_for_waves(n_rows)), dim3(256), 0, (hipStream_t)stream, n_rows, d, row_stride,
                       w, k, out);
    LAUNCH_CHECK("k_sigmoid_dot");
    return BRIDGES_OK;
}

int bridges_bias_relu(float* x, const float* bias, int64_t n, int32_t C, int32_t hw, void* stream) {
    if (n < 0 || C <= 0 || hw <= 0 || (hw & 3) || !x || !bias) return fail_arg("bridges_bias_relu");
    if (((uintptr_t)x) & 15) return fail_arg("bias_relu: x must be 16-byte aligned");
    if (n == 0) return BRIDGES_OK;
    const int64_t n4 = n * C * (hw >> 2);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_bias_relu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, n4, hw >> 2, C);
    LAUNCH_CHECK("k_bias_relu");
    return BRIDGES_OK;
}

int bridges_bias_relu_pool2(const float* x, const float* bias, float* out, int64_t n, int32_t C, int32_t H, int32_t W,
                            void* stream) {
    if (n < 0 || C <= 0 || H <= 0 || W <= 0 || (W & 3) || (H & 1) || !x || !bias || !out) return fail_arg("bridges_bias_relu_pool2");
    if ((((uintptr_t)x) & 15) || (((uintptr_t)out) & 7)) return fail_arg("bias_relu_pool2: x must be 16-byte, out 8-byte aligned");
    if (n == 0) return BRIDGES_OK;
    const int64_t items = n * C * (H >> 1) * (W >> 2);
    int64_t blocks = (items + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_bias_relu_pool2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, out, items, H, W, C);
    LAUNCH_CHECK("k_bias_relu_pool2");
    return BRIDGES_OK;
}

// ---- small-batch MLP training step (mlp_kernels.hip) ------------------------------------------------------------
static int ceil_div(int a, int b) { return (a + b - 1) / b; }

int bridges_linear_forward(int32_t rows, int32_t K, int32_t N, const float* x, const float* W, const float* bias,
                           int32_t relu, float* y, float* ws, int64_t ws_floats, const int64_t* x_block, void* stream) {
    if (rows <= 0 || (rows & 31) || K <= 0 || N <= 0 || !x || !W || !bias || !y) return fail_arg("bridges_linear_forward: rows must be a positive multiple of 32");
    const int n_tiles = ceil_div(N, 32), m_tiles = rows / 32;
    // enough workgroups to fill the chip; a split holds at least 64 k values and its partial sums must fit in ws
    // (a short K or enough output tiles: one launch, no partial sums)
    int splits = (K <= 512 || n_tiles * m_tiles >= 128) ? 1 : ceil_div(512, n_tiles * m_tiles);
    const int max_by_k = ceil_div(K, 64);
    if (splits > max_by_k) splits = max_by_k;
    const int64_t per_split = (int64_t)rows * N;
    if (!ws || (int64_t)splits * per_split > ws_floats) splits = ws ? (int)(ws_floats / per_split) : 1;
    if (splits < 1) splits = 1;
    int kchunk = ceil_div(ceil_div(K, splits), 32) * 32;
    splits = ceil_div(K, kchunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_lin_fwd, dim3(n_tiles, splits, m_tiles), dim3(256), 0, st, K, N, kchunk, x, W, bias, relu, y,
                       splits > 1 ? ws : (float*)nullptr, x_block);
    LAUNCH_CHECK("k_lin_fwd");
    if (splits > 1) {
        int blocks = ceil_div(rows * N, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_lin_fwd_finish, dim3(blocks), dim3(256), 0, st, rows, N, splits, ws, bias, relu, y);
        LAUNCH_CHECK("k_lin_fwd_finish");
    }
    return BRIDGES_OK;
}

static int linear_backward_impl(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, const float* W,
                                float* dW, float* db, const float* act_below, float* dz_below, float* ws, int64_t ws_floats,
                                const int64_t* a_block, int32_t a_block_bias, LossLog log, void* stream) {
    if (rows <= 0 || (rows & 31) || K <= 0 || N <= 0 || !dz || !a_in || !W || !dW || !db) return fail_arg("bridges_linear_backward");
    const int n_ntiles = ceil_div(N, 32), n_ktiles = ceil_div(K, 32), m_tiles = rows / 32;
    int per_job = ceil_div(n_ntiles * n_ktiles, 1024);           // k tiles per dW job: ~1024 jobs on the big layers
    if (per_job < 4) per_job = 4;                                 // one tile per wave at least
    const int n_dw_jobs = n_ntiles * ceil_div(n_ktiles, per_job);
    int nsplit = 0, nchunk = 0, n_dx_jobs = 0;
    if (dz_below) {
        if (!ws) return fail_arg("bridges_linear_backward: workspace needed for the input gradient");
        nsplit = (N <= 512) ? 1 : ceil_div(256, n_ktiles * m_tiles);
        const int max_by_n = ceil_div(N, 64);
        if (nsplit > max_by_n) nsplit = max_by_n;
        const int64_t per_split = (int64_t)rows * K;
        if ((int64_t)nsplit * per_split > ws_floats) nsplit = (int)(ws_floats / per_split);
        if (nsplit < 1) return fail_arg("bridges_linear_backward: workspace too small");
        nchunk = ceil_div(ceil_div(N, nsplit), 32) * 32;
        nsplit = ceil_div(N, nchunk);
        n_dx_jobs = n_ktiles * nsplit * m_tiles;
    }
    hipStream_t st = (hipStream_t)stream;
    // one split: the input gradient goes straight to dz_below (masked), no partial sums
    hipLaunchKernelGGL(k_lin_bwd<false>, dim3(n_dw_jobs + n_dx_jobs), dim3(256), 0, st, rows, K, N, dz, a_in, W, dW, db,
                       !dz_below ? (float*)nullptr : (nsplit == 1 ? dz_below : ws), nsplit == 1 ? act_below : (const float*)nullptr,
                       n_dw_jobs, per_job, nsplit, nchunk, AdamFold{}, a_block, (int)a_block_bias, log);
    LAUNCH_CHECK("k_lin_bwd");
    if (dz_below && nsplit > 1) {
        int blocks = ceil_div(rows * K, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_lin_dx_finish, dim3(blocks), dim3(256), 0, st, rows, K, nsplit, ws, act_below, dz_below);
        LAUNCH_CHECK("k_lin_dx_finish");
    }
    return BRIDGES_OK;
}

int bridges_linear_backward(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, const float* W,
                            float* dW, float* db, const float* act_below, float* dz_below, float* ws, int64_t ws_floats,
                            const int64_t* a_block, int32_t a_block_bias, void* stream) {
    return linear_backward_impl(rows, K, N, dz, a_in, W, dW, db, act_below, dz_below, ws, ws_floats, a_block, a_block_bias, LossLog{}, stream);
}

int bridges_linear_backward_log(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, const float* W,
                                float* dW, float* db, const float* act_below, float* dz_below, float* ws, int64_t ws_floats,
                                const float* loss_rows, int32_t batch, float* losses, int32_t n_losses, int64_t* counter,
                                float* adam_step, void* stream) {
    if (!loss_rows || batch <= 0 || batch > rows || !counter || (losses && n_losses <= 0)) return fail_arg("bridges_linear_backward_log");
    return linear_backward_impl(rows, K, N, dz, a_in, W, dW, db, act_below, dz_below, ws, ws_floats, nullptr, 0,
                                LossLog{loss_rows, (int)batch, losses, (int)n_losses, counter, adam_step}, stream);
}

// the middle stack the k_mid_* kernels are instantiated for: SuccessorMLP's 256-128-64-128-256 (successor_dqn.py:366)
static bool mid_dims_supported(int32_t n_layers, const int32_t* dims) {
    static const int32_t want[5] = {256, 128, 64, 128, 256};
    if (n_layers != 4 || !dims) return false;
    for (int i = 0; i < 5; ++i) if (dims[i] != want[i]) return false;
    return true;
}
static int mid_ptrs_fill(const char* who, MidPtrs& p, const float* const* W, const float* const* bias, float* const* dW,
                         float* const* db, float* const* acts, float* const* dz) {
    if (!W || !acts) return fail_arg(who);
    for (int l = 0; l < 4; ++l) {
        if (!W[l] || (((uintptr_t)W[l]) & 15) || (bias && !bias[l]) || (dW && !dW[l]) || (db && !db[l])) return fail_arg(who);
        p.W[l] = W[l]; p.bias[l] = bias ? bias[l] : nullptr; p.dW[l] = dW ? dW[l] : nullptr; p.db[l] = db ? db[l] : nullptr;
    }
    for (int l = 0; l < 5; ++l) {
        if (!acts[l] || (((uintptr_t)acts[l]) & 15) || (dz && (l == 0 || l == 4) && (!dz[l] || (((uintptr_t)dz[l]) & 15)))) return fail_arg(who);
        p.act[l] = acts[l]; p.dz[l] = dz ? dz[l] : nullptr;
    }
    return BRIDGES_OK;
}

int bridges_mlp_mid_supported(int32_t rows, int32_t n_layers, const int32_t* dims) {
    return (rows == 32 && mid_dims_supported(n_layers, dims)) ? 1 : 0;
}

int bridges_mlp_mid_forward(int32_t rows, int32_t n_layers, const int32_t* dims, const float* const* W, const float* const* bias,
                            float* const* acts, void* stream) {
    if (rows != 32 || !mid_dims_supported(n_layers, dims) || !bias) return fail_arg("bridges_mlp_mid_forward: 32 rows of 256-128-64-128-256 only");
    MidPtrs p{};
    int rc = mid_ptrs_fill("bridges_mlp_mid_forward", p, W, bias, nullptr, nullptr, acts, nullptr);
    if (rc != BRIDGES_OK) return rc;
    hipLaunchKernelGGL((k_mid_fwd<256, 128, 64, 128, 256>), dim3(256 / 32), dim3(1024), 0, (hipStream_t)stream, p);
    LAUNCH_CHECK("k_mid_fwd");
    return BRIDGES_OK;
}

int bridges_mlp_mid_rows(int32_t n_rows, int32_t n_layers, const int32_t* dims, const float* const* W, const float* const* bias,
                         const float* x, int64_t x_stride, float* y, int64_t y_stride, float* mid, void* stream) {
    if (n_rows < 0 || !mid_dims_supported(n_layers, dims) || !W || !bias || !x || !y || !mid || x_stride < dims[0] || y_stride < dims[4] ||
        (x_stride & 3) || ((uintptr_t)x & 15) || ((uintptr_t)mid & 15))
        return fail_arg("bridges_mlp_mid_rows: 256-128-64-128-256 only, 16-byte aligned input rows, scratch of n x 64 floats");
    if (n_rows == 0) return BRIDGES_OK;
    for (int l = 0; l < 4; ++l)
        if (!W[l] || !bias[l] || (((uintptr_t)W[l]) & 15)) return fail_arg("bridges_mlp_mid_rows");
    const int n_tiles = (n_rows + 31) / 32;
    const dim3 grid((unsigned)(n_tiles < 256 ? n_tiles : 256));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((k_rows2<256, 128, 64, true>), grid, dim3(1024), 0, st, W[0], bias[0], W[1], bias[1], (int)n_rows, x, x_stride, mid,
                       (int64_t)64);
    LAUNCH_CHECK("k_rows2<256,128,64>");
    hipLaunchKernelGGL((k_rows2<64, 128, 256, false>), grid, dim3(1024), 0, st, W[2], bias[2], W[3], bias[3], (int)n_rows, (const float*)mid,
                       (int64_t)64, y, y_stride);
    LAUNCH_CHECK("k_rows2<64,128,256>");
    return BRIDGES_OK;
}

int bridges_mlp_mid_backward(int32_t rows, int32_t n_layers, const int32_t* dims, const float* const* W, float* const* dW,
                             float* const* db, float* const* acts, float* const* dz, float* rest_param, const float* rest_grad,
                             float* rest_exp_avg, float* rest_exp_avg_sq, int64_t rest_n, const float* step, double lr, double beta1,
                             double beta2, double eps, void* stream) {
    if (rows != 32 || !mid_dims_supported(n_layers, dims) || !dW || !db || !dz) return fail_arg("bridges_mlp_mid_backward: 32 rows of 256-128-64-128-256 only");
    if (rest_n < 0 || (rest_n & 3) || (rest_n > 0 && (!rest_param || !rest_grad || !rest_exp_avg || !rest_exp_avg_sq || !step)))
        return fail_arg("bridges_mlp_mid_backward: the Adam range must be a multiple of 4 floats with all four buffers and the step");
    if (rest_n > 0 && ((((uintptr_t)rest_param) | ((uintptr_t)rest_grad) | ((uintptr_t)rest_exp_avg) | ((uintptr_t)rest_exp_avg_sq)) & 15))
        return fail_arg("bridges_mlp_mid_backward: Adam buffers must be 16-byte aligned");
    if (rest_n > 0 && (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)))
        return fail_arg("bridges_mlp_mid_backward: hyper-parameters");
    MidPtrs p{};
    int rc = mid_ptrs_fill("bridges_mlp_mid_backward", p, W, nullptr, dW, db, acts, dz);
    if (rc != BRIDGES_OK) return rc;
    p.rest_p = rest_param; p.rest_g = rest_grad; p.rest_m = rest_exp_avg; p.rest_v = rest_exp_avg_sq; p.rest_n = (long long)rest_n;
    p.step = step; p.lr = lr; p.beta1 = beta1; p.beta2 = beta2; p.eps = eps;
    // riders: ~4 float4 groups per thread over the range, at most 248 workgroups (one per CU beside the stack's eight)
    int64_t riders = rest_n > 0 ? ((rest_n >> 2) + 4095) / 4096 : 0;
    if (riders > 248) riders = 248;
    hipLaunchKernelGGL((k_mid_bwd<256, 128, 64, 128, 256>), dim3(256 / 32 + (unsigned)riders), dim3(1024), 0, (hipStream_t)stream, p);
    LAUNCH_CHECK("k_mid_bwd");
    return BRIDGES_OK;
}

int bridges_linear_backward_adam(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, float* W, float* bias,
                                 float* exp_avg_w, float* exp_avg_sq_w, float* exp_avg_b, float* exp_avg_sq_b, float* rest_param,
                                 const float* rest_grad, float* rest_exp_avg, float* rest_exp_avg_sq, int64_t rest_n, const float* step,
                                 double lr, double beta1, double beta2, double eps, const int64_t* a_block, int32_t a_block_bias,
                                 void* stream) {
    if (rows != 32 || K <= 0 || N <= 0 || !dz || !a_in || !W || !bias || !exp_avg_w || !exp_avg_sq_w || !exp_avg_b || !exp_avg_sq_b || !step)
        return fail_arg("bridges_linear_backward_adam: one 32-row batch tile, all buffers given");
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return fail_arg("bridges_linear_backward_adam: hyper-parameters");
    if (rest_n < 0 || (rest_n & 3) || (rest_n > 0 && (!rest_param || !rest_grad || !rest_exp_avg || !rest_exp_avg_sq)))
        return fail_arg("bridges_linear_backward_adam: rest range must be a multiple of 4 floats with all four buffers");
    if (rest_n > 0 && ((((uintptr_t)rest_param) | ((uintptr_t)rest_grad) | ((uintptr_t)rest_exp_avg) | ((uintptr_t)rest_exp_avg_sq)) & 15))
        return fail_arg("bridges_linear_backward_adam: rest buffers must be 16-byte aligned");
    const int n_ntiles = ceil_div(N, 32), n_ktiles = ceil_div(K, 32);
    // ~256 weight-gradient jobs (a workgroup then walks ~2 KB of every row of W / m / v) and <= 256 workgroups for the other
    // layers' range: measured 142 us per optimiser step against 156 us with 1024 + 1024 (the update is bound by DRAM locality
    // of six strided streams, not by parallelism; 128 jobs: 146 us, 64: 170 us)
    int per_job = ceil_div(n_ntiles * n_ktiles, 256);             // (128 / 192 / 384 / 512 jobs with 512 threads: 119 / 114 / 115 / 117 us per step, 256: 112-114)
    if (per_job < 4) per_job = 4;
    const int n_dw_jobs = n_ntiles * ceil_div(n_ktiles, per_job);
    int64_t rest_jobs = ((rest_n >> 2) + 255) / 256;
    if (rest_jobs > 256) rest_jobs = 256;
    AdamFold ad{W, bias, exp_avg_w, exp_avg_sq_w, exp_avg_b, exp_avg_sq_b, step, lr, beta1, beta2, eps,
                rest_param, rest_grad, rest_exp_avg, rest_exp_avg_sq, (long long)rest_n};
    // 512-thread workgroups: the eight waves of a weight-gradient job walk eight ADJACENT k tiles, 1 KB of every row of W / m / v
    // at a time (two workgroups per CU at the kernel's 247 registers would do the same with 512 B)
    hipLaunchKernelGGL(k_lin_bwd<true>, dim3(n_dw_jobs + (int)rest_jobs), dim3(512), 0, (hipStream_t)stream, rows, K, N, dz, a_in,
                       (const float*)W, (float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, n_dw_jobs, per_job,
                       0, 0, ad, a_block, (int)a_block_bias, LossLog{});
    LAUNCH_CHECK("k_lin_bwd<adam>");
    return BRIDGES_OK;
}

int bridges_mlp_input(int32_t batch, int32_t rows, int32_t px, int32_t nf, const int64_t* counter, const float* block_all,
                      const float* action_all, const float* binary_all, const float* reward, const float* obstacle,
                      float* x, void* stream) {
    if (batch <= 0 || rows < batch || (rows & 31) || px <= 0 || nf < 0 || !counter || !block_all || !action_all || !reward || !obstacle || !x)
        return fail_arg("bridges_mlp_input");
    int blocks = ceil_div(rows * (4 * px + nf), 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mlp_input, dim3(blocks), dim3(256), 0, (hipStream_t)stream, batch, rows, px, nf, counter, block_all,
                       action_all, binary_all, reward, obstacle, x);
    LAUNCH_CHECK("k_mlp_input");
    return BRIDGES_OK;
}

int bridges_mlp_input_batches(int32_t n_batches, int32_t batch, int32_t rows, int32_t px, int32_t nf, const float* block_all,
                              const float* action_all, const float* binary_all, const float* reward, const float* obstacle,
                              float* x_all, void* stream) {
    if (n_batches <= 0 || batch <= 0 || rows < batch || (rows & 31) || px <= 0 || nf < 0 || !block_all || !action_all || !reward || !obstacle || !x_all)
        return fail_arg("bridges_mlp_input_batches");
    int blocks = ceil_div(rows * (4 * px + nf), 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mlp_input, dim3(blocks, n_batches), dim3(256), 0, (hipStream_t)stream, batch, rows, px, nf,
                       (const int64_t*)nullptr, block_all, action_all, binary_all, reward, obstacle, x_all);
    LAUNCH_CHECK("k_mlp_input (all batches)");
    return BRIDGES_OK;
}

int bridges_successor_loss(int32_t batch, int32_t rows, int32_t px, int32_t nf, const float* y, const float* reward,
                           const int64_t* counter, const float* q_target_all, const float* sf_target_all, int32_t use_q,
                           int32_t use_sf, float* dy, float* loss_rows, float* q_out, float* losses, int32_t n_losses,
                           int64_t* counter_inc, int32_t* ticket, float* adam_step, void* stream) {
    if (batch <= 0 || rows < batch || px <= 0 || nf < 0 || !y || !reward || !counter || !dy || !loss_rows || !q_out)
        return fail_arg("bridges_successor_loss");
    if ((use_q && !q_target_all) || (use_sf && !sf_target_all)) return fail_arg("bridges_successor_loss: target missing");
    if (ticket && !counter_inc) return fail_arg("bridges_successor_loss: a ticket needs counter_inc");
    if (adam_step && !ticket) return fail_arg("bridges_successor_loss: adam_step is advanced by the ticket holder");
    hipStream_t st = (hipStream_t)stream;
    // with a ticket word (zero before the first call; the kernel re-arms it) the logging happens inside the loss kernel
    hipLaunchKernelGGL(k_successor_loss, dim3(rows), dim3(LOSS_THREADS), 0, st, batch, px, nf, y, reward, counter, q_target_all,
                       sf_target_all, use_q, use_sf, dy, loss_rows, q_out, losses, n_losses, ticket ? counter_inc : (int64_t*)nullptr,
                       ticket, adam_step);
    LAUNCH_CHECK("k_successor_loss");
    if (!ticket && losses && counter_inc) {
        hipLaunchKernelGGL(k_loss_log, dim3(1), dim3(64), 0, st, batch, loss_rows, losses, n_losses, counter_inc);
        LAUNCH_CHECK("k_loss_log");
    }
    return BRIDGES_OK;
}

int bridges_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* step,
                      double lr, double beta1, double beta2, double eps, void* stream) {
    if (n < 0 || !param || !grad || !exp_avg || !exp_avg_sq || !step) return fail_arg("bridges_adam_step");
    if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15)
        return fail_arg("bridges_adam_step: buffers must be 16-byte aligned");
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return fail_arg("bridges_adam_step: hyper-parameters");
    if (n == 0) return BRIDGES_OK;
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_adam_flat, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, step,
                       lr, beta1, beta2, eps);
    LAUNCH_CHECK("k_adam_flat");
    return BRIDGES_OK;
}

int bridges_adam_multi(const bridges_adam_slot* slots, int32_t n_slots, const int32_t* chunk_slot, const int32_t* chunk_off,
                       int32_t n_chunks, const float* step, double lr, double beta1, double beta2, double eps, void* stream) {
    if (n_slots < 0 || n_chunks < 0 || !step || (n_chunks > 0 && (!slots || !chunk_slot || !chunk_off || n_slots < 1))) return fail_arg("bridges_adam_multi");
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return fail_arg("bridges_adam_multi: hyper-parameters");
    if (n_chunks == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, slots, chunk_slot, chunk_off, step, lr, beta1,
                       beta2, eps);
    LAUNCH_CHECK("k_adam_multi");
    return BRIDGES_OK;
}

// ---- conv3x3 + bias + ReLU [+ pool] for the 64-wide, 16-output-channel layers (conv_kernels.hip) ------------------
int bridges_conv3x3_relu_o16_ex(const float* x, const float* x2, const float* w, const float* bias, float* out, float* out2,
                                const float* proj_w, const float* proj_b, int64_t n, int32_t c_in, int32_t c_in2, int32_t H,
                                int32_t W, int32_t mode, void* stream) {
    if (n < 0 || !x || !w || !bias || !out) return fail_arg("bridges_conv3x3_relu_o16");
    if (W != CONV_W || H <= 0 || (H % CONV_BAND) != 0) return fail_arg("bridges_conv3x3_relu_o16: W must be 64 and H a multiple of 8");
    if (mode < CONV_EPI_PLAIN || mode > CONV_EPI_PROJ) return fail_arg("bridges_conv3x3_relu_o16: mode");
    if (mode == CONV_EPI_BOTH && !out2) return fail_arg("bridges_conv3x3_relu_o16: out2 missing");
    if (mode == CONV_EPI_PROJ && (!proj_w || !proj_b)) return fail_arg("bridges_conv3x3_relu_o16: projection weights missing");
    if (x2) {
        if (c_in != 16 || c_in2 != 16) return fail_arg("bridges_conv3x3_relu_o16: two inputs must hold 16 channels each");
    } else {
        c_in2 = 0;
        if (!(c_in >= 1 && c_in <= 4) && c_in != 16 && c_in != 32) return fail_arg("bridges_conv3x3_relu_o16: C_in must be 1..4, 16 or 32");
    }
    if ((((uintptr_t)x) & 15) || (((uintptr_t)out) & 15) || (((uintptr_t)x2) & 15) || (((uintptr_t)out2) & 7))
        return fail_arg("bridges_conv3x3_relu_o16: tensors must be 16-byte aligned");
    if (n == 0) return BRIDGES_OK;
    const int64_t blocks = n * (H / CONV_BAND);
    if (blocks > 0x7fffffff) return fail_arg("bridges_conv3x3_relu_o16: too many images");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)blocks), b(256);
#define CONV_LAUNCH_E(CC, NC, E)                                                                                        \
    hipLaunchKernelGGL((k_conv3x3_o16<CC, NC, E>), g, b, 0, st, x, x2, w, bias, out, out2, proj_w, proj_b, (int)H, (int)c_in, (int)c_in2)
#define CONV_LAUNCH(CC, NC)                                                                                             \
    do {                                                                                                                \
        if (mode == CONV_EPI_PLAIN) CONV_LAUNCH_E(CC, NC, CONV_EPI_PLAIN);                                              \
        else if (mode == CONV_EPI_POOL) CONV_LAUNCH_E(CC, NC, CONV_EPI_POOL);                                           \
        else if (mode == CONV_EPI_BOTH) CONV_LAUNCH_E(CC, NC, CONV_EPI_BOTH);                                           \
        else CONV_LAUNCH_E(CC, NC, CONV_EPI_PROJ);                                                                      \
    } while (0)
    if (x2 || c_in == 32) CONV_LAUNCH(16, 2);
    else if (c_in <= 4) CONV_LAUNCH(4, 1);
    else CONV_LAUNCH(16, 1);
#undef CONV_LAUNCH
#undef CONV_LAUNCH_E
    LAUNCH_CHECK("k_conv3x3_o16");
    return BRIDGES_OK;
}

int bridges_conv3x3_relu_o16(const float* x, const float* w, const float* bias, float* out, int64_t n, int32_t c_in,
                             int32_t H, int32_t W, int32_t pool, void* stream) {
    return bridges_conv3x3_relu_o16_ex(x, nullptr, w, bias, out, nullptr, nullptr, nullptr, n, c_in, 0, H, W,
                                       pool ? CONV_EPI_POOL : CONV_EPI_PLAIN, stream);
}

int bridges_upconv2x2(const float* x, const float* w, const float* bias, float* out, int64_t n, int32_t c_in, int32_t c_out,
                      int32_t H, int32_t W, void* stream) {
    if (n < 0 || !x || !w || !bias || !out || H <= 0 || W <= 0 || (W % 16) != 0) return fail_arg("bridges_upconv2x2: W must be a multiple of 16");
    if (!((c_in == 32 && c_out == 16) || (c_in == 64 && c_out == 32))) return fail_arg("bridges_upconv2x2: (C_in, C_out) must be (32, 16) or (64, 32)");
    if ((((uintptr_t)out) & 15) || (((uintptr_t)w) & 7)) return fail_arg("bridges_upconv2x2: out must be 16-byte, w 8-byte aligned");
    if (n == 0) return BRIDGES_OK;
    const int64_t tiles = n * H * (W / 16);
    const int64_t blocks = (tiles + 3) / 4;
    if (blocks > 0x7fffffff) return fail_arg("bridges_upconv2x2: too many images");
    hipStream_t st = (hipStream_t)stream;
    if (c_in == 32) hipLaunchKernelGGL((k_upconv2x2<32, 1>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, bias, out, (int)H, (int)W, (long)tiles);
    else hipLaunchKernelGGL((k_upconv2x2<64, 2>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, bias, out, (int)H, (int)W, (long)tiles);
    LAUNCH_CHECK("k_upconv2x2");
    return BRIDGES_OK;
}

}  // extern "C"

// ---- K11: ConvBlock training passes (csrc/conv_train_kernels.hip) ---------------------------------------------------------
template <int W, int CH>
static void launch_c3(int mode, dim3 grid, hipStream_t s, const float* x, const float* in_mask, const float* w, const float* bias,
                      const float* mask, float* out, int c_in, int c_out, int w_sin, int w_sout, int flip) {
    if (mode == C3_EPI_BIAS_RELU) hipLaunchKernelGGL((k_c3<W, CH, C3_EPI_BIAS_RELU>), grid, dim3(256), 0, s, x, in_mask, w, bias, mask, out, c_in, c_out, w_sin, w_sout, flip);
    else if (mode == C3_EPI_MASK) hipLaunchKernelGGL((k_c3<W, CH, C3_EPI_MASK>), grid, dim3(256), 0, s, x, in_mask, w, bias, mask, out, c_in, c_out, w_sin, w_sout, flip);
    else hipLaunchKernelGGL((k_c3<W, CH, C3_EPI_RAW>), grid, dim3(256), 0, s, x, in_mask, w, bias, mask, out, c_in, c_out, w_sin, w_sout, flip);
}

int bridges_conv3x3(const float* x, const float* in_mask, const float* w, const float* bias, const float* mask, float* out, int64_t n,
                    int32_t c_in, int32_t c_out, int32_t W, int32_t mode, int32_t transposed, void* stream) {
    if (n < 0 || !x || !w || !out || c_in < 1 || c_out < 16 || (c_out & 15)) return fail_arg("bridges_conv3x3: channels (C_out must be a multiple of 16)");
    if (W != 8 && W != 16 && W != 32 && W != 64) return fail_arg("bridges_conv3x3: W must be 8, 16, 32 or 64 (square images)");
    if (mode < C3_EPI_RAW || mode > C3_EPI_MASK || (mode == C3_EPI_BIAS_RELU && !bias) || (mode == C3_EPI_MASK && !mask)) return fail_arg("bridges_conv3x3: mode");
    if ((((uintptr_t)out) | ((uintptr_t)mask) | ((uintptr_t)x) | ((uintptr_t)in_mask)) & 15)
        return fail_arg("bridges_conv3x3: x / in_mask / out / mask must be 16-byte aligned");
    if (n == 0) return BRIDGES_OK;
    const int bands = W / c3_band_rows(W);
    if (n * bands > 0x7fffffff) return fail_arg("bridges_conv3x3: too many images");
    const dim3 grid((unsigned)(n * bands), (unsigned)(c_out / 16));
    // forward: w [c_out, c_in, 3, 3]; transposed (input gradient): w [c_in, c_out, 3, 3] of the layer, taps flipped
    const int w_sin = transposed ? c_out * 9 : 9, w_sout = transposed ? 9 : c_in * 9, flip = transposed ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
#define C3_DISPATCH(WW)                                                                                                \
    if (c_in <= 4) launch_c3<WW, 4>(mode, grid, s, x, in_mask, w, bias, mask, out, c_in, c_out, w_sin, w_sout, flip);  \
    else launch_c3<WW, 16>(mode, grid, s, x, in_mask, w, bias, mask, out, c_in, c_out, w_sin, w_sout, flip);
    if (W == 64) { C3_DISPATCH(64) } else if (W == 32) { C3_DISPATCH(32) } else if (W == 16) { C3_DISPATCH(16) } else { C3_DISPATCH(8) }
#undef C3_DISPATCH
    LAUNCH_CHECK("k_c3");
    return BRIDGES_OK;
}

int bridges_conv3x3_wgrad_scratch(int64_t n, int32_t c_in, int32_t c_out, int32_t W, int64_t* floats) {
    if (!floats || n < 1 || c_in < 1 || c_out < 16 || (c_out & 15) || (W != 8 && W != 16 && W != 32 && W != 64)) return fail_arg("bridges_conv3x3_wgrad_scratch");
    const int64_t units = n * (W / c3_wgrad_rows(W)), tiles = (int64_t)(c_out / 16) * ((c_in + 15) / 16);
    int64_t splits = 512 / tiles;
    if (splits < 1) splits = 1;
    if (splits > units) splits = units;
    const int64_t ups = (units + splits - 1) / splits;
    splits = (units + ups - 1) / ups;
    *floats = splits * ((int64_t)c_out * c_in * 9 + c_out);
    return BRIDGES_OK;
}

int bridges_conv3x3_wgrad(const float* g, const float* g_mask, const float* x, float* dw, float* db, float* scratch, int64_t scratch_floats,
                          int64_t n, int32_t c_in, int32_t c_out, int32_t W, void* stream) {
    if (!g || !x || !scratch || (!dw) != (!db)) return fail_arg("bridges_conv3x3_wgrad");
    if ((((uintptr_t)g) | ((uintptr_t)g_mask) | ((uintptr_t)x)) & 15) return fail_arg("bridges_conv3x3_wgrad: g / g_mask / x must be 16-byte aligned");
    int64_t need = 0;
    int rc = bridges_conv3x3_wgrad_scratch(n, c_in, c_out, W, &need);
    if (rc != BRIDGES_OK) return rc;
    if (scratch_floats < need) return fail_arg("bridges_conv3x3_wgrad: scratch too small (bridges_conv3x3_wgrad_scratch)");
    const int64_t units = n * (W / c3_wgrad_rows(W));
    const int tiles = (c_out / 16) * ((c_in + 15) / 16);
    const int64_t n_w = (int64_t)c_out * c_in * 9;
    const int splits = (int)(need / (n_w + c_out));
    const int ups = (int)((units + splits - 1) / splits);
    float* part = scratch;
    float* part_b = scratch + (int64_t)splits * n_w;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    if (W == 64) hipLaunchKernelGGL(k_c3_wgrad<64>, grid, dim3(256), 0, s, g, g_mask, x, part, part_b, (int)n, c_in, c_out, ups);
    else if (W == 32) hipLaunchKernelGGL(k_c3_wgrad<32>, grid, dim3(256), 0, s, g, g_mask, x, part, part_b, (int)n, c_in, c_out, ups);
    else if (W == 16) hipLaunchKernelGGL(k_c3_wgrad<16>, grid, dim3(256), 0, s, g, g_mask, x, part, part_b, (int)n, c_in, c_out, ups);
    else hipLaunchKernelGGL(k_c3_wgrad<8>, grid, dim3(256), 0, s, g, g_mask, x, part, part_b, (int)n, c_in, c_out, ups);
    LAUNCH_CHECK("k_c3_wgrad");
    if (!dw) return BRIDGES_OK;                                    // partial sums only: the caller reduces them (bridges_reduce_jobs)
    const int64_t tot = n_w + c_out;
    hipLaunchKernelGGL(k_c3_reduce, dim3((unsigned)c3_reduce_blocks(tot, splits)), dim3(256), 0, s, (const float*)part, (const float*)part_b, dw, db,
                       (int)n_w, c_out, splits);
    LAUNCH_CHECK("k_c3_reduce");
    return BRIDGES_OK;
}

int bridges_maxpool2(const float* a, float* y, int64_t nc, int32_t H, int32_t W, void* stream) {
    if (nc < 0 || !a || !y || H <= 0 || W <= 0 || (W & 3) || (H & 1)) return fail_arg("bridges_maxpool2");
    if ((((uintptr_t)a) & 15) || (((uintptr_t)y) & 7)) return fail_arg("bridges_maxpool2: alignment");
    const int64_t items = nc * (H / 2) * (W / 4);
    if (items == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_maxpool2, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, y, items, H, W);
    LAUNCH_CHECK("k_maxpool2");
    return BRIDGES_OK;
}

int bridges_maxpool2_relu_backward(const float* a, const float* dy, float* g, int64_t nc, int32_t H, int32_t W, void* stream) {
    if (nc < 0 || !a || !dy || !g || H <= 0 || W <= 0 || (W & 3) || (H & 1)) return fail_arg("bridges_maxpool2_relu_backward");
    if (((((uintptr_t)a) | ((uintptr_t)g)) & 15) || (((uintptr_t)dy) & 7)) return fail_arg("bridges_maxpool2_relu_backward: alignment");
    const int64_t items = nc * (H / 2) * (W / 4);
    if (items == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_maxpool2_relu_bwd, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, dy, g, items, H, W);
    LAUNCH_CHECK("k_maxpool2_relu_bwd");
    return BRIDGES_OK;
}

int bridges_bias_grad(const float* g, float* db, float* scratch, int64_t scratch_floats, int64_t n, int32_t C, int32_t hw, void* stream) {
    if (!g || !db || !scratch || n < 1 || C < 1 || hw < 1) return fail_arg("bridges_bias_grad");
    int S = (int)(n < 32 ? n : 32);
    if (scratch_floats < (int64_t)S * C) return fail_arg("bridges_bias_grad: scratch needs min(n, 32) * C floats");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bias_grad_part, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, scratch, (int)n, C, hw, S);
    LAUNCH_CHECK("k_bias_grad_part");
    hipLaunchKernelGGL(k_c3_reduce, dim3((unsigned)c3_reduce_blocks(C, S)), dim3(256), 0, s, (const float*)scratch, (const float*)scratch, db, db, 0, C, S);
    LAUNCH_CHECK("k_c3_reduce");
    return BRIDGES_OK;
}

// ---- backward of the U-Net's transposed / 1x1 convolutions (conv_train_kernels.hip) -----------------------------------------------
static int up2_splits(int64_t tiles, int* tps) {
    int per = (int)((tiles + 255) / 256);
    if (per < 1) per = 1;
    *tps = per;
    return (int)((tiles + per - 1) / per);
}

extern "C" {

int bridges_upconv2x2_backward_scratch(int64_t n, int32_t c_in, int32_t c_out, int32_t H, int32_t W, int64_t* floats) {
    if (!floats || n < 0 || H < 1 || W < 1 || (((int64_t)H * W) & 63)) return fail_arg("bridges_upconv2x2_backward_scratch: H * W must be a multiple of 64");
    if (!((c_in == 32 && c_out == 16) || (c_in == 64 && c_out == 32))) return fail_arg("bridges_upconv2x2_backward_scratch: (C_in, C_out) must be (32, 16) or (64, 32)");
    int tps;
    const int S = up2_splits(n * H * W / 64, &tps);
    *floats = (int64_t)(S < 1 ? 1 : S) * ((int64_t)c_in * c_out * 4 + c_out);
    return BRIDGES_OK;
}

int bridges_upconv2x2_backward(const float* x, const float* g, const float* w, float* dx, float* dw, float* db, float* scratch,
                               int64_t scratch_floats, int64_t n, int32_t c_in, int32_t c_out, int32_t H, int32_t W, void* stream) {
    if (!x || !g || !w || !scratch || (!dw) != (!db)) return fail_arg("bridges_upconv2x2_backward");
    int64_t need = 0;
    int rc = bridges_upconv2x2_backward_scratch(n, c_in, c_out, H, W, &need);
    if (rc != BRIDGES_OK) return rc;
    if (scratch_floats < need) return fail_arg("bridges_upconv2x2_backward: scratch too small (bridges_upconv2x2_backward_scratch)");
    if (((uintptr_t)g) & 7) return fail_arg("bridges_upconv2x2_backward: g must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = n * H * W / 64;
    if (tiles > 0x7fffffff) return fail_arg("bridges_upconv2x2_backward: too many images");
    const int K = c_out * 4;
    float* part = scratch;
    int tps;
    const int S = up2_splits(tiles, &tps);
    float* part_b = scratch + (size_t)(S < 1 ? 1 : S) * c_in * K;
    if (tiles == 0 && dw) {                                         // no image: zero gradients
        if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)c_in * K, s) != hipSuccess || hipMemsetAsync(db, 0, sizeof(float) * (size_t)c_out, s) != hipSuccess)
            return fail_arg("bridges_upconv2x2_backward: hipMemsetAsync");
        return BRIDGES_OK;
    }
    if (dx) {
        hipLaunchKernelGGL(k_up2_dx, dim3((unsigned)tiles, (unsigned)((c_in + 15) / 16)), dim3(256), (size_t)K * 80 * sizeof(float), s, g, w, dx,
                           c_in, c_out, H, W);
        LAUNCH_CHECK("k_up2_dx");
    }
    if (c_in == 64) hipLaunchKernelGGL((k_up2_wgrad<4, 8>), dim3((unsigned)S), dim3(256), 0, s, x, g, part, part_b, c_out, H, W, (int)tiles, tps);
    else hipLaunchKernelGGL((k_up2_wgrad<2, 4>), dim3((unsigned)S), dim3(256), 0, s, x, g, part, part_b, c_out, H, W, (int)tiles, tps);
    LAUNCH_CHECK("k_up2_wgrad");
    if (!dw) return BRIDGES_OK;                                    // partial sums only (bridges_reduce_jobs)
    const int n_w = c_in * K;
    hipLaunchKernelGGL(k_c3_reduce, dim3((unsigned)c3_reduce_blocks(n_w + c_out, S)), dim3(256), 0, s, (const float*)part, (const float*)part_b, dw, db,
                       n_w, c_out, S);
    LAUNCH_CHECK("k_c3_reduce");
    return BRIDGES_OK;
}

int bridges_conv1x1_o1_forward(const float* x, const float* w, const float* bias, float* y, int64_t n, int32_t c_in, int32_t hw, void* stream) {
    if (!x || !w || !bias || !y || n < 0 || c_in < 1 || hw < 4 || (hw & 3)) return fail_arg("bridges_conv1x1_o1_forward: H * W must be a multiple of 4");
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return fail_arg("bridges_conv1x1_o1_forward: x / y must be 16-byte aligned");
    const int64_t quads = n * hw / 4;
    if (quads == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_pw1_fwd, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, c_in, hw, quads);
    LAUNCH_CHECK("k_pw1_fwd");
    return BRIDGES_OK;
}

int bridges_conv1x1_o1_backward(const float* x, const float* g, const float* w, float* dx, float* dw, float* db, float* scratch,
                                int64_t scratch_floats, int64_t n, int32_t c_in, int32_t hw, void* stream) {
    if (!x || !g || !w || !dx || !scratch || (!dw) != (!db) || n < 0 || c_in < 1 || c_in > 32 || hw < 4 || (hw & 3))
        return fail_arg("bridges_conv1x1_o1_backward: C_in <= 32, H * W a multiple of 4");
    if ((((uintptr_t)x) | ((uintptr_t)g) | ((uintptr_t)dx)) & 15) return fail_arg("bridges_conv1x1_o1_backward: x / g / dx must be 16-byte aligned");
    const int64_t quads = n * hw / 4;
    int64_t S = (quads + 255) / 256;
    if (S > 256) S = 256;
    if (S < 1) S = 1;
    if (scratch_floats < S * (c_in + 1)) return fail_arg("bridges_conv1x1_o1_backward: scratch needs min(256, ceil(n * hw / 1024)) * (C_in + 1) floats");
    hipStream_t s = (hipStream_t)stream;
    float* part = scratch;
    float* part_b = scratch + S * c_in;
    hipLaunchKernelGGL(k_pw1_bwd, dim3((unsigned)S), dim3(256), 0, s, x, g, w, dx, part, part_b, c_in, hw, quads);
    LAUNCH_CHECK("k_pw1_bwd");
    if (!dw) return BRIDGES_OK;                                    // partial sums only (bridges_reduce_jobs)
    hipLaunchKernelGGL(k_c3_reduce, dim3((unsigned)c3_reduce_blocks(c_in + 1, (int)S)), dim3(256), 0, s, (const float*)part, (const float*)part_b, dw, db, c_in, 1,
                       (int)S);
    LAUNCH_CHECK("k_c3_reduce");
    return BRIDGES_OK;
}

}  // extern "C"

extern "C" int bridges_reduce_jobs(const bridges_reduce_job* jobs_dev, int32_t n_jobs, int32_t total_blocks, void* stream) {
    if (n_jobs < 0 || total_blocks < 0 || (n_jobs > 0 && (!jobs_dev || total_blocks < 1))) return fail_arg("bridges_reduce_jobs");
    if (n_jobs == 0) return BRIDGES_OK;
    hipLaunchKernelGGL(k_reduce_jobs, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    LAUNCH_CHECK("k_reduce_jobs");
    return BRIDGES_OK;
}

/*
 * bridges_hip.h -- C ABI of libbridges_hip.so (MI355X / gfx950).
 *
 * The reference (syghmon/bridges-with-reinforcement-learning) is pure Python and has
 * no FFI layer; this header is the boundary SURVEY.md §8(b) defines for its hot path.
 * Every entry point names the reference code it replaces (paths relative to the
 * reference root).  Conventions:
 *   - all buffers are DEVICE pointers owned by the caller (PyTorch tensors in the
 *     shipped host code); the library allocates only small constant tables inside
 *     *_create and nothing afterwards;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is
 *     enqueued asynchronously, nothing synchronises;
 *   - return value 0 = ok, negative = BRIDGES_E_*; no exceptions cross the ABI.
 *
 * Reference-side binding: INTEGRATION.md shows the ctypes stub a maintainer adds.
 */
#ifndef BRIDGES_HIP_H
#define BRIDGES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRIDGES_OK 0
#define BRIDGES_E_ARG (-1)      /* bad argument / size over a compiled limit */
#define BRIDGES_E_HIP (-2)      /* a HIP runtime call failed (bridges_last_error()) */
#define BRIDGES_E_NODEV (-3)    /* no HIP device */

#define BRIDGES_MAX_VERTS 6     /* hexagon */
#define BRIDGES_MAX_BLOCKS 16   /* --max_steps <= 16 */
#define BRIDGES_MAX_GROUPS 24   /* (shape, target face) pairs of one task */
#define BRIDGES_MAX_TARGETS 8
#define BRIDGES_MAX_INTERFACES 64
#define BRIDGES_IMG 64          /* rasters are 64x64 (successor_dqn.py:585 default) */
/* doubles of lp_ws per environment: 64 (header, basis) + 2 halves of (3*MAX_BLOCKS+2) rows x (4*MAX_IF + 2 + 3*MAX_BLOCKS + 1) columns */
#define BRIDGES_LP_WS_DOUBLES (64 + 2 * (3 * BRIDGES_MAX_BLOCKS + 2) * (4 * BRIDGES_MAX_INTERFACES + 2 + 3 * BRIDGES_MAX_BLOCKS + 1))
#define BRIDGES_LP_SNAP_DOUBLES (64 + (3 * BRIDGES_MAX_BLOCKS + 2) * (4 * BRIDGES_MAX_INTERFACES + 2 + 3 * BRIDGES_MAX_BLOCKS + 1))
#define BRIDGES_CAND_WS_SLOTS 1024

/* One block shape: a convex (x,z) outline extruded along y.
 * Replaces Shape.from_urdf / from_mesh / get_face_frame_2d
 * (assembly_gym/assembly_gym/envs/assembly_env.py:45-68, 118-124).  The host
 * fills the derived fields with the arithmetic contract of DESIGN.md. */
typedef struct {
    int32_t nv;                              /* vertices == 2-D faces */
    int32_t pad_;
    double vx[BRIDGES_MAX_VERTS], vz[BRIDGES_MAX_VERTS];
    int32_t fa[BRIDGES_MAX_VERTS], fb[BRIDGES_MAX_VERTS];   /* face f = directed edge fa->fb */
    double fcx[BRIDGES_MAX_VERTS], fcz[BRIDGES_MAX_VERTS];  /* local face centre */
    double fnx[BRIDGES_MAX_VERTS], fnz[BRIDGES_MAX_VERTS];  /* local outward normal */
    double depth;                            /* extent along y */
    double volume;                           /* area * depth */
    double gx, gz;                           /* local centroid */
} bridges_shape;

/* Static description of one vectorised task (all environments share it).
 * Replaces the arguments of AssemblyGym.__init__/reset, AssemblyEnv.__init__
 * (gym_env.py:116-139, 255-289; assembly_env.py:164-199) and the constants of
 * successor_dqn.py:611-616. */
typedef struct {
    int32_t n_envs;
    int32_t max_blocks;        /* state capacity K (>= max_steps) */
    int32_t max_steps;         /* 0 = None (gym_env.py:143) */
    int32_t a_max;             /* candidate capacity per env */
    int32_t n_shapes;          /* entries of `shapes` */
    int32_t n_groups;          /* (shape, target_face) pairs in generate_actions order (actions.py:16-19) */
    int32_t group_shape[BRIDGES_MAX_GROUPS];
    int32_t group_face[BRIDGES_MAX_GROUPS];
    int32_t n_ground;          /* len(x_discr_ground) */
    int32_t n_offsets;         /* len(offset_values) */
    int32_t n_targets;
    int32_t debug;             /* 0.  Diagnostic builds (-DBRIDGES_DIAG) read timing-experiment switches from it; the
                                  product library refuses any other value */
    int32_t env_id_base;       /* global id of env 0 (policy RNG stream = seed, env_id_base + e) */
    int32_t img_size;          /* S of the S x S images (--image_size, successor_dqn.py:585); 0 = 64.  S < 64: every
                                  image buffer keeps its 64x64 / 64-word layout and the image is its top-left S x S
                                  corner (the rest is zero) */
    double mu, density;        /* assembly_env.py:164 */
    double floor_half_width;   /* assembly_env.py:290-296 */
    double floor_depth;
    double xlim[2], ylim[2];   /* successor_dqn.py:615-616 */
    double targets[BRIDGES_MAX_TARGETS][3];
    uint64_t seed;             /* synthetic random policy */
    const bridges_shape* shapes;   /* HOST pointer, n_shapes entries (copied) */
    const double* x_ground;        /* HOST, n_ground */
    const double* offsets;         /* HOST, n_offsets */
    const double* grid_x;          /* HOST, S: np.linspace(xlim0, xlim1, S) */
    const double* grid_y;          /* HOST, S: np.linspace(ylim1, ylim0, S) */
} bridges_task;

/* Caller-owned device buffers of a vectorised environment.  E = n_envs,
 * K = max_blocks, C = E * a_max (capacity of the compact candidate arrays:
 * candidate a of env e lives at index cand_offset[e] + a). */
typedef struct {
    /* --- state (struct of arrays) --- */
    int32_t* n_blocks;         /* [E] */
    int32_t* blk_shape;        /* [E,K] */
    double* blk_pose;          /* [E,K,4]  x, z, cos, sin */
    double* blk_verts;         /* [E,K,6,2] world (x,z) */
    uint8_t* blk_occ;          /* [E,K] bit f = face f occupied (gym_env.py:228-232 block_graph) */
    uint32_t* targets_left;    /* [E] bit t = target t not reached yet */
    uint64_t* state_bits;      /* [E,64] row r = 64-pixel mask of the state raster */
    int32_t* n_if;             /* [E] contact interfaces */
    int32_t* if_body;          /* [E,MAX_IF,2] body A, body B (-1 = floor) */
    double* if_geom;           /* [E,MAX_IF,8] p_lo.xz, p_hi.xz, n.xz, t.xz */
    uint64_t* draw_counter;    /* [E] policy draws so far */
    uint8_t* needs_reset;      /* [E] */
    /* --- per-step results --- */
    int32_t* sel_index;        /* [E] in: candidate to place */
    uint8_t* step_flags;       /* [E,8] valid_step, stable_frozen, stable_unfrozen, terminated, truncated, done, no_actions,
                                  lp_error (bit 0 solver error, bit 1 contact-list overflow; bit 2 is not an error: the step's
                                  continued tableau said 'unstable' by a small margin and was solved again from scratch;
                                  bit 3: the state has more raw candidates than a_max -- its candidate set was cut to a_max,
                                  something the reference's generate_actions (actions.py:7-52) never does: treat as an error) */
    float* reward;             /* [E] sparse_reward (gym_env.py:11-22) */
    float* lin_reward;         /* [E] successor_dqn.py:397-401 */
    int32_t* n_reached;        /* [E] */
    /* --- candidates of the current state --- */
    int32_t* n_cand;           /* [E] raw candidates of the state, clamped to a_max by reset / step / refresh (a producer that fills
                                  the state arrays itself writes the unclamped count: the clamp is flagged and counted) */
    int32_t* n_valid;          /* [E] */
    int32_t* cand_offset;      /* [E+1] exclusive prefix sum of n_cand */
    int32_t* cand_env;         /* [C] owning env */
    int32_t* cand_desc;        /* [C,4] target_block, target_face, shape, face */
    double* cand_ox;           /* [C] offset_x */
    double* cand_pose;         /* [C,4] */
    double* cand_verts;        /* [C,6,2] */
    double* cand_frames;       /* [C,6,4] world face frames (centre.xz, normal.xz) of the candidate block */
    int32_t* cand_rows;        /* [C,2] rasteriser work descriptor: row_lo | row_hi<<8 | nv<<16 | in_bounds<<24, owning env */
    uint8_t* cand_inb;         /* [C] inside xlim/ylim (gym_env.py:304-323) */
    uint8_t* cand_mask;        /* [C] filter_actions result (actions.py:71-82) */
    float* cand_lin;           /* [C] sum(action_raster * reward_map) */
    uint64_t* cand_bits;       /* [C,64] bit-packed action rasters */
    float* cand_raster;        /* [C,64,64] f32 action rasters (may be NULL: skip) */
    float* state_raster;       /* [E,64,64] f32 (may be NULL) */
    int32_t* cand_raster_nz;   /* [C] or NULL.  Non-NULL = sparse update of cand_raster: bit g = row group g (rows 4g..4g+3)
                                  of slot i holds non-zero pixels.  The rasteriser then writes only the groups that are
                                  non-zero now or were non-zero before; the buffers must start zeroed with zeroed masks. */
    int32_t* state_raster_nz;  /* [E] or NULL: the same for state_raster (both or neither) */
    /* --- task features --- */
    const uint64_t* obstacle_bits; /* [64] */
    const float* reward_map;       /* [64,64] */
    const double* reward_prefix;   /* [64,65] float64 row prefix sums of reward_map: [r][x] = sum of reward_map[r][0..x-1]
                                      (the rasteriser takes a candidate's sum(raster * reward_map) from its row runs) */
    /* --- scratch --- */
    double* lp_ws;             /* [E, lp_ws_stride] per-env persistent simplex tableau (incremental solve of bridges_env_step):
                                  header + basis + two tableau halves; owned by the library between reset and step calls */
    int64_t lp_ws_stride;      /* >= BRIDGES_LP_WS_DOUBLES */
    uint64_t* stats;           /* [16] sum n_cand, sum n_blocks, env-steps, reset-only steps, lp errors, if overflow, lock-steps,
                                  sum n_valid, continued (warm) 'unstable' verdicts solved again from scratch, candidate sets cut to
                                  a_max (must stay 0: size a_max from the task's bound); rest reserved */
    /* --- candidate stability (bridges_env_candidate_stability; all three may be NULL if it is never called) --- */
    uint8_t* cand_stable;      /* [C] 1 = stable, 0 = unstable or masked-out candidate, 2 = solver error / contact overflow */
    int32_t* cand_queue;       /* [C] scratch: candidates whose tableau needs the large workspace */
    int32_t* cand_counters;    /* [4] scratch: queue length, queue head */
    double* cand_ws;           /* [BRIDGES_CAND_WS_SLOTS, cand_ws_stride] scratch: tableaux too large for LDS */
    int64_t cand_ws_stride;    /* >= (3K+2)*(4*MAX_IF+3) */
    double* lp_snap;           /* [E, lp_snap_stride] or NULL: per env the simplex tableau of "last block frozen" of its
                                  current state, written by bridges_env_step; bridges_env_candidate_stability continues every
                                  candidate's LP from it (NULL: candidates are solved from scratch) */
    int64_t lp_snap_stride;    /* >= BRIDGES_LP_SNAP_DOUBLES */
} bridges_env_buffers;

typedef struct bridges_env bridges_env;

const char* bridges_last_error(void);
int bridges_device_count(void);
/* The stamp the library was compiled with: sha256 over csrc/<every file> and include/<every file> (file names and contents,
 * sorted by name; -DBRIDGES_SRC_HASH=... by __graft_entry__.build()), "unstamped" for a build without it.  The Python
 * binding refuses a library whose stamp is not the hash of the sources lying beside it: a stale .so cannot pass for a build. */
const char* bridges_source_hash(void);

/* --- vectorised environment ------------------------------------------------ */
int bridges_env_create(const bridges_task* task, const bridges_env_buffers* buf, bridges_env** out);
int bridges_env_destroy(bridges_env* env);
/* AssemblyGym.reset for every env + first candidate set. */
int bridges_env_reset(bridges_env* env, void* stream);
/* One lock-step: AssemblyGym.step + stabilities_freezing (gym_env.py:218-253, 325-333),
 * sparse_reward, auto-reset, then generate_actions / get_action_features /
 * filter_actions / linear reward for the new state (successor_dqn.py:392-411).
 * Places sel_index[e] for every env. */
int bridges_env_step(bridges_env* env, void* stream);
/* Synthetic uniform-random policy over the valid candidates -> sel_index. */
int bridges_env_select_random(bridges_env* env, void* stream);
/* select_random + step in one call (the synthetic-rollout inner loop). */
int bridges_env_lockstep_random(bridges_env* env, void* stream);
/* Time the dominant kernel (the rasteriser) of the next <= max_launches lock-steps with HIP events recorded on
 * the launch stream; timing_end synchronises on them and returns the summed duration. */
int bridges_env_timing_begin(bridges_env* env, int32_t max_launches);
int bridges_env_timing_end(bridges_env* env, double* raster_ms_total, int32_t* n_launches);
/* Raster gate: environments that share a gate (one per GPU) never run their rasterisers concurrently -- each
 * rasteriser launch waits (hipStreamWaitEvent) for the previous one of the gate.  With several env groups on several
 * streams this serialises the bandwidth-bound kernels while the latency-bound task kernels of the other groups run
 * beside them.  A gate must outlive the environments attached to it. */
typedef struct bridges_gate bridges_gate;
int bridges_gate_create(bridges_gate** out);
int bridges_gate_destroy(bridges_gate* gate);
int bridges_env_set_gate(bridges_env* env, bridges_gate* gate);
/* Candidate refresh only (used after the host edited the state). */
int bridges_env_refresh(bridges_env* env, void* stream);
/* is_action_stable_rbe (assembly_gym/assembly_gym/utils/stability.py:122-130) for EVERY valid candidate of every
 * environment in one pass -> cand_stable[cand_offset[e] + a].  The candidate block is appended to the env's assembly
 * (free; the last placed block stays frozen, gym_env.py:238-240); its contacts are found against the env's persistent
 * contact list, which must be current (i.e. the state was reached through bridges_env_reset / bridges_env_step). */
int bridges_env_candidate_stability(bridges_env* env, void* stream);

/* --- stand-alone operators (same kernels, caller-shaped batches) ------------ */
/* K1: create_block / align_frames_2d (gym_env.py:204-216, geometry.py:39-50).
 * frame1: [n,6] target frame (c.xz, t.xz, n.xz); shape_id,face: [n]; ox,oy: [n]
 * -> pose [n,4], verts [n,6,2]. */
int bridges_place(const bridges_shape* shapes_dev, int32_t n, const double* frame1, const int32_t* shape_id,
                  const int32_t* face, const double* ox, const double* oy, double* pose, double* verts,
                  void* stream);
/* K1 as AssemblyGym.create_block (gym_env.py:204-216): target = floor (target_face[i] < 0) or face target_face[i]
 * of a posed block (target_verts [n,6,2], target_shape [n]).  target_frame_out [n,6] may be NULL. */
int bridges_create_block(const bridges_shape* shapes_dev, int32_t n, const double* target_verts,
                         const int32_t* target_shape, const int32_t* target_face, const int32_t* shape_id,
                         const int32_t* face, const double* ox, const double* oy, double* pose, double* verts,
                         double* target_frame_out, void* stream);
/* Block.__init__ (assembly_env.py:146-153): world vertices of shapes posed by (x, z, cos, sin). */
int bridges_pose_block(const bridges_shape* shapes_dev, int32_t n, const int32_t* shape_id, const double* pose,
                       double* verts, void* stream);
/* Shape.get_face_frame_2d on posed blocks (assembly_env.py:118-124): frames [n,6,6] = centre.xz, x-axis.xz, normal.xz. */
int bridges_face_frames(const bridges_shape* shapes_dev, int32_t n, const int32_t* shape_id, const double* verts,
                        double* frames, void* stream);
/* Shape.contains_2d (assembly_env.py:126-137) of ONE posed block for n arbitrary points (x, z):
 * verts [6,2], points [n,2] f64 -> inside [n] u8. */
int bridges_contains_points(const bridges_shape* shapes_dev, int32_t shape_id, const double* verts, int32_t n,
                            const double* points, uint8_t* inside, void* stream);
/* K4: render_blocks_2d (rendering.py:105-113) of n posed outlines, one image each.
 * verts [n,6,2] world vertices in shape-vertex order, shape_id [n], grid_x/grid_y [64] DEVICE
 * -> bits [n,64] and/or f32 [n,64,64] (either may be NULL). */
int bridges_raster(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                   const double* grid_x, const double* grid_y, uint64_t* bits, float* img, void* stream);
/* render_blocks_2d (rendering.py:105-113) at any image size (the reference's default is 512 x 512): the UNION of n posed
 * outlines on the pixel grid grid_x [W] x grid_y [H] (DEVICE; row 0 = grid_y[0]) -> out [H, W] u8 {0, 1}.  Same pixel test,
 * operation for operation, as the 64-wide rasterisers. */
int bridges_render_blocks(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                          const double* grid_x, int32_t W, const double* grid_y, int32_t H, uint8_t* out, void* stream);
/* The same for S x S images, 2 <= S <= 64 (render_blocks_2d's img_size argument, rendering.py:105): grid_x/grid_y
 * hold S values; the outputs keep the 64-word / 64x64 layout, the image is the top-left S x S corner, the rest 0. */
int bridges_raster_sized(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                         const double* grid_x, const double* grid_y, int32_t size, uint64_t* bits, float* img,
                         void* stream);
/* get_action_features + filter_actions + the rollout's linear reward, fused, for n posed candidate outlines against one
 * state (successor_dqn.py:84-94, 397-401; actions.py:71-82; gym_env.py:304-323): bits [n,64] u64 and / or img [n,64,64] f32
 * (either may be NULL) = the rasters; mask [n] u8 = every vertex inside [xlim, ylim] and above z = 0 (tolerance 1e-6) and
 * no pixel in common with state_bits | obstacle_bits ([64] u64 each, may be NULL = empty); lin_reward [n] f32 =
 * sum(raster * reward_map) taken from reward_prefix ([64,65] f64 row prefix sums of the map, see bridges_env_buffers).
 * size = S of an S x S image (grid_x / grid_y hold S samples).  The lock-step does the same inside bridges_env_step. */
int bridges_action_features(const bridges_shape* shapes_dev, int32_t n, const double* verts, const int32_t* shape_id,
                            const double* grid_x, const double* grid_y, int32_t size, double xlim0, double xlim1, double ylim0,
                            double ylim1, const uint64_t* state_bits, const uint64_t* obstacle_bits, const double* reward_prefix,
                            uint64_t* bits, float* img, uint8_t* mask, float* lin_reward, void* stream);
/* OR-reduce groups of bit rasters: out[g] = OR bits[ranges[g][0] .. ranges[g][1]);  ranges int32 [n_groups,2]. */
int bridges_bits_or(int32_t n_groups, const int32_t* ranges, const uint64_t* bits, uint64_t* out, void* stream);
/* bit raster -> f32 image. */
int bridges_bits_to_f32(int32_t n, const uint64_t* bits, float* img, void* stream);
/* K8: linear layer over flattened binary 64x64 rasters, fed with the bit-packed rasters (replaces the product of
 * SuccessorMLP's first layer with the action / block image, robotoddler/models/cv.py:95-97):
 *   out[r, :] = (base ? base[base_row ? base_row[r] : 0, :] : 0) + sum_{p set in bits[bits_row ? bits_row[r] : r]} wt[p, :]
 * bits [*,64] u64 (row y of an image = one word, bit x = pixel (y, x), flattened pixel p = 64*y + x);
 * wt [4096, d] f32 = the transposed weight slice; base [*, d]; out [n_rows, d]; d % 4 == 0; pixels are added in
 * ascending p, so the result is deterministic. */
int bridges_bits_linear(int32_t n_rows, const uint64_t* bits, const int64_t* bits_row, const float* wt, int32_t d,
                        const float* base, const int64_t* base_row, float* out, void* stream);
/* EpsilonGreedy's count-based exploration (successor_dqn.py:112-131) on bit-packed rasters, for many environments at once:
 * bridges_bits_dot: out[r] = sum(img[slot[r]] * raster(bits[bits_row[r]])) -- the overlap of candidate r with the count image of
 * its episode step (img [n_slots,64,64] f32, slot [n] i64; bits_row NULL = identity);
 * bridges_bits_accumulate: img[slot[r]] += weight[r] * raster(bits[bits_row[r]]) (weight NULL = 1; float atomics on the set
 * pixels).  The count images hold small integers, so both are exact whatever the order of the additions. */
int bridges_bits_dot(int32_t n_rows, const uint64_t* bits, const int64_t* bits_row, const float* img, const int64_t* slot,
                     float* out, void* stream);
int bridges_bits_accumulate(int32_t n_rows, const uint64_t* bits, const int64_t* bits_row, const float* weight,
                            const int64_t* slot, float* img, void* stream);
/* The same head with the product inside: out[r] = sum_j w[j] * sigmoid(h[r,:] . Wd[j,:] + bd[j]), h [n, K = 256] (row stride
 * h_stride floats), Wd [N, 256] row-major (= W_out[px:2px] - W_out[:px] of SuccessorMLP), on the f32 matrix cores; the [n, N]
 * product is never written.  splits > 1 cuts the N columns into that many ranges per 128-row workgroup (so that a launch fills
 * the 256 CUs evenly; the ranges' sums go to part [splits, n] and are added in a fixed order: no atomics, deterministic). */
int bridges_head_sigmoid_dot(int32_t n_rows, int32_t K, int32_t N, const float* h, int64_t h_stride, const float* Wd,
                             const float* bd, const float* w, float* out, float* part, int32_t splits, void* stream);
/* out[r] = sum_j w[j] * sigmoid(d[r * row_stride + j]), j < k: the q head of SuccessorMLP's factored forward
 * (q = sum(softmax(psi)[:, 1] * reward_map), cv.py:101-104, with psi1 - psi0 = d) in one pass.  k % 4 == 0. */
int bridges_sigmoid_dot(int32_t n_rows, const float* d, int64_t row_stride, const float* w, int32_t k, float* out,
                        void* stream);
/* K2+K3: is_stable_rbe (stability.py:49-71) for n independent assemblies given as
 * padded block lists.  verts [n,K,6,2], shape_id [n,K], n_blocks [n], fixed_mask [n] (bit b = block b
 * is_static)
 * -> stable [n] u8, info [n,8] f64 (phase-1 objective, n_interfaces, pivots, error, shader cycles spent in
 *    interface detection, shader cycles spent in the LP, 0, 0).
 * lp_ws: [n, lp_ws_stride] doubles, lp_ws_stride >= 9*MAX_INTERFACES + (3K+2)*(4*MAX_INTERFACES+3). */
int bridges_stability(const bridges_shape* shapes_dev, int32_t n, int32_t K, const double* pose,
                      const double* verts, const int32_t* shape_id, const int32_t* n_blocks,
                      const uint32_t* fixed_mask, double mu, double density, double floor_half_width,
                      double floor_depth, uint8_t* stable, double* info, double* lp_ws, int64_t lp_ws_stride,
                      void* stream);
/* is_stable_rbe_penalty (assembly_gym/assembly_gym/utils/stability.py:75-88; compas_cra rbe_solve(penalty=True) +
 * maximum_tension, assembly_gym/assembly_gym/utils/geometry.py:132-143): contact points may also pull; stable iff some
 * equilibrium keeps the total tension <= tension_tol (> 0; 0 = plain is_stable_rbe).  forces (may be NULL):
 * [n, MAX_INTERFACES, 2, 3] = per contact point (compression c_np, tension c_nn, tangential force) of the equilibrium
 * found, zeros when unstable.  No output of the reference pins this variant ("parity unpinned"). */
int bridges_stability_penalty(const bridges_shape* shapes_dev, int32_t n, int32_t K, const double* pose,
                              const double* verts, const int32_t* shape_id, const int32_t* n_blocks,
                              const uint32_t* fixed_mask, double mu, double density, double floor_half_width,
                              double floor_depth, double tension_tol, uint8_t* stable, double* info, double* forces,
                              double* lp_ws, int64_t lp_ws_stride, void* stream);
/* upload a shape table; returns a device pointer the stand-alone operators take. */
int bridges_shapes_upload(const bridges_shape* shapes_host, int32_t n_shapes, bridges_shape** out_dev);
int bridges_shapes_free(bridges_shape* dev);

/* Environments in the same state: rep[e] = the smallest env index whose state -- n_blocks and the shape id, pose (bit
 * pattern) and face occupancy of its live block slots, plus the caller's flag[e] byte (may be NULL) -- equals env e's word
 * for word (found through a 64-bit hash, hkey [E] scratch, then verified: a hash collision costs the sharing, never
 * correctness).  What a Q-network is asked about a state (its candidate rows, their values) depends on the state alone, so
 * the envs of a group share their representative's rows (bridges_valid_rows with rep).  K = block slots per env (<= 64). */
int bridges_env_groups(int32_t E, int32_t K, const int32_t* n_blocks, const int32_t* blk_shape, const double* blk_pose,
                       const uint8_t* blk_occ, const uint8_t* flag, uint64_t* hkey, int32_t* rep, void* stream);

/* The rows a Q-network is fed (filter_actions, actions.py:71-82, for every env at once): compact indices of the candidates
 * with cand_mask != 0, env-major in candidate order, their owning env, and seg[e] .. seg[e + 1] = the rows of env e
 * (seg [E + 1]; n_valid[e] = the env's count, as bridges_env_step leaves it).  idx / row_env need room for every candidate.
 * With rep (bridges_env_groups; may be NULL) only the envs with rep[e] == e get rows; seg_lo[e] .. seg_hi[e] ([E] each; may
 * both be NULL when rep is) is then the row range of env e's representative -- the rows every env of the group reads.
 * The total goes to *h_total, a HOST-visible word (pinned, device-accessible): valid once the stream has passed the call. */
int bridges_valid_rows(int32_t E, const int32_t* cand_offset, const int32_t* n_cand, const int32_t* n_valid, const uint8_t* cand_mask,
                       const int32_t* rep, int32_t* seg, int32_t* seg_lo, int32_t* seg_hi, int64_t* idx, int64_t* row_env,
                       int32_t* h_total, void* stream);

/* EpsilonGreedy.select (successor_dqn.py:98-132) for every env at once.  Rows seg_lo[e] .. seg_hi[e] of q / join / idx belong to
 * env e (bridges_valid_rows; a prefix-sum array seg is passed as seg, seg + 1); rep (may be NULL): the env whose candidates
 * those rows index (bridges_env_groups).  The env explores when u[e] <= eps and greedy == 0: its row is then the FIRST minimum of join (the
 * overlap of the candidate raster with the count image of the env's episode step), else the FIRST maximum of q.
 * -> sel_compact[e] = idx[row], sel_index[e] = max(sel_compact - cand_offset[rep ? rep[e] : e], 0), q_sel[e] = q[row],
 *    explore_w[e] = 1 if the env explored (the weight of its count-image update) -- an env without rows: idx[0], 0, 0. */
int bridges_eps_greedy_select(int32_t E, int32_t n_rows, const int32_t* seg_lo, const int32_t* seg_hi, const float* q, const float* join,
                              const float* u, float eps, int32_t greedy, const int64_t* idx, const int32_t* cand_offset,
                              const int32_t* rep, int64_t* sel_compact, int32_t* sel_index, float* q_sel, float* explore_w, void* stream);

/* --- transition records of the vectorised loop ------------------------------------------------------------------
 * One float64 row per transition: the compact form of the reference's Transition (successor_dqn.py:27-44) -- the block
 * list of s, the placed block and the scalars; rasters and candidate sets are re-generated when a record is sampled.
 * Integers are stored exactly as float64.  The same rows are what the ranks all-gather per lock-step. */
#define BRIDGES_REC_K 16                                    /* block slots of a record */
#define BRIDGES_REC_NB 0                                    /* n_blocks of s */
#define BRIDGES_REC_SHAPE 1                                 /* [K] shape ids */
#define BRIDGES_REC_POSE (1 + BRIDGES_REC_K)                /* [K][4] (x, z, cos, sin) */
#define BRIDGES_REC_OCC (1 + 5 * BRIDGES_REC_K)             /* [K] face-occupancy bit masks */
#define BRIDGES_REC_ASHAPE (1 + 6 * BRIDGES_REC_K)          /* action: shape id, pose[4], target_block, target_face, face */
#define BRIDGES_REC_APOSE (BRIDGES_REC_ASHAPE + 1)
#define BRIDGES_REC_ATB (BRIDGES_REC_APOSE + 4)
#define BRIDGES_REC_ATF (BRIDGES_REC_APOSE + 5)
#define BRIDGES_REC_AFACE (BRIDGES_REC_APOSE + 6)
#define BRIDGES_REC_REWARD (BRIDGES_REC_AFACE + 1)
#define BRIDGES_REC_LIN (BRIDGES_REC_REWARD + 1)
#define BRIDGES_REC_DONE (BRIDGES_REC_REWARD + 2)           /* terminated | truncated | no next action */
#define BRIDGES_REC_STABLE_S (BRIDGES_REC_REWARD + 3)
#define BRIDGES_REC_STABLE_N (BRIDGES_REC_REWARD + 4)       /* stable(s') with the last block frozen */
#define BRIDGES_REC_TD (BRIDGES_REC_STABLE_N + 1)           /* rollout td_error (successor_dqn.py:413-426), 0 unless prioritised */
#define BRIDGES_REC_WIDTH (BRIDGES_REC_TD + 1)              /* 111 doubles = 888 B */
/* Before the lock-step's step: state s of every env (the env's own arrays, K = its block slots <= 16) and the candidate
 * sel_row[e] (compact row of cand_desc / cand_pose) it is about to place -> rec[e, 0 : REC_REWARD], stable(s); the rest 0. */
int bridges_record_state(int32_t E, int32_t K, const int32_t* n_blocks, const int32_t* blk_shape, const double* blk_pose,
                         const uint8_t* blk_occ, const uint8_t* step_flags, const int64_t* sel_row, const int32_t* cand_desc,
                         const double* cand_pose, double* rec, void* stream);
/* After it: reward, lin_reward, done | no_actions, stable(s') into the same rows; valid[e] = the lock-step was a real
 * env-step of env e (reset-only lock-steps are not transitions). */
int bridges_record_result(int32_t E, const float* reward, const float* lin_reward, const uint8_t* step_flags, double* rec,
                          uint8_t* valid, void* stream);
/* Sampled records -> the state arrays of a replay env of E >= n_rec envs (envs >= n_rec repeat record 0): s' = s plus the
 * action block with the occupancy update of gym_env.py:228-232, its RAW candidate count
 * n_groups * (n_ground + free faces * n_off) (generate_actions, actions.py:7-52; bridges_env_refresh clamps it to the env's
 * a_max and flags / counts the truncation), the block ranges [e*K, e*K + n) of s' and of s (for bridges_bits_or over the
 * per-block rasters) and the scalars the targets need. */
int bridges_replay_unpack(int32_t E, int32_t n_rec, int32_t K, const double* rec, const int32_t* shape_faces, int32_t n_shapes,
                          int32_t n_groups, int32_t n_ground, int32_t n_off, int32_t* n_blocks, int32_t* blk_shape,
                          double* blk_pose, uint8_t* blk_occ, int32_t* n_cand, int32_t* ranges_next, int32_t* ranges_prev,
                          float* lin, float* stable_s, uint8_t* done, uint8_t* stable_n, void* stream);

/* --- K7: DQN ops (robotoddler/training/successor_dqn.py) --------------------- */
/* update_target_net (successor_dqn.py:280-288): target = policy*tau + target*one_minus_tau, the two products and the
 * sum rounded separately in float32 as torch evaluates it (one_minus_tau = (float)(1.0 - tau) from the host). */
int bridges_soft_update(float* target, const float* policy, int64_t n, float tau, float one_minus_tau, void* stream);
/* train_policy_net target construction (successor_dqn.py:197-213, 222, 230):
 * per transition i with rows [seg_lo[i], seg_hi[i]) of the target net's output (a prefix-sum array: seg, seg + 1;
 * transitions whose next states are the same state may share a range, bridges_valid_rows with rep):
 *   j* = argmax next_q (first maximum), q_target[i] = lin_reward[i] + gamma * (done ? 0 : next_q[j*]),
 *   sf_target[i,:] = action_raster[i,:] + gamma * (done ? 0 : next_sf[j*,:])   (sf_dim may be 0). */
int bridges_td_target(int32_t n_trans, const int32_t* seg_lo, const int32_t* seg_hi, const float* next_q, const float* next_sf,
                      int64_t next_sf_row_stride, const float* action_raster, const float* lin_reward,
                      const uint8_t* done, float gamma, int32_t sf_dim, float* q_target, float* sf_target,
                      int32_t* argmax_row, void* stream);

/* Inference epilogues of the conv Q-networks (robotoddler/models/cv.py:5-17, 41-73, 108-170: Conv2d -> ReLU
 * [-> MaxPool2d(2)]), one pass instead of torch's three; bit-identical to relu(conv + bias) / maxpool(relu(conv + bias)).
 * x [n, C, hw] f32 contiguous NCHW output of a bias-free convolution, updated in place (hw % 4 == 0). */
int bridges_bias_relu(float* x, const float* bias, int64_t n, int32_t C, int32_t hw, void* stream);
/* x [n, C, H, W] -> out [n, C, H/2, W/2] (W % 4 == 0, H % 2 == 0). */
int bridges_bias_relu_pool2(const float* x, const float* bias, float* out, int64_t n, int32_t C, int32_t H, int32_t W,
                            void* stream);

/* ---- one optimiser step of SuccessorMLP at replay-batch size (robotoddler/models/cv.py:76-105;
 * train_policy_net, robotoddler/training/successor_dqn.py:157-235: forward, MSE losses, backward) on the f32 matrix
 * cores.  All tensors f32, row-major, contiguous; `rows` = the batch padded to a multiple of 32 (padding rows zero).
 * `ws` = scratch for split partial sums (ws_floats floats; the split count adapts to it). */
/* y [rows,N] = act(x [rows,K] . W [N,K]^T + bias), act = ReLU if relu else identity (nn.Linear [+ nn.ReLU], cv.py:20-38). */
int bridges_linear_forward(int32_t rows, int32_t K, int32_t N, const float* x, const float* W, const float* bias,
                           int32_t relu, float* y, float* ws, int64_t ws_floats, const int64_t* x_block, void* stream);
/* Backward of the same layer from dz [rows,N] (gradient at its pre-activation) and its input a_in [rows,K]:
 * dW [N,K] = dz^T . a_in, db [N] = column sums of dz, and -- unless dz_below is NULL (first layer) --
 * dz_below [rows,K] = (dz . W) masked by act_below > 0 (act_below = the ReLU output that was this layer's input; NULL =
 * no mask). */
int bridges_linear_backward(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, const float* W,
                            float* dW, float* db, const float* act_below, float* dz_below, float* ws, int64_t ws_floats,
                            const int64_t* a_block, int32_t a_block_bias, void* stream);
/* The middle Linear + ReLU layers of the step in ONE launch each way and without traffic between workgroups: a 1024-thread
 * workgroup per 32-column tile of the stack's last layer (backward: of the gradient handed below the stack) computes the
 * layers that tile depends on itself, in full, handing activations / gradients from layer to layer through LDS; every weight
 * fragment is requested before the first MFMA.  Built for the reference's stack (successor_dqn.py:366): n_layers = 4,
 * dims = {256, 128, 64, 128, 256} (layer l: dims[l] -> dims[l + 1]), one 32-row batch tile; bridges_mlp_mid_supported says
 * whether a shape is (1 / 0), anything else is refused by the two calls.
 *   forward:  acts[l + 1] = relu(acts[l] . W[l]^T + bias[l])                      (acts: 5 arrays [32, dims[l]])
 *   backward: dW[l] = dz[l + 1]^T . acts[l], db[l] = column sums of dz[l + 1], dz[0] = the gradient handed below the stack
 *             (dz: 5 pointers like acts; dz[4] given, dz[0] written, dz[1..3] live in LDS only and are not touched)
 * dims and the pointer tables are HOST arrays.  Bit-identical to four calls of bridges_linear_forward / _backward.
 * The backward launch can carry an Adam update of ANOTHER layer's flat range (rest_*, as bridges_linear_backward_adam's; rest_n
 * = 0: none): the head's, whose gradient is complete and whose weights nothing reads any more -- extra workgroups beside the
 * stack's eight, so that traffic is out of the step's last launch. */
int bridges_mlp_mid_supported(int32_t rows, int32_t n_layers, const int32_t* dims);
int bridges_mlp_mid_forward(int32_t rows, int32_t n_layers, const int32_t* dims, const float* const* W, const float* const* bias,
                            float* const* acts, void* stream);
int bridges_mlp_mid_backward(int32_t rows, int32_t n_layers, const int32_t* dims, const float* const* W, float* const* dW,
                             float* const* db, float* const* acts, float* const* dz, float* rest_param, const float* rest_grad,
                             float* rest_exp_avg, float* rest_exp_avg_sq, int64_t rest_n, const float* step, double lr, double beta1,
                             double beta2, double eps, void* stream);
/* The same stack for MANY rows -- the acting / target forward between the first layer and the head (cv.py:20-38 inside
 * SuccessorMLP.forward): y [n, 256] = the four Linear + ReLU layers applied to relu(x), x [n, 256] the pre-activation of the
 * layer in front (row strides in floats); mid = scratch of n x 64 floats (the narrow activation between the two launches).
 * Two launches of two layers each instead of four library GEMMs; the arithmetic of the training step's forward, tile by tile.
 * dims / W / bias as for bridges_mlp_mid_forward. */
int bridges_mlp_mid_rows(int32_t n_rows, int32_t n_layers, const int32_t* dims, const float* const* W, const float* const* bias,
                         const float* x, int64_t x_stride, float* y, int64_t y_stride, float* mid, void* stream);
/* bridges_linear_backward of the LAST layer with the step's book-keeping riding along (one thread of the launch, beside its
 * jobs): losses[*counter] = sum of loss_rows[0 .. batch) in row order, ++*counter, ++*adam_step (adam_step may be NULL).  The
 * loss rows are those bridges_successor_loss (called with ticket = NULL, losses = NULL) has just written; nothing between the
 * two calls may read *counter.  No x_block / a_block indirection (the head's input is the stack's output). */
int bridges_linear_backward_log(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, const float* W,
                                float* dW, float* db, const float* act_below, float* dz_below, float* ws, int64_t ws_floats,
                                const float* loss_rows, int32_t batch, float* losses, int32_t n_losses, int64_t* counter,
                                float* adam_step, void* stream);
/* Backward of a Linear layer that needs no input gradient (the first layer) with the optimiser update inside: W, bias and
 * their moments are updated in place from the weight-gradient tiles in the matrix-core accumulators (that gradient is never
 * written; rows must be 32: one batch tile), and extra workgroups of the same launch apply Adam to `rest_n` further
 * parameters (the other layers: one contiguous, 16-byte-aligned range of flat parameter / gradient / moment buffers, rest_n
 * a multiple of 4; 0 = none) -- valid because this is the last backward launch of a step.  *step as in bridges_adam_step. */
int bridges_linear_backward_adam(int32_t rows, int32_t K, int32_t N, const float* dz, const float* a_in, float* W, float* bias,
                                 float* exp_avg_w, float* exp_avg_sq_w, float* exp_avg_b, float* exp_avg_sq_b, float* rest_param,
                                 const float* rest_grad, float* rest_exp_avg, float* rest_exp_avg_sq, int64_t rest_n, const float* step,
                                 double lr, double beta1, double beta2, double eps, const int64_t* a_block, int32_t a_block_bias,
                                 void* stream);
/* Input rows of replay batch *counter: x [rows, 4 px + nf] = [block | action | reward | obstacle | binary]
 * (cv.py:100-103) from block_all / action_all [n,px], binary_all [n,nf] (row *counter * batch + b), reward / obstacle [px]. */
int bridges_mlp_input(int32_t batch, int32_t rows, int32_t px, int32_t nf, const int64_t* counter, const float* block_all,
                      const float* action_all, const float* binary_all, const float* reward, const float* obstacle,
                      float* x, void* stream);
/* The same rows for ALL n_batches batches of a train_policy_net call in one launch: x_all [n_batches * rows, 4 px + nf], batch c in
 * rows [c * rows, (c + 1) * rows).  bridges_linear_forward / _backward / _backward_adam take a device word `x_block` /
 * `a_block` (may be NULL = 0) that selects the block of such an array (x + *x_block * rows * K; the backward entry points add the
 * host constant a_block_bias to the word: -1 when the step's loss kernel has advanced the counter in between), so a replayed
 * graph of one optimiser step reads batch *counter of the pre-built inputs instead of building its rows first. */
int bridges_mlp_input_batches(int32_t n_batches, int32_t batch, int32_t rows, int32_t px, int32_t nf, const float* block_all,
                              const float* action_all, const float* binary_all, const float* reward, const float* obstacle,
                              float* x_all, void* stream);
/* Head + loss + its gradient (cv.py:104-108; successor_dqn.py:215-232): y [rows, 2 px + 2 nf] = (psi0 | psi1 | binary),
 * q = sum_j softmax(psi)[1][j] * reward[j], loss = [use_q] mean (q - q_target)^2 + [use_sf] mean (psi0 - sf_target)^2
 * with the targets of batch *counter (q_target_all [n], sf_target_all [n,px]).  Writes dy [rows, 2 px + 2 nf],
 * loss_rows [rows] (per-row share of the loss), q_out [rows]; if losses != NULL: losses[*counter_inc] = sum of
 * loss_rows and ++*counter_inc (the step counter that selects the next batch). */
int bridges_successor_loss(int32_t batch, int32_t rows, int32_t px, int32_t nf, const float* y, const float* reward,
                           const int64_t* counter, const float* q_target_all, const float* sf_target_all, int32_t use_q,
                           int32_t use_sf, float* dy, float* loss_rows, float* q_out, float* losses, int32_t n_losses,
                           int64_t* counter_inc, int32_t* ticket, float* adam_step, void* stream);
/* `ticket` (device int32, zero before the first call; may be NULL): the logging (losses[*counter_inc] = sum of loss_rows,
 * ++*counter_inc, and ++*adam_step if given) is done inside the loss kernel by the row workgroup that arrives last
 * instead of a second launch; the kernel re-arms the ticket.
 * One Adam step of torch.optim.Adam (successor_dqn.py:640; amsgrad / weight decay / maximize off) over a flat float32
 * buffer of n parameters with their gradients and moments; *step (device float) is the step number of THIS update,
 * already incremented by the caller. */
int bridges_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* step,
                      double lr, double beta1, double beta2, double eps, void* stream);
/* The same update over many tensors in ONE launch (a conv Q-network's 34-60 parameter tensors with their autograd gradients):
 * slots [n_slots] (device) name the tensors; chunk c of the launch updates elements [chunk_off[c] * 1024, + 1024) of tensor
 * chunk_slot[c] (device int32 arrays, n_chunks entries: every tensor cut into 1024-element chunks).  *step (device float) =
 * the number of updates done SO FAR: this update is number *step + 1; the launch does not write *step (advance it after the
 * call), it writes the new count into every slot's own `step` word when that is not NULL (torch.optim.Adam's state['step']). */
typedef struct bridges_adam_slot {
    float* p; const float* g; float* m; float* v; float* step; int64_t n;
} bridges_adam_slot;
int bridges_adam_multi(const bridges_adam_slot* slots, int32_t n_slots, const int32_t* chunk_slot, const int32_t* chunk_off,
                       int32_t n_chunks, const float* step, double lr, double beta1, double beta2, double eps, void* stream);

/* relu(conv3x3(x, w, padding 1) + bias) [then MaxPool2d(2)] for the 64-pixel-wide layers with 16 output channels of
 * the conv Q-networks (cv.py:5-17 ConvBlock(4,16) / (16,16); cv.py:138-254 UNet e11, e12, d41, d42), inference passes:
 * x [n, c_in, H, 64] f32 NCHW contiguous, w [16, c_in, 3, 3], bias [16] -> out [n, 16, H, 64] (pool: [n, 16, H/2, 32]).
 * c_in in {1..4, 16, 32}, H % 8 == 0.  Same function as torch's conv2d + relu (+ max_pool2d) up to f32 summation order. */
int bridges_conv3x3_relu_o16(const float* x, const float* w, const float* bias, float* out, int64_t n, int32_t c_in,
                             int32_t H, int32_t W, int32_t pool, void* stream);
/* The same kernel with the U-Net's neighbours folded in (cv.py:184-197):
 *   x2 != NULL: the 32 input channels are torch.cat([x, x2], dim=1) of two 16-channel tensors (decoder: up-convolution
 *               output + skip tensor), never materialised;
 *   mode 0 plain, 1 pooled, 2 both (out = relu(conv) [n,16,H,64] AND out2 = its MaxPool2d(2) [n,16,H/2,32]: the encoder's
 *        skip tensor and next level), 3 projection (out [n,1,H,64] = sum_c proj_w[c] * relu(conv)_c + proj_b[0]: the
 *        1x1 outconv to one channel behind the last decoder convolution). */
int bridges_conv3x3_relu_o16_ex(const float* x, const float* x2, const float* w, const float* bias, float* out, float* out2,
                                const float* proj_w, const float* proj_b, int64_t n, int32_t c_in, int32_t c_in2, int32_t H,
                                int32_t W, int32_t mode, void* stream);
/* ConvTranspose2d(kernel_size=2, stride=2) + bias of the U-Net decoder (cv.py:176, 179 upconv3 / upconv4), inference:
 * x [n, c_in, H, W] f32 NCHW (W % 16 == 0), w [c_in, c_out, 2, 2], bias [c_out] -> out [n, c_out, 2H, 2W];
 * (c_in, c_out) = (32, 16) or (64, 32). */
int bridges_upconv2x2(const float* x, const float* w, const float* bias, float* out, int64_t n, int32_t c_in, int32_t c_out,
                      int32_t H, int32_t W, void* stream);

/* --- K11: the ConvBlock stacks (robotoddler/models/cv.py:5-17, 41-73) in the TRAINING passes of train_policy_net
 * (successor_dqn.py:157-277), on the f32 matrix cores.  Square W x W float32 NCHW images, W in {8, 16, 32, 64}; C_out a multiple
 * of 16.
 * bridges_conv3x3: out [n, c_out, W, W] = conv3x3(x [n, c_in, W, W], padding 1) with
 *   mode 0: nothing else;  1: + bias, ReLU;  2: x [mask > 0] (mask [n, c_out, W, W]: the ReLU of the layer whose output the
 *   gradient flows into);
 *   transposed 0: w = Conv2d.weight [c_out, c_in, 3, 3];  1: the INPUT GRADIENT of a layer with weight w [c_in, c_out, 3, 3]
 *   (x = the gradient at that layer's output with c_in channels, out = the gradient at its input with c_out channels);
 *   in_mask (may be NULL; shape of x): x counts only where in_mask > 0 -- x = the gradient at a ReLU's output, in_mask = that
 *   ReLU's activation (likewise g_mask of bridges_conv3x3_wgrad).
 * bridges_conv3x3_wgrad: dw [c_out, c_in, 3, 3] and db [c_out] from g [n, c_out, W, W] (gradient at the layer's output, ReLU
 *   mask applied) and the layer's input x [n, c_in, W, W]; partial sums over pixel ranges are added in a fixed order
 *   (deterministic); scratch: bridges_conv3x3_wgrad_scratch floats.
 * Image tensors (x, in_mask, mask, out, g, g_mask) must be 16-byte aligned: rows are read and written as float4.
 * bridges_maxpool2 / bridges_maxpool2_relu_backward: MaxPool2d(2) of a [nc, H, W] and, from dy [nc, H/2, W/2], the gradient at
 *   the pre-pool activation a = relu(.): the first maximum of a window in scan order takes dy (as torch), times [a > 0]. */
int bridges_conv3x3(const float* x, const float* in_mask, const float* w, const float* bias, const float* mask, float* out, int64_t n,
                    int32_t c_in, int32_t c_out, int32_t W, int32_t mode, int32_t transposed, void* stream);
/* bridges_conv3x3_wgrad with dw == db == NULL leaves the partial sums in scratch ([splits][c_out * c_in * 9] then
 * [splits][c_out], splits = scratch floats / (c_out * c_in * 9 + c_out)); bridges_reduce_jobs then adds the partial sums of
 * SEVERAL layers in one launch: job j = workgroups [block_start, block_start + B_j) of total_blocks with B_j = ceil((n_w + n_b) /
 * 256) when splits <= 16 and ceil((n_w + n_b) / 16) otherwise; same arithmetic and order as the single-layer form (jobs_dev:
 * device array, ascending block_start). */
typedef struct bridges_reduce_job {
    const float* part; const float* part_b; float* dw; float* db; int32_t n_w; int32_t n_b; int32_t splits; int32_t block_start;
} bridges_reduce_job;
int bridges_reduce_jobs(const bridges_reduce_job* jobs_dev, int32_t n_jobs, int32_t total_blocks, void* stream);
int bridges_conv3x3_wgrad_scratch(int64_t n, int32_t c_in, int32_t c_out, int32_t W, int64_t* floats);
int bridges_conv3x3_wgrad(const float* g, const float* g_mask, const float* x, float* dw, float* db, float* scratch, int64_t scratch_floats,
                          int64_t n, int32_t c_in, int32_t c_out, int32_t W, void* stream);
int bridges_maxpool2(const float* a, float* y, int64_t nc, int32_t H, int32_t W, void* stream);
/* Backward of the U-Net decoder's ConvTranspose2d(kernel_size=2, stride=2) (cv.py:176, 179; forward: bridges_upconv2x2): from
 * the layer's input x [n, c_in, H, W], the gradient g [n, c_out, 2H, 2W] at its output and w [c_in, c_out, 2, 2] ->
 * dx [n, c_in, H, W] (may be NULL), dw [c_in, c_out, 2, 2], db [c_out]; (c_in, c_out) = (32, 16) or (64, 32), H * W a multiple
 * of 64; partial sums per workgroup added in a fixed order (deterministic).  scratch: bridges_upconv2x2_backward_scratch floats.
 * bridges_conv1x1_o1_*: Conv2d(c_in, 1, kernel_size=1) (cv.py:182 outconv of UNet(1)): y [n, hw] = b + sum_c x[n, c, hw] w[c];
 * backward: dx = g w[c], dw [c_in], db [1]; c_in <= 32, hw a multiple of 4; scratch: min(256, ceil(n * hw / 1024)) * (c_in + 1).
 * Both backward entry points leave the partial sums in scratch ([splits][n_w] then [splits][n_b]) when dw == db == NULL, for
 * bridges_reduce_jobs. */
int bridges_upconv2x2_backward_scratch(int64_t n, int32_t c_in, int32_t c_out, int32_t H, int32_t W, int64_t* floats);
int bridges_upconv2x2_backward(const float* x, const float* g, const float* w, float* dx, float* dw, float* db, float* scratch,
                               int64_t scratch_floats, int64_t n, int32_t c_in, int32_t c_out, int32_t H, int32_t W, void* stream);
int bridges_conv1x1_o1_forward(const float* x, const float* w, const float* bias, float* y, int64_t n, int32_t c_in, int32_t hw, void* stream);
int bridges_conv1x1_o1_backward(const float* x, const float* g, const float* w, float* dx, float* dw, float* db, float* scratch,
                                int64_t scratch_floats, int64_t n, int32_t c_in, int32_t hw, void* stream);
/* db [C] = sum over n and the hw pixels of g [n, C, hw] in a fixed order (deterministic, single-workgroup stages): the bias
 * gradient of a convolution whose weights stay with the library.  scratch: min(n, 32) * C floats. */
int bridges_bias_grad(const float* g, float* db, float* scratch, int64_t scratch_floats, int64_t n, int32_t C, int32_t hw, void* stream);
int bridges_maxpool2_relu_backward(const float* a, const float* dy, float* g, int64_t nc, int32_t H, int32_t W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BRIDGES_HIP_H */

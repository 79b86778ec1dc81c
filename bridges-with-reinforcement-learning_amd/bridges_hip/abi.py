"""ctypes mirror of include/bridges_hip.h and loader of libbridges_hip.so.

The library is the product: if it is missing or no HIP device is present the
entry points raise ``BridgesHipError`` -- there is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BRIDGES_LIB", os.path.join(HERE, "libbridges_hip.so"))      # override: diagnostic builds
CSRC_DIR = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
STAMP_PREFIX = "BRIDGES_SRC_HASH="


def source_hash():
    """sha256 over the library's sources: every file of csrc/ and include/ (relative name + contents, sorted by name).
    __graft_entry__.build() compiles it into the library (bridges_source_hash()); lib() compares."""
    import hashlib
    h = hashlib.sha256()
    for tag, d in (("csrc", CSRC_DIR), ("include", INCLUDE_DIR)):
        for name in sorted(os.listdir(d)):
            path = os.path.join(d, name)
            if os.path.isfile(path) and not name.startswith("."):
                h.update(f"{tag}/{name}\0".encode())
                with open(path, "rb") as fh:
                    h.update(fh.read())
                h.update(b"\0")
    return h.hexdigest()


def library_stamp(path=None):
    """The stamp inside a built library file, read from its bytes without loading it (None: no such file / no stamp)."""
    import re
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        m = re.search(rb"BRIDGES_SRC_HASH=([0-9a-f]{64}|unstamped)", fh.read())
    return m.group(1).decode() if m else None

MAX_VERTS = 6
MAX_BLOCKS = 16
MAX_GROUPS = 24
MAX_TARGETS = 8
MAX_INTERFACES = 64


def lp_ws_stride(max_blocks):
    """Doubles of simplex workspace per assembly of the stand-alone stability operator (bridges_stability) and per
    slot of the candidate-stability scratch: interface list + the largest tableau, (3K equilibrium rows + budget +
    cost) x (4*MAX_IF generators + slack + rhs, padded to an odd stride)."""
    return 9 * MAX_INTERFACES + (3 * max_blocks + 2) * (4 * MAX_INTERFACES + 3)


# include/bridges_hip.h: BRIDGES_LP_WS_DOUBLES -- the per-env persistent tableau of the incremental simplex
ENV_LP_WS_DOUBLES = 64 + 2 * (3 * MAX_BLOCKS + 2) * (4 * MAX_INTERFACES + 2 + 3 * MAX_BLOCKS + 1)
CAND_WS_SLOTS = 1024


IMG = 64

FLAG_NAMES = ("valid_step", "stable_frozen", "stable_unfrozen", "terminated", "truncated", "done",
              "no_actions", "lp_error")
STAT_NAMES = ("sum_cand", "sum_blocks", "env_steps", "reset_only", "lp_errors", "if_overflow", "locksteps",
              "sum_valid", "warm_resolved", "cand_overflow")


class BridgesHipError(RuntimeError):
    pass


class Shape(C.Structure):
    _fields_ = [
        ("nv", C.c_int32), ("pad_", C.c_int32),
        ("vx", C.c_double * MAX_VERTS), ("vz", C.c_double * MAX_VERTS),
        ("fa", C.c_int32 * MAX_VERTS), ("fb", C.c_int32 * MAX_VERTS),
        ("fcx", C.c_double * MAX_VERTS), ("fcz", C.c_double * MAX_VERTS),
        ("fnx", C.c_double * MAX_VERTS), ("fnz", C.c_double * MAX_VERTS),
        ("depth", C.c_double), ("volume", C.c_double), ("gx", C.c_double), ("gz", C.c_double),
    ]


class Task(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("max_blocks", C.c_int32), ("max_steps", C.c_int32), ("a_max", C.c_int32),
        ("n_shapes", C.c_int32), ("n_groups", C.c_int32),
        ("group_shape", C.c_int32 * MAX_GROUPS), ("group_face", C.c_int32 * MAX_GROUPS),
        ("n_ground", C.c_int32), ("n_offsets", C.c_int32), ("n_targets", C.c_int32), ("debug", C.c_int32),
        ("env_id_base", C.c_int32), ("img_size", C.c_int32),
        ("mu", C.c_double), ("density", C.c_double),
        ("floor_half_width", C.c_double), ("floor_depth", C.c_double),
        ("xlim", C.c_double * 2), ("ylim", C.c_double * 2),
        ("targets", (C.c_double * 3) * MAX_TARGETS),
        ("seed", C.c_uint64),
        ("shapes", C.POINTER(Shape)),
        ("x_ground", C.POINTER(C.c_double)), ("offsets", C.POINTER(C.c_double)),
        ("grid_x", C.POINTER(C.c_double)), ("grid_y", C.POINTER(C.c_double)),
    ]


# (field name, torch dtype name, shape expression in E, K, C) -- order == bridges_env_buffers
ENV_BUFFER_FIELDS = [
    ("n_blocks", "int32", "E"),
    ("blk_shape", "int32", "E,K"),
    ("blk_pose", "float64", "E,K,4"),
    ("blk_verts", "float64", "E,K,6,2"),
    ("blk_occ", "uint8", "E,K"),
    ("targets_left", "int32", "E"),          # uint32 on the device
    ("state_bits", "int64", "E,64"),         # uint64 on the device
    ("n_if", "int32", "E"),
    ("if_body", "int32", "E,IF,2"),
    ("if_geom", "float64", "E,IF,8"),
    ("draw_counter", "int64", "E"),
    ("needs_reset", "uint8", "E"),
    ("sel_index", "int32", "E"),
    ("step_flags", "uint8", "E,8"),
    ("reward", "float32", "E"),
    ("lin_reward", "float32", "E"),
    ("n_reached", "int32", "E"),
    ("n_cand", "int32", "E"),
    ("n_valid", "int32", "E"),
    ("cand_offset", "int32", "E1"),
    ("cand_env", "int32", "C"),
    ("cand_desc", "int32", "C,4"),
    ("cand_ox", "float64", "C"),
    ("cand_pose", "float64", "C,4"),
    ("cand_verts", "float64", "C,6,2"),
    ("cand_frames", "float64", "C,6,4"),
    ("cand_rows", "int32", "C,2"),
    ("cand_inb", "uint8", "C"),
    ("cand_mask", "uint8", "C"),
    ("cand_lin", "float32", "C"),
    ("cand_bits", "int64", "C,64"),
    ("cand_raster", "float32", "C,64,64"),
    ("state_raster", "float32", "E,64,64"),
    ("cand_raster_nz", "int32", "C"),
    ("state_raster_nz", "int32", "E"),
    ("obstacle_bits", "int64", "64"),
    ("reward_map", "float32", "64,64"),
    ("reward_prefix", "float64", "64,65"),
    ("lp_ws", "float64", "E,WS"),
]


# buffers that follow lp_ws_stride / stats in bridges_env_buffers
ENV_BUFFER_FIELDS_TAIL = [
    ("cand_stable", "uint8", "C"),
    ("cand_queue", "int32", "C"),
    ("cand_counters", "int32", "4"),
    ("cand_ws", "float64", "CWS,WSC"),
]
ENV_LP_SNAP_DOUBLES = 64 + (3 * MAX_BLOCKS + 2) * (4 * MAX_INTERFACES + 2 + 3 * MAX_BLOCKS + 1)


class EnvBuffers(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in ENV_BUFFER_FIELDS] + [
        ("lp_ws_stride", C.c_int64), ("stats", C.c_void_p)] + [(name, C.c_void_p) for name, _, _ in ENV_BUFFER_FIELDS_TAIL] + [
        ("cand_ws_stride", C.c_int64), ("lp_snap", C.c_void_p), ("lp_snap_stride", C.c_int64)]


# put lp_ws_stride right after lp_ws as in the header (fields above are already in header order)
assert [f[0] for f in EnvBuffers._fields_][-10:-7] == ["lp_ws", "lp_ws_stride", "stats"]

_lib = None


def lib():
    """Load libbridges_hip.so (once).  Raises BridgesHipError if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BridgesHipError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_float
    try:
        L.bridges_source_hash.restype = C.c_char_p
        stamp = L.bridges_source_hash().decode()[len(STAMP_PREFIX):]
    except AttributeError:
        stamp = "missing"
    want = source_hash()
    if stamp != want:
        raise BridgesHipError(f"{LIB_PATH} is stale: it was built from other sources (stamp {stamp[:16]}, sources {want[:16]}); "
                              "rebuild it with `python __graft_entry__.py build`")
    L.bridges_last_error.restype = C.c_char_p
    L.bridges_device_count.restype = C.c_int
    sigs = {
        "bridges_env_create": [C.POINTER(Task), C.POINTER(EnvBuffers), C.POINTER(vp)],
        "bridges_env_destroy": [vp],
        "bridges_env_reset": [vp, vp],
        "bridges_env_step": [vp, vp],
        "bridges_env_select_random": [vp, vp],
        "bridges_env_lockstep_random": [vp, vp],
        "bridges_env_refresh": [vp, vp],
        "bridges_env_candidate_stability": [vp, vp],
        "bridges_gate_create": [C.POINTER(vp)],
        "bridges_gate_destroy": [vp],
        "bridges_env_set_gate": [vp, vp],
        "bridges_env_timing_begin": [vp, i32],
        "bridges_env_timing_end": [vp, C.POINTER(C.c_double), C.POINTER(i32)],
        "bridges_shapes_upload": [C.POINTER(Shape), i32, C.POINTER(vp)],
        "bridges_shapes_free": [vp],
        "bridges_place": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_create_block": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_pose_block": [vp, i32, vp, vp, vp, vp],
        "bridges_face_frames": [vp, i32, vp, vp, vp, vp],
        "bridges_contains_points": [vp, i32, vp, i32, vp, vp, vp],
        "bridges_raster": [vp, i32, vp, vp, vp, vp, vp, vp, vp],
        "bridges_raster_sized": [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp],
        "bridges_render_blocks": [vp, i32, vp, vp, vp, i32, vp, i32, vp, vp],
        "bridges_bits_or": [i32, vp, vp, vp, vp],
        "bridges_action_features": [vp, i32, vp, vp, vp, vp, i32, f64, f64, f64, f64, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_bits_to_f32": [i32, vp, vp, vp],
        "bridges_bits_linear": [i32, vp, vp, vp, i32, vp, vp, vp, vp],
        "bridges_sigmoid_dot": [i32, vp, i64, vp, i32, vp, vp],
        "bridges_bits_dot": [i32, vp, vp, vp, vp, vp, vp],
        "bridges_head_sigmoid_dot": [i32, i32, i32, vp, i64, vp, vp, vp, vp, vp, i32, vp],
        "bridges_linear_backward_log": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, vp, i32, vp, vp, vp],
        "bridges_mlp_mid_rows": [i32, i32, vp, vp, vp, vp, i64, vp, i64, vp, vp],
        "bridges_mlp_mid_supported": [i32, i32, vp],
        "bridges_mlp_mid_forward": [i32, i32, vp, vp, vp, vp, vp],
        "bridges_mlp_mid_backward": [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, vp],
        "bridges_eps_greedy_select": [i32, i32, vp, vp, vp, vp, vp, C.c_float, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_valid_rows": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_env_groups": [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_record_state": [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_record_result": [i32, vp, vp, vp, vp, vp, vp],
        "bridges_replay_unpack": [i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_bits_accumulate": [i32, vp, vp, vp, vp, vp, vp],
        "bridges_stability": [vp, i32, i32, vp, vp, vp, vp, vp, f64, f64, f64, f64, vp, vp, vp, i64, vp],
        "bridges_stability_penalty": [vp, i32, i32, vp, vp, vp, vp, vp, f64, f64, f64, f64, f64, vp, vp, vp, vp, i64, vp],
        "bridges_soft_update": [vp, vp, i64, f32, f32, vp],
        "bridges_bias_relu": [vp, vp, i64, i32, i32, vp],
        "bridges_bias_relu_pool2": [vp, vp, vp, i64, i32, i32, i32, vp],
        "bridges_conv3x3_relu_o16": [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp],
        "bridges_conv3x3_relu_o16_ex": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp],
        "bridges_upconv2x2": [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp],
        "bridges_linear_forward": [i32, i32, i32, vp, vp, vp, i32, vp, vp, i64, vp, vp],
        "bridges_linear_backward": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, vp],
        "bridges_mlp_input": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "bridges_mlp_input_batches": [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
        "bridges_successor_loss": [i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp],
        "bridges_adam_step": [vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, vp],
        "bridges_linear_backward_adam": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, vp, i32, vp],
        "bridges_conv3x3": [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp],
        "bridges_conv3x3_wgrad_scratch": [i64, i32, i32, i32, C.POINTER(i64)],
        "bridges_conv3x3_wgrad": [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp],
        "bridges_maxpool2": [vp, vp, i64, i32, i32, vp],
        "bridges_bias_grad": [vp, vp, vp, i64, i64, i32, i32, vp],
        "bridges_adam_multi": [vp, i32, vp, vp, i32, vp, f64, f64, f64, f64, vp],
        "bridges_reduce_jobs": [vp, i32, i32, vp],
        "bridges_upconv2x2_backward_scratch": [i64, i32, i32, i32, i32, C.POINTER(i64)],
        "bridges_upconv2x2_backward": [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp],
        "bridges_conv1x1_o1_forward": [vp, vp, vp, vp, i64, i32, i32, vp],
        "bridges_conv1x1_o1_backward": [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp],
        "bridges_maxpool2_relu_backward": [vp, vp, vp, i64, i32, i32, vp],
        "bridges_td_target": [i32, vp, vp, vp, vp, i64, vp, vp, vp, f32, i32, vp, vp, vp, vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "bridges_last_error", "bridges_device_count", "bridges_source_hash", "bridges_env_create", "bridges_env_destroy",
    "bridges_env_reset", "bridges_env_step", "bridges_env_select_random", "bridges_env_lockstep_random", "bridges_env_refresh",
    "bridges_env_candidate_stability",
    "bridges_gate_create", "bridges_gate_destroy", "bridges_env_set_gate",
    "bridges_env_timing_begin", "bridges_env_timing_end",
    "bridges_place", "bridges_create_block", "bridges_pose_block", "bridges_face_frames", "bridges_contains_points", "bridges_raster", "bridges_raster_sized", "bridges_render_blocks", "bridges_action_features", "bridges_bits_or", "bridges_bits_to_f32", "bridges_bits_linear", "bridges_bits_dot", "bridges_bits_accumulate", "bridges_head_sigmoid_dot", "bridges_linear_backward_log", "bridges_mlp_mid_rows", "bridges_mlp_mid_supported", "bridges_mlp_mid_forward", "bridges_mlp_mid_backward", "bridges_eps_greedy_select", "bridges_valid_rows", "bridges_env_groups", "bridges_record_state", "bridges_record_result", "bridges_replay_unpack", "bridges_sigmoid_dot", "bridges_stability", "bridges_stability_penalty",
    "bridges_shapes_upload", "bridges_shapes_free", "bridges_soft_update", "bridges_td_target", "bridges_bias_relu", "bridges_bias_relu_pool2",
    "bridges_conv3x3_relu_o16", "bridges_conv3x3_relu_o16_ex", "bridges_conv3x3", "bridges_conv3x3_wgrad_scratch", "bridges_conv3x3_wgrad",
    "bridges_maxpool2", "bridges_maxpool2_relu_backward", "bridges_bias_grad", "bridges_adam_multi", "bridges_reduce_jobs", "bridges_upconv2x2_backward_scratch", "bridges_upconv2x2_backward", "bridges_conv1x1_o1_forward", "bridges_conv1x1_o1_backward", "bridges_upconv2x2", "bridges_linear_forward", "bridges_linear_backward", "bridges_mlp_input", "bridges_mlp_input_batches", "bridges_successor_loss", "bridges_adam_step", "bridges_linear_backward_adam",
)


def check(rc, what=""):
    if rc != 0:
        msg = lib().bridges_last_error().decode(errors="replace")
        raise BridgesHipError(f"{what} failed (rc={rc}): {msg}")


def require_gpu():
    """Fail loudly when the HIP path cannot run."""
    import torch
    L = lib()
    if not torch.cuda.is_available() or L.bridges_device_count() <= 0:
        raise BridgesHipError("no MI355X / HIP device visible: the assembly_gym hot path runs only on the "
                              "HIP kernels of libbridges_hip.so (no CPU fallback)")
    return L


def current_stream():
    """torch's current stream of the current device as the C ABI's ``void* stream`` (the raw getter: a tenth of the cost of
    building a torch.cuda.Stream object per operator call)."""
    import torch
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))

"""GPU parity: the HIP lock-step environment against the CPU oracle, env by env, step by step.
Bit-exact: candidate lists, poses/vertices (float64), in-bounds + overlap masks, bit rasters, f32 rasters,
stability booleans (both variants), rewards, termination, selected actions.  1e-5: linear rewards."""
import numpy as np
import pytest
import torch

from oracle.env import OracleGym, OracleLockstep, bridge_setup, horizontal_bridge_setup, policy_draw

pytestmark = pytest.mark.gpu


def bits_to_bool(bits_row):
    """[64] int64 (row masks, bit q = column q) -> bool[64,64]."""
    u = np.asarray(bits_row).astype(np.uint64)
    return ((u[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)


def canvas_equals(bits_row, image):
    """The 64x64 canvas of ``bits_row`` holds the S x S ``image`` in its top-left corner and nothing else."""
    canvas = bits_to_bool(bits_row)
    h, w = image.shape
    ref = np.zeros((64, 64), dtype=bool)
    ref[:h, :w] = image
    return np.array_equal(canvas, ref)


class _ShapeSpec:
    """A shape with the reference's optional face restrictions (gym_env.py:82-88 hard_tower_setup)."""

    def __init__(self, geometry, target_faces_2d=None):
        self.geometry = geometry
        if target_faces_2d is not None:
            self.target_faces_2d = target_faces_2d


def make_pair(setup_kwargs, setup_fn, E, max_steps, seed, shapes_names, **kw):
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGym
    setup = setup_fn(**setup_kwargs)
    geoms = []
    for n, s in zip(shapes_names, setup["shapes"]):
        g = load_urdf(f"shapes/{n}.urdf")
        geoms.append(_ShapeSpec(g, list(s.target_faces_2d)) if s._target else g)
    vec = VecAssemblyGym(E, geoms, setup["obstacles"], setup["targets"], max_steps=max_steps, seed=seed, **kw)
    oracles = [OracleLockstep(OracleGym(**setup, max_steps=max_steps, **kw)) for _ in range(E)]
    return vec, oracles


def compare_candidates(vec, oracles, check_f32=True):
    off = vec.cand_offset.cpu().numpy()
    n_cand = vec.n_cand.cpu().numpy()
    desc = vec.cand_desc.cpu().numpy()
    ox = vec.cand_ox.cpu().numpy()
    pose = vec.cand_pose.cpu().numpy()
    verts = vec.cand_verts.cpu().numpy()
    mask = vec.cand_mask.cpu().numpy()
    lin = vec.cand_lin.cpu().numpy()
    bits = vec.cand_bits.cpu().numpy()
    sbits = vec.state_bits.cpu().numpy()
    nvalid = vec.n_valid.cpu().numpy()
    for e, o in enumerate(oracles):
        c = o.cand
        A = len(c["actions"])
        assert n_cand[e] == A, (e, n_cand[e], A)
        sl = slice(off[e], off[e] + A)
        acts = np.array([a[:4] for a in c["actions"]], dtype=np.int32).reshape(A, 4)
        assert np.array_equal(desc[sl], acts)
        assert np.array_equal(ox[sl], np.array([a[4] for a in c["actions"]]))
        for i, b in enumerate(c["blocks"]):
            assert tuple(pose[off[e] + i]) == (b.pos[0], b.pos[1], b.cs[0], b.cs[1]), (e, i)
            nv = len(b.verts)
            assert np.array_equal(verts[off[e] + i, :nv], np.array(b.verts)), (e, i)
            assert canvas_equals(bits[off[e] + i], c["rasters"][i]), (e, i)
        assert np.array_equal(mask[sl].astype(bool), c["mask"]), e
        assert nvalid[e] == int(c["mask"].sum())
        np.testing.assert_allclose(lin[sl], c["lin_reward"], rtol=1e-5, atol=1e-6)
        assert canvas_equals(sbits[e], c["state"]), e
    if check_f32 and vec.cand_raster is not None:
        total = int(off[-1])
        img = vec.cand_raster[:total].cpu().numpy()
        ref = np.stack([bits_to_bool(b) for b in bits[:total]]).astype(np.float32) if total else img
        assert np.array_equal(img, ref)
        simg = vec.state_raster.cpu().numpy()
        assert np.array_equal(simg, np.stack([bits_to_bool(b) for b in sbits]).astype(np.float32))


def run_lockstep_parity(vec, oracles, seed, n_lock, counters=None):
    E = len(oracles)
    counters = [0] * E if counters is None else counters
    compare_candidates(vec, oracles)
    n_real = 0
    for it in range(n_lock):
        vec.select_random()
        sel = vec.sel_index.cpu().numpy()
        outs = []
        for e, o in enumerate(oracles):
            picked = {}

            def pick(nv, e=e):
                r = policy_draw(seed, e, counters[e]) % nv
                counters[e] += 1
                picked["rank"] = r
                return r
            outs.append(o.lockstep(pick))
        vec.step()
        flags = {k: v.cpu().numpy() for k, v in vec.flags().items()}
        reward = vec.reward.cpu().numpy()
        lin = vec.lin_reward.cpu().numpy()
        nre = vec.n_reached.cpu().numpy()
        for e, out in enumerate(outs):
            assert bool(flags["valid_step"][e]) == out["valid_step"], (it, e)
            assert bool(flags["no_actions"][e]) == out["no_actions"], (it, e)
            assert not flags["lp_error"][e]
            if not out["valid_step"]:
                continue
            n_real += 1
            assert sel[e] == out["action_index"], (it, e, sel[e], out["action_index"])
            assert bool(flags["stable_frozen"][e]) == out["stable_frozen"], (it, e)
            assert bool(flags["stable_unfrozen"][e]) == out["stable_unfrozen"], (it, e)
            assert bool(flags["terminated"][e]) == out["terminated"], (it, e)
            assert bool(flags["truncated"][e]) == out["truncated"], (it, e)
            assert bool(flags["done"][e]) == out["done"], (it, e)
            assert reward[e] == out["reward"], (it, e)
            assert nre[e] == out["targets_reached"]
            np.testing.assert_allclose(lin[e], out["lin_reward"], rtol=1e-5, atol=1e-7)
        compare_candidates(vec, oracles, check_f32=(it % 8 == 0))
    return n_real


@pytest.mark.parametrize("tower_height,max_steps", [(2, 10), (4, 15)])
def test_tower_lockstep_parity(tower_height, max_steps):
    E, seed = 48, 3
    vec, oracles = make_pair(dict(num_stories=tower_height), bridge_setup, E, max_steps, seed, ["trapezoid"])
    n = run_lockstep_parity(vec, oracles, seed, n_lock=24)
    assert n > E * 12
    st = vec.read_stats()
    assert st["lp_errors"] == 0 and st["if_overflow"] == 0
    assert st["env_steps"] == n


@pytest.mark.parametrize("size", [32, 17])
def test_smaller_image_sizes_parity(size):
    """--image_size S x S with S < 64 (successor_dqn.py:585): every image is the top-left S x S corner of the 64x64
    canvas the kernels write, bit-exact with the numpy oracle rendered at that size (rasters, overlap masks, linear
    rewards against the S x S reward map), and the rest of the canvas stays zero."""
    E, seed = 24, 5
    vec, oracles = make_pair(dict(num_stories=3), bridge_setup, E, 12, seed, ["trapezoid"], img_size=(size, size))
    g = oracles[0].gym
    assert canvas_equals(vec.obstacle_bits.cpu().numpy(), g.obstacle_raster)
    assert tuple(vec.reward_features.shape) == (1, size, size) and tuple(vec.obstacle_raster.shape) == (1, size, size)
    np.testing.assert_allclose(vec.reward_features[0].cpu().numpy(), g.reward_map, rtol=1e-5, atol=1e-7)
    rm = vec.reward_map.cpu().numpy()
    assert not rm[size:].any() and not rm[:, size:].any()
    n = run_lockstep_parity(vec, oracles, seed, n_lock=14)
    assert n > E * 6
    total = vec.total_candidates()
    img = vec.crop(vec.cand_raster[:total])
    assert tuple(img.shape[1:]) == (size, size)
    assert float(vec.cand_raster[:total].sum()) == float(img.sum())          # nothing outside the corner
    with pytest.raises(NotImplementedError):
        make_pair(dict(num_stories=2), bridge_setup, 2, 5, 0, ["trapezoid"], img_size=(128, 128))
    with pytest.raises(NotImplementedError):
        make_pair(dict(num_stories=2), bridge_setup, 2, 5, 0, ["trapezoid"], img_size=(64, 32))


def test_sparse_raster_update_gives_the_same_images():
    """sparse_raster_update stores only the row groups of a slot that hold pixels now or held pixels before; the f32
    images must be what the full rewrite produces, lock-step after lock-step (slots are re-used by other candidates)."""
    from bridges_hip.vec_env import VecAssemblyGym
    E, seed = 96, 5
    full, oracles = make_pair(dict(num_stories=4), bridge_setup, E, 15, seed, ["trapezoid"])
    sparse = VecAssemblyGym(E, full.shapes, full.obstacles, full.targets, max_steps=15, seed=seed, sparse_raster_update=True)
    assert sparse.cand_raster_nz is not None and full.cand_raster_nz is None
    n = run_lockstep_parity(sparse, oracles, seed, n_lock=9)          # f32 images against the oracle at lock-steps 0 and 8
    assert n > E * 4
    for _ in range(9):                                                # same seed, same policy stream: same trajectory
        full.select_random()
        full.step()
    for it in range(40):
        total = int(full.cand_offset[E])
        assert total == int(sparse.cand_offset[E])
        assert torch.equal(full.cand_raster[:total], sparse.cand_raster[:total]), it
        assert torch.equal(full.state_raster, sparse.state_raster), it
        assert torch.equal(full.cand_mask[:total], sparse.cand_mask[:total])
        for env in (full, sparse):
            env.select_random()
            env.step()


def test_hexagon_bridge_lockstep_parity():
    E, seed = 32, 11
    from oracle.shapes import get_shape
    vec, oracles = make_pair(dict(num_obstacles=3, trapezoid=False, hexagon=True), horizontal_bridge_setup, E, 15, seed,
                             ["hexagon"], mu=0.8)
    run_lockstep_parity(vec, oracles, seed, n_lock=16)


def test_several_targets_in_one_bounding_box_parity():
    """Three targets that one block's bounding box can hold at once + a far one: the reference's remove-while-iterating
    bookkeeping (gym_env.py:163-169) skips the target after a reached one; k_step and the oracle agree step by step."""
    from oracle.shapes import get_shape

    def setup():
        return dict(shapes=[get_shape("trapezoid")], obstacles=[], targets=[(-1.0, 0, 0.4), (-0.9, 0, 0.4), (-0.8, 0, 0.4), (5.0, 0, 5.0)])
    E, seed = 32, 8
    vec, oracles = make_pair({}, setup, E, 10, seed, ["trapezoid"])
    run_lockstep_parity(vec, oracles, seed, n_lock=14)
    assert int(vec.n_reached.max()) >= 1


def test_task_features_match_oracle():
    vec, oracles = make_pair(dict(num_stories=4), bridge_setup, 4, 15, 0, ["trapezoid"])
    g = oracles[0].gym
    assert np.array_equal(bits_to_bool(vec.obstacle_bits.cpu().numpy()), g.obstacle_raster)
    np.testing.assert_allclose(vec.reward_map.cpu().numpy(), g.reward_map, rtol=1e-5, atol=1e-7)


def test_candidate_stability_mask_matches_is_action_stable_rbe():
    """stability of every valid candidate placement, batched."""
    E, seed = 24, 7
    vec, oracles = make_pair(dict(num_stories=2), bridge_setup, E, 10, seed, ["trapezoid"])
    counters = [0] * E
    checked = 0
    for it in range(5):
        rows, stable = vec.candidate_stability()
        rows, stable = rows.cpu().numpy(), stable.cpu().numpy()
        off = vec.cand_offset.cpu().numpy()
        k = 0
        for e, o in enumerate(oracles):
            valid = np.flatnonzero(o.cand["mask"])
            for a in valid:
                assert rows[k] == off[e] + a
                assert bool(stable[k]) == o.gym.is_action_stable(o.cand["actions"][a]), (it, e, a)
                k += 1
                checked += 1
        assert k == len(rows)
        vec.select_random()
        for e, o in enumerate(oracles):
            def pick(nv, e=e):
                r = policy_draw(seed, e, counters[e]) % nv
                counters[e] += 1
                return r
            o.lockstep(pick)
        vec.step()
    assert checked > 500


def test_mixed_shapes_high_friction_parity():
    """trapezoid + hexagon in one task (10 candidate groups), mu = 2.0, a 4-obstacle bridge."""
    E, seed = 24, 21
    vec, oracles = make_pair(dict(num_obstacles=4, trapezoid=True, hexagon=True), horizontal_bridge_setup, E, 12, seed,
                             ["trapezoid", "hexagon"], mu=2.0)
    n = run_lockstep_parity(vec, oracles, seed, n_lock=14)
    assert n > E * 8


def test_restricted_target_faces_two_targets_no_step_limit_parity():
    """hard_tower_setup: a cube that may only be attached by face 2, two targets, an obstacle, max_steps=None
    (K = 16 block slots, 'truncated' never set)."""
    from oracle.env import hard_tower_setup
    E, seed = 16, 33
    vec, oracles = make_pair({}, hard_tower_setup, E, None, seed, ["trapezoid", "cube1"])
    assert len(vec.groups) == 5                      # 4 trapezoid faces + the cube's single target face
    run_lockstep_parity(vec, oracles, seed, n_lock=14)
    assert not bool(vec.flags()["truncated"].any())


@pytest.mark.parametrize("task,envs,locksteps", [("tower4", 96, 14), ("hexbridge", 48, 12), ("mixed", 40, 10)])
def test_fused_candidate_stability_against_the_c_oracle(task, envs, locksteps):
    """>= 5000 candidate placements per task decided by bridges_env_candidate_stability and by the C oracle's
    is_action_stable_rbe: 0 mismatches, 0 solver errors."""
    import subprocess, sys, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress", "stress_candidate_stability.py"), "--envs", str(envs),
                          "--locksteps", str(locksteps), "--task", task], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"RESULT .*: (\d+) candidate decisions compared \((\d+) unstable.*, 0 mismatches, 0 lp errors", out.stdout)
    assert m, out.stdout[-500:]
    assert int(m.group(1)) >= 5000 and int(m.group(2)) > 100


def test_fused_candidate_stability_equals_the_unfused_operator_path():
    """Second, independent GPU path at a size the CPU oracles do not reach: gathered copies of every assembly + the
    stand-alone bridges_stability operator (interfaces re-detected from scratch) give the same booleans."""
    from gpu_helpers import candidate_stability_unfused
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGym
    setup = bridge_setup(num_stories=4)
    vec = VecAssemblyGym(1024, [load_urdf("shapes/trapezoid.urdf")], setup["obstacles"], setup["targets"], max_steps=15, seed=3,
                         f32_rasters=False)
    n = 0
    for it in range(8):
        rows, stable = vec.candidate_stability()
        rows2, stable2, err2 = candidate_stability_unfused(vec)
        assert torch.equal(rows, rows2) and not bool(err2.any())
        assert torch.equal(stable, stable2), it
        assert int((vec.cand_stable[rows] == 2).sum()) == 0
        n += rows.numel()
        vec.select_random()
        vec.step()
    assert n > 50000


@pytest.mark.parametrize("density", [0.1, 50.0])
def test_density_scaled_tolerances_parity(density):
    """AssemblyEnv(density=...) is a public input (assembly_env.py:164): every absolute LP tolerance scales with it,
    so the lock-step booleans at density d agree with the oracle at density d (and, by the CPU test, with density 1)."""
    E, seed = 24, 41
    vec, oracles = make_pair(dict(num_stories=4), bridge_setup, E, 15, seed, ["trapezoid"], density=density)
    n = run_lockstep_parity(vec, oracles, seed, n_lock=12)
    assert n > E * 8
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress", "stress_candidate_stability.py"), "--envs", "32",
                          "--locksteps", "6", "--task", "tower4", "--density", str(density)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "0 mismatches, 0 lp errors" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_state_capacity_is_a_forced_truncation_and_neighbours_stay_intact():
    """max_steps=None -> K = 16 block slots.  A state that fills them is truncated and auto-reset (slot nb == K is
    never written); every env's block list stays what was placed into it, bit for bit."""
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGym
    # ground-only placements of a small cube, 0.8 apart: they never collide and never become unstable
    cube = load_urdf("shapes/cube06.urdf")
    E = 3
    vec = VecAssemblyGym(E, [cube], [], [(50.0, 0.0, 50.0)], max_steps=None, seed=1, xlim=(-3.0, 17.0), ylim=(0.0, 20.0),
                         x_discr_ground=np.linspace(-2.4, 16.0, 24), f32_rasters=False,
                         bounds=((-30.0, -3.0, -1.0), (30.0, 3.0, 9.0)))
    assert vec.K == 16
    mirror = [[] for _ in range(E)]
    truncations = 0
    for it in range(40):
        off = vec.cand_offset.cpu().numpy(); mask = vec.cand_mask.cpu().numpy(); desc = vec.cand_desc.cpu().numpy()
        cpose = vec.cand_pose.cpu().numpy()
        sel = []
        for e in range(E):
            rows = [a for a in range(off[e], off[e + 1]) if mask[a] and desc[a, 0] < 0]      # ground candidates only
            assert rows, "no free ground slot left"
            pick = rows[(it + e) % len(rows)]                  # envs take different slots: their block lists differ
            sel.append(pick - off[e])
            mirror[e].append(cpose[pick].copy())
        vec.step(torch.tensor(sel, dtype=torch.int32))
        fl = {k: v.cpu().numpy() for k, v in vec.flags().items()}
        nb = vec.n_blocks.cpu().numpy()
        pose = vec.blk_pose.cpu().numpy()
        assert not fl["lp_error"].any() and fl["stable_frozen"].all()
        for e in range(E):
            if len(mirror[e]) >= 16:
                assert fl["truncated"][e] and fl["done"][e] and nb[e] == 0
                mirror[e] = []
                truncations += 1
            else:
                assert not fl["truncated"][e] and not fl["done"][e] and nb[e] == len(mirror[e])
                assert np.array_equal(pose[e, :nb[e]], np.array(mirror[e])), (it, e)
    assert truncations == E * 2


@pytest.mark.parametrize("task", ["tower4", "mixed"])
def test_stress_parity_against_the_c_oracle(task):
    """Thousands of env-steps against the plain-C oracle (fast enough to follow): every selected action, both
    stability booleans, rewards, termination, candidate / valid counts."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress", "stress_parity.py"), "--envs", "256", "--locksteps", "40",
                          "--task", task, "--seed", "29"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 mismatches" in out.stdout


def test_candidate_capacity_overflow_is_flagged_and_counted():
    """a_max is a public constructor argument.  A state with more raw candidates than a_max has its candidate set cut to
    a_max -- data loss the reference cannot have (generate_actions enumerates everything, actions.py:7-52) -- so the cut
    must be visible: bit 3 of the env's lp_error flag, flags()['cand_overflow'], and the cand_overflow statistic.  The
    default capacity (the task's bound) never overflows."""
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGym
    setup = bridge_setup(num_stories=2)
    E = 32
    small = VecAssemblyGym(E, [load_urdf("shapes/trapezoid.urdf")], setup["obstacles"], setup["targets"], max_steps=10, seed=3,
                           a_max=44, f32_rasters=False)
    full = VecAssemblyGym(E, [load_urdf("shapes/trapezoid.urdf")], setup["obstacles"], setup["targets"], max_steps=10, seed=3,
                          f32_rasters=False)
    assert small.read_stats()["cand_overflow"] == 0 and not small.flags()["cand_overflow"].any()     # 40 ground candidates fit
    seen = 0
    for _ in range(6):
        for env in (small, full):
            env.select_random()
            env.step()
        nb, nc = small.n_blocks.cpu().numpy(), small.n_cand.cpu().numpy()
        ovf = small.flags()["cand_overflow"].cpu().numpy()
        raw = full.n_cand.cpu().numpy()
        # the first lock-step is the same on both (the 40 ground candidates); afterwards the cut sets lead elsewhere
        if seen == 0:
            assert np.array_equal(nb, full.n_blocks.cpu().numpy())
            assert np.array_equal(ovf, raw > 44) and np.array_equal(nc, np.minimum(raw, 44))
        assert np.array_equal(ovf, nb >= 1) and (nc[ovf] == 44).all() and (nc[~ovf] == 40).all()     # one block: 4 * (10 + 3) = 52 > 44
        assert not (small.flags()["lp_error"].cpu().numpy()).any()                                      # not mixed into the solver's error bits
        seen += int(ovf.sum())
        assert small.read_stats()["cand_overflow"] == seen
    assert seen > 0
    assert full.read_stats()["cand_overflow"] == 0 and not full.flags()["cand_overflow"].any()
    # the replay path (records -> states) goes through the same clamp
    from robotoddler.training import records as R
    rec = torch.zeros((E, R.RECORD_WIDTH), dtype=torch.float64, device=small.device)
    rec[:, R.O_NB] = 1                                              # s holds one block, the action adds a second: 4 * (10 + 6) = 64
    rec[:, R.O_POSE + 2] = 1.0
    rec[:, R.O_POSE] = -1.0
    rec[:, R.O_OCC] = 8
    rec[:, R.O_APOSE:R.O_APOSE + 4] = torch.tensor([1.5, 0.3595713675022125, 1.0, 0.0], dtype=torch.float64)
    rec[:, R.O_ATB], rec[:, R.O_AFACE] = -1, 3
    before = small.read_stats()["cand_overflow"]
    small.load_records(rec)
    assert small.read_stats()["cand_overflow"] == before + E and (small.n_cand.cpu().numpy() == 44).all()
    full.load_records(rec)
    assert full.read_stats()["cand_overflow"] == 0 and (full.n_cand.cpu().numpy() == 64).all()

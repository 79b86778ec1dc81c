"""GPU: the assembly_gym drop-in API (single environment, stand-alone HIP operators) replays every recorded
output the reference holds -- the same golden vectors the oracle is pinned with."""
import json
import os

import numpy as np
import torch
import pytest

pytestmark = pytest.mark.gpu

FREEZE_DEFAULT = dict(trapezoid_bridge=True, hexagon_bridge_3=True, hexagon_bridge_5=True,
                      horizontal_bridge=True, levitating_block=False)
KNOWN_DEVIATIONS = {("trapezoid_bridge", True, 0.8, 8), ("trapezoid_bridge", False, 0.8, 8)}


def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_stability_table_through_the_api(golden_dir):
    from assembly_gym.envs.assembly_env import AssemblyEnv, Shape
    from assembly_gym.envs.gym_env import Action, AssemblyGym, sparse_reward
    from assembly_gym.utils.stability import is_stable_rbe
    table = _load(golden_dir, "stability_table.json")
    structs = _load(golden_dir, "structures.json")
    cache, diffs = {}, set()
    for row in table:
        name, kw, mu = row["structure"], row["kwargs"], row["mu"]
        key = (name, json.dumps(kw, sort_keys=True), mu)
        if key not in cache:
            st = structs[name]
            acts = st["actions"][:kw.get("num_blocks", 3)] if name == "tower" else st["actions"]
            fl = kw.get("freeze_last", FREEZE_DEFAULT.get(name, False))
            env = AssemblyGym(shapes=[Shape(urdf_file=f"shapes/{s}.urdf") for s in st["shapes"]], targets=[], obstacles=[],
                              reward_fct=sparse_reward, restrict_2d=True,
                              assembly_env=AssemblyEnv(render=False, mu=mu, density=1.0, stability=None))
            out = []
            for a in acts:
                frozen = fl if a[6] == "FL" else a[6]
                # the table predates the forced freeze of gym_env.py:238: replay its freeze semantics by hand
                blk = env.create_block(Action(*a[:6]))
                if env.assembly_env.blocks and env.assembly_env.blocks[-1].is_static:
                    env.assembly_env.unfreeze_block(len(env.assembly_env.blocks) - 1)
                env.assembly_env.blocks.append(blk)
                if frozen:
                    env.assembly_env.freeze_block(len(env.assembly_env.blocks) - 1)
                out.append(is_stable_rbe(env.assembly_env)[0])
            cache[key] = out
        if cache[key][row["step"]] != row["rbe"]:
            diffs.add((name, kw.get("freeze_last"), mu, row["step"]))
    assert diffs == KNOWN_DEVIATIONS


def test_structure_harness_reproduces_the_recorded_table(golden_dir, tmp_path):
    """assembly_gym.utils.test_suite (the reference's regression harness over assembly_gym.utils.structures): its
    structure.json files carry the recorded RBE column and the hand labels of the stored table."""
    from assembly_gym.utils import test_suite
    table = _load(golden_dir, "stability_table.json")
    got = {}
    for mu in (0.8, 2.0):
        for path in test_suite.run(str(tmp_path / f"mu{mu}"), mu=mu):
            data = json.load(open(path))
            assert sorted(m["name"] for m in data["methods"].values()) == ["rbe", "rbe_penalty"]
            mid = next(k for k, m in data["methods"].items() if m["name"] == "rbe")
            for t in data["tests"].values():
                key = (data["structure"]["name"], json.dumps(data["structure"]["kwargs"], sort_keys=True), mu, t["step"])
                got[key] = (t[mid]["is_stable"], t["is_stable"])
    assert len(got) == len(table) == 96
    diffs = set()
    for row in table:
        rbe, expected = got[(row["structure"], json.dumps(row["kwargs"], sort_keys=True), row["mu"], row["step"])]
        assert expected == row["expected"], row                  # the hand labels (structures.py) as rules
        if rbe != row["rbe"]:
            diffs.add((row["structure"], row["kwargs"].get("freeze_last"), row["mu"], row["step"]))
    assert diffs == KNOWN_DEVIATIONS


def test_notebook_bridge_episode_through_gym_step(golden_dir):
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, horizontal_bridge_setup, sparse_reward
    g = _load(golden_dir, "assembly_env_notebook.json")["cell21"]
    st = _load(golden_dir, "structures.json")["notebook_cell21"]
    env = AssemblyGym(**horizontal_bridge_setup(num_obstacles=7), reward_fct=sparse_reward, restrict_2d=True,
                      assembly_env=AssemblyEnv(render=False, mu=2.0))
    env.reset()
    for a, gold in zip(st["actions"], g):
        obs, reward, terminated, truncated, info = env.step(Action(*a))
        assert (obs["stable"], len(obs["targets_reached"]), reward, terminated) == \
               (gold["stable"], gold["targets_reached"], gold["reward"], gold["terminated"])
        assert truncated is None                                  # 'Truncated: None' in the notebook print


def test_notebook_hard_tower_distances_bit_exact(golden_dir):
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, hard_tower_setup, sparse_reward
    g = _load(golden_dir, "assembly_env_notebook.json")["cell24_25"]
    st = _load(golden_dir, "structures.json")["notebook_cell24_25"]
    env = AssemblyGym(**hard_tower_setup(), reward_fct=sparse_reward, restrict_2d=True,
                      assembly_env=AssemblyEnv(render=False))
    for a, gold in zip(st["actions"], g):
        obs, reward, terminated, truncated, info = env.step(Action(*a))
        assert obs["stable"] == gold["stable"] and reward == gold["reward"] and terminated == gold["terminated"]
        assert obs["distance_to_targets"] == gold["distance_to_targets"]
        assert len(obs["targets_reached"]) == gold["targets_reached"]


def test_render_and_stabilities_freezing_match_oracle():
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, bridge_setup, sparse_reward
    from assembly_gym.utils.rendering import render_blocks_2d
    from oracle.env import OracleGym
    from oracle.env import bridge_setup as o_bridge_setup
    env = AssemblyGym(**bridge_setup(num_stories=2), reward_fct=sparse_reward, restrict_2d=True, max_steps=10,
                      assembly_env=AssemblyEnv(render=False))
    og = OracleGym(**o_bridge_setup(num_stories=2), max_steps=10)
    acts = [(-1, 0, 0, 3, -1.3333333333333335, 0.0), (0, 1, 0, 3, 0.0, 0.0), (1, 2, 0, 0, 0.0, 0.0)]
    for a in acts:
        obs, r, term, trunc, _ = env.step(Action(*a))
        stable, r2, term2, trunc2 = og.step(a)
        assert (obs["stable"], r, bool(term), bool(trunc)) == (stable, r2, term2, trunc2)
        assert env.stabilities_freezing() == og.stabilities_freezing()
        img = render_blocks_2d(obs["blocks"], xlim=(-3, 7), ylim=(0, 10), img_size=(64, 64))
        assert np.array_equal(img, og.state_raster())
        for size in (48, 16):                                   # render_blocks_2d's img_size argument, S <= 64
            from oracle import raster as o_raster
            small = render_blocks_2d(obs["blocks"], xlim=(-3, 7), ylim=(0, 10), img_size=(size, size))
            assert small.shape == (size, size)
            assert np.array_equal(small, o_raster.render_blocks_2d(og.blocks, (-3, 7), (0, 10), (size, size)))
        for b, ob in zip(obs["blocks"], og.blocks):
            assert np.array_equal(b.verts_2d, np.array(ob.verts))


def test_feature_functions_at_a_smaller_image_size():
    """get_state_features / get_task_features / get_action_features with --image_size 32x32 (successor_dqn.py:47-94,
    585) against the numpy oracle rendered at that size."""
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, bridge_setup, sparse_reward
    from oracle import raster as o_raster
    from oracle.env import OracleGym
    from oracle.env import bridge_setup as o_bridge_setup
    from robotoddler.training.successor_dqn import get_action_features, get_state_features, get_task_features
    size, lim = (32, 32), dict(xlim=(-3, 7), ylim=(0, 10))
    env = AssemblyGym(**bridge_setup(num_stories=2), reward_fct=sparse_reward, restrict_2d=True, max_steps=10,
                      assembly_env=AssemblyEnv(render=False))
    og = OracleGym(**o_bridge_setup(num_stories=2), max_steps=10, img_size=size)
    obs, _ = env.reset()
    obs, *_ = env.step(Action(-1, 0, 0, 3, -1.3333333333333335, 0.0))
    og.step((-1, 0, 0, 3, -1.3333333333333335, 0.0))
    image, binary = get_state_features(obs, img_size=size, **lim)
    assert tuple(image.shape) == (1, 32, 32) and np.array_equal(image[0].cpu().numpy().astype(bool), og.state_raster())
    reward, obstacle = get_task_features(obs, img_size=size, **lim)
    np.testing.assert_allclose(reward[0].cpu().numpy(), og.reward_map, rtol=1e-5, atol=1e-7)
    assert np.array_equal(obstacle[0].cpu().numpy().astype(bool), og.obstacle_raster)
    cand = og.candidates()
    actions = [Action(*a) for a in cand["actions"]]
    feats = get_action_features(env, actions, img_size=size, **lim)
    assert tuple(feats.shape) == (len(actions), 1, 32, 32)
    assert np.array_equal(feats[:, 0].cpu().numpy().astype(bool), cand["rasters"])


def test_contains_2d_points_and_scripted_rollout():
    import torch
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, horizontal_bridge_setup, sparse_reward
    from oracle.env import OracleGym
    from oracle.env import horizontal_bridge_setup as o_setup
    from oracle.raster import contains_2d, pixel_grid
    from robotoddler.training.successor_dqn import rollout_episode_scripted
    env = AssemblyGym(**horizontal_bridge_setup(num_obstacles=2), reward_fct=sparse_reward, restrict_2d=True, max_steps=10,
                      assembly_env=AssemblyEnv(render=False))
    acts = [Action(-1, 0, 0, 2, -0.9, 0.0), Action(0, 0, 0, 2, 0.0, 0.0)]
    trans, _ = rollout_episode_scripted(env, acts, lambda: horizontal_bridge_setup(num_obstacles=2), np.linspace(-2, 0, 10),
                                        device=torch.device("cuda"))
    assert len(trans) == 2 and trans[0].td_error == 0 and trans[0].lin_reward.shape == (1,)
    og = OracleGym(**o_setup(num_obstacles=2), max_steps=10)
    for a in acts:
        og.step((a.target_block, a.target_face, a.shape, a.face, a.offset_x, a.offset_y))
    X, Y = pixel_grid((-3, 7), (0, 10))
    XX, YY = np.meshgrid(X, Y)
    pts = np.stack([XX.ravel(), YY.ravel()], axis=1)
    for b, ob in zip(env.assembly_env.blocks, og.blocks):
        assert np.array_equal(b.contains_2d(pts).reshape(64, 64), contains_2d(ob, X, Y))
    # lin_reward of the scripted rollout = sum(action raster * reward map)
    np.testing.assert_allclose(trans[0].lin_reward.item(), og.reward_map[contains_2d(og.blocks[0], X, Y)].sum(), rtol=1e-5)


def test_rbe_penalty_variant_against_the_oracle(golden_dir):
    """is_stable_rbe_penalty (stability.py:75-88 of the reference; maximum_tension geometry.py:132-143) through the
    drop-in API on every build step of the known-answer structures, against the oracle's restatement of the same
    predicate (HiGHS).  The reference holds NO recorded output of this variant: parity unpinned, oracle only.
    Properties on top: RBE-stable implies penalty-stable; a penalty-stable verdict comes with forces whose largest net
    tension is <= tol; AssemblyEnv(stability='rbe_penalty') serves it; the CRA variants say that they are not built."""
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.utils import structures
    from assembly_gym.utils.stability import is_stable_cra, is_stable_rbe, is_stable_rbe_penalty
    from oracle import rbe as ORB
    from oracle.geometry import Block
    from oracle.shapes import get_shape
    n = differ_from_rbe = 0
    for name, kw in (("trapezoid_bridge", dict(freeze_last=False)), ("hexagon", dict()), ("hexagon_bridge_5", dict(freeze_last=True)),
                     ("horizontal_bridge", dict(freeze_last=False)), ("tower", dict(num_blocks=6)), ("levitating_block", dict())):
        for mu in (0.4, 0.8, 2.0):
            env, actions = structures.build(name, mu=mu, **kw)
            for step, action, expected, (rbe, _) in structures.replay(env, actions, method=is_stable_rbe):
                asm = env.assembly_env
                pen, extra = is_stable_rbe_penalty(asm, tol=1e-3)
                loose, _ = is_stable_rbe_penalty(asm, tol=50.0)
                blocks = [Block(get_shape(b.shape.name), (b.pose[0], b.pose[1]), (b.pose[2], b.pose[3])) for b in asm.blocks]
                fixed = {i for i, b in enumerate(asm.blocks) if b.is_static}
                want, info = ORB.is_stable_rbe_penalty(blocks, fixed, mu=mu, tol=1e-3, return_info=True)
                assert pen == want, (name, kw, mu, step, info)
                assert loose == ORB.is_stable_rbe_penalty(blocks, fixed, mu=mu, tol=50.0), (name, kw, mu, step)
                if rbe:
                    assert pen
                if extra is None:                          # an assembly without contacts: (no free block, None), stability.py:77-81
                    assert len(ORB.find_interfaces(blocks)) == 0
                elif pen:
                    assert extra["max_tension"] is not None and extra["max_tension"] <= 1e-3 + 1e-9
                else:
                    assert extra["max_tension"] is None and (info["min_tension"] is None or info["min_tension"] > 1e-3)
                differ_from_rbe += int(bool(loose) != bool(rbe))
                n += 1
    assert n > 70 and differ_from_rbe > 0              # a generous tension allowance does change verdicts
    env = AssemblyEnv(render=False, stability="rbe_penalty")
    assert env.stability_fct is is_stable_rbe_penalty
    with pytest.raises(NotImplementedError):
        is_stable_cra(env)
    with pytest.raises(NotImplementedError):
        AssemblyEnv(render=False, stability="cra")


def test_action_features_operator_equals_the_reference_composition():
    """bridges_action_features (rasters + in-bounds / no-overlap mask + linear reward in one call) against what
    the reference composes from get_action_features, filter_actions and sum(action * reward) (successor_dqn.py:84-94,
    397-401; actions.py:71-82), through the drop-in functions and against the numpy oracle, at 64x64 and 32x32."""
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, bridge_setup, sparse_reward
    from assembly_gym.utils.rendering import render_blocks_2d_bits
    from bridges_hip import ops
    from oracle.env import OracleGym
    from oracle.env import bridge_setup as o_bridge_setup
    from robotoddler.training.successor_dqn import get_action_features, get_state_features, get_task_features
    from robotoddler.utils.actions import filter_actions
    for size in ((64, 64), (32, 32)):
        lim = dict(xlim=(-3, 7), ylim=(0, 10))
        env = AssemblyGym(**bridge_setup(num_stories=2), reward_fct=sparse_reward, restrict_2d=True, max_steps=10,
                          assembly_env=AssemblyEnv(render=False))
        og = OracleGym(**o_bridge_setup(num_stories=2), max_steps=10, img_size=size)
        obs, _ = env.reset()
        for a in ((-1, 0, 0, 3, -1.3333333333333335, 0.0), (0, 1, 0, 3, 0.0, 0.0)):
            obs, *_ = env.step(Action(*a))
            og.step(a)
        cand = og.candidates()
        actions = [Action(*a) for a in cand["actions"]]
        blocks = env.create_blocks(actions)
        state_bits = render_blocks_2d_bits(obs["blocks"], lim["xlim"], lim["ylim"], size).reshape(64)
        obst_bits = render_blocks_2d_bits(obs["obstacle_blocks"], lim["xlim"], lim["ylim"], size).reshape(64)
        reward, obstacle = get_task_features(obs, img_size=size, **lim)
        canvas = torch.zeros((64, 64), dtype=torch.float32, device=reward.device)
        canvas[:size[0], :size[1]] = reward[0]
        bits, img, mask, lin = ops.action_features(blocks, lim["xlim"], lim["ylim"], state_bits=state_bits, obstacle_bits=obst_bits,
                                                   reward_map=canvas, img_size=size, want_f32=True)
        S = size[0]
        # rasters: the oracle's and get_action_features'
        assert np.array_equal(img[:, :S, :S].cpu().numpy().astype(bool), cand["rasters"])
        feats = get_action_features(env, actions, img_size=size, **lim)
        assert torch.equal(img[:, :S, :S], feats[:, 0])
        assert torch.equal(ops.bits_to_f32(bits), img)
        # mask: the oracle's and filter_actions'
        assert np.array_equal(mask.cpu().numpy(), cand["mask"])
        block_f, _ = get_state_features(obs, img_size=size, **lim)
        kept, _ = filter_actions(env, actions, feats, block_features=block_f, obstacle_features=obstacle, **lim)
        assert [a for a, m in zip(actions, mask.tolist()) if m] == kept
        # linear reward: sum(action raster * reward map) of the rollout, 1e-5
        want = (feats[:, 0] * reward).sum(dim=(1, 2))
        assert torch.allclose(lin, want, rtol=1e-5, atol=1e-6), float((lin - want).abs().max())


def test_raster_with_edges_through_pixel_centres():
    """The row-run rasteriser evaluates the reference's comparison  ((X - c.x) n.x) + ((Y - c.z) n.z) <= 0  through a
    binary search per face and row: the cases where it decides pixel by pixel -- edges that pass EXACTLY through sample
    points, axis-aligned and diagonal, blocks hanging over every border of the image -- against the per-pixel numpy oracle,
    at three image sizes."""
    from bridges_hip import ops
    from oracle import raster as o_raster
    from oracle.geometry import Block as OBlock
    from oracle.shapes import get_shape
    from assembly_gym.envs.assembly_env import Block, Shape
    import math
    rng = np.random.default_rng(3)
    for S in (64, 33, 17):
        xlim, ylim = (-3.0, 7.0), (0.0, 10.0)
        X, Y = o_raster.pixel_grid(xlim, ylim, (S, S))
        cases = []
        for name in ("cube06", "cube1", "trapezoid", "hexagon", "block"):
            for k in range(12):
                gxi, gyi = rng.integers(0, S), rng.integers(0, S)
                ang = [0.0, math.pi / 2, math.pi / 4, math.pi, rng.uniform(0, 2 * math.pi)][k % 5]
                # centre chosen so that a vertex or an edge midpoint of the (rotated) block lands on a sample point
                cases.append((name, float(X[gxi]) + (0.3 if name == "cube06" and k % 2 else 0.0), float(Y[gyi]), ang))
            cases.append((name, xlim[0] - 0.2, 5.0, 0.3)); cases.append((name, 6.9, ylim[1] + 0.1, 1.1)); cases.append((name, 2.0, -0.2, 0.0))
        blocks, oblocks = [], []
        for name, x, z, ang in cases:
            c, s_ = math.cos(ang), math.sin(ang)
            from assembly_gym.envs.assembly_env import Quaternion
            b = Block(Shape(urdf_file=f"shapes/{name}.urdf"), position=[x, 0.0, z], orientation=Quaternion.from_cos_sin(c, s_))
            ob = OBlock(get_shape(name), (b.pose[0], b.pose[1]), (b.pose[2], b.pose[3]))     # the pose the drop-in block holds
            assert np.array_equal(b.verts_2d, np.array(ob.verts))
            blocks.append(b); oblocks.append(ob)
        bits = ops.raster_bits(blocks, xlim, ylim, (S, S))
        img = ops.bits_to_f32(bits)[:, :S, :S].cpu().numpy().astype(bool)
        on_edge = 0
        for i, ob in enumerate(oblocks):
            want = o_raster.contains_2d(ob, X, Y)
            assert np.array_equal(img[i], want), (S, cases[i])
            for (c, _t, n) in ob.frames:                    # how many sample points sit exactly on an edge line
                d = ((X - c[0]) * n[0])[None, :] + ((Y - c[1]) * n[1])[:, None]
                on_edge += int((d == 0).sum())
        assert on_edge > 50                                  # the degenerate cases are really in there
        canvas = ops.bits_to_f32(bits).cpu().numpy()
        assert not canvas[:, S:, :].any() and not canvas[:, :, S:].any()


def test_cra_notebook_cell6_through_the_drop_in_surface(golden_dir):
    """notebooks/CRA_Assembly.ipynb cell 6 as the notebook writes it (Shape / AssemblyEnv / align_frames_2d / Block /
    add_block), on the HIP operators: the two recorded `'stable'` prints, and bridges_stability's interface count against
    cell 7's "Number of interfaces: 4" (one face pair per touching body pair in this assembly) and cell 8's three free blocks."""
    from assembly_gym.envs.assembly_env import AssemblyEnv, Block, Shape
    from assembly_gym.utils.geometry import align_frames_2d
    from bridges_hip import ops
    g = _load(golden_dir, "cra_assembly_notebook.json")
    trapezoid = Shape(urdf_file='shapes/trapezoid.urdf')
    env = AssemblyEnv(render=False)
    env.reset()
    shape = trapezoid
    position, rotation = align_frames_2d(env.get_floor_frame(), shape.get_face_frame_2d(face=3), frame1_coordinates=[0.35, 0, 0])
    block1 = Block(shape=shape, position=position, orientation=rotation.quaternion)
    info1 = env.add_block(block1)
    position, rotation = align_frames_2d(block1.get_face_frame_2d(face=1), trapezoid.get_face_frame_2d(face=2), frame1_coordinates=[0.8, 0.0, 0])
    block2 = Block(shape=trapezoid, position=position, orientation=rotation.quaternion)
    info2 = env.add_block(block2)
    p2 = block2.get_face_frame_2d(1).to_world_coordinates([-0.0, 0, 0])
    position, rotation = align_frames_2d(block2.get_face_frame_2d(face=1), trapezoid.get_face_frame_2d(face=2), frame1_coordinates=[-0.0, 0.0, 0])
    block3 = Block(shape=trapezoid, position=position, orientation=rotation.quaternion)
    info3 = env.add_block(block3)
    assert info1["stable"] is True
    assert [info2["stable"], info3["stable"]] == g["cell6_stable"] == [False, True]
    assert info3["frozen_block"] is None and info3["collision"] is False
    assert len(p2) == 3 and p2[1] == 0.0
    counts = []
    for n in (1, 2, 3):
        stable, info = ops.stability(env.blocks[:n], set(), env.mu, env.density, env.floor_half_width, env.floor_depth)
        counts.append(info["n_interfaces"])
    assert counts == [1, 2, 4] and counts[-1] == g["cell7_number_of_edges"]
    assert g["cell8_free_blocks"] == len(env.blocks)
    # the block the kernel posed from (position, rotation.quaternion) is the block the oracle places
    from oracle.geometry import create_block
    from oracle.shapes import get_shape
    st = _load(golden_dir, "structures.json")["notebook_cra_cell6"]
    shapes, oblocks = [get_shape("trapezoid")], []
    for p in st["placements"]:
        oblocks.append(create_block(shapes, oblocks, p))
    for ob, b in zip(oblocks, env.blocks):
        assert np.array_equal(np.asarray(ob.verts), b.verts_2d)


def test_cra_notebook_tutorial_box_through_the_c_abi(golden_dir):
    """notebooks/CRA_Assembly.ipynb cells 2-4 (the compas_cra tutorial: a 1 x 3 x 1 box on a fixed 4 x 2 x 1 support)
    through bridges_stability / bridges_stability_penalty: one interface, stable, and the contact compressions of the
    equilibrium found sum to the recorded 4 x 0.75 (= density x volume of the free box)."""
    from assembly_gym.envs.assembly_env import Block, Shape
    from bridges_hip import ops
    from bridges_hip.shapes import ShapeGeometry, _box_outline
    g = _load(golden_dir, "cra_assembly_notebook.json")
    st = _load(golden_dir, "structures.json")["notebook_cra_cell2"]
    blocks, fixed = [], set()
    for i, (sx, sy, sz, pos, fx) in enumerate(st["boxes"]):
        blocks.append(Block(Shape(mesh=ShapeGeometry(*_box_outline(sx, sy, sz), name=f"box{i}")), position=pos))
        if fx:
            fixed.add(i)
    stable, info = ops.stability(blocks, fixed, st["mu"], st["density"], 5.0, 10.0)
    assert stable is True and info["n_interfaces"] == g["cell2_number_of_edges"] == 1
    stable, info = ops.stability(blocks, fixed, st["mu"], st["density"], 5.0, 10.0, tension_tol=1e-3)
    assert stable is True and info["forces"].shape == (1, 2, 3)
    assert info["forces"][:, :, 0].sum() == pytest.approx(sum(g["cell4_normal_forces"]), abs=1e-5)     # 3.0 within the LP's residual budget (FEAS_TOL * density)
    assert info["forces"][:, :, 1].sum() == pytest.approx(0.0, abs=1e-5)                                   # nothing pulls


def test_candidate_block_cache_does_not_outlive_a_reset():
    """create_blocks keeps the blocks of the current state's candidates; reset(shapes=...) to OTHER shapes with no step in
    between must not serve the previous episode's blocks (the reference builds every block afresh, gym_env.py:204-216)."""
    from assembly_gym.envs.assembly_env import AssemblyEnv, Shape
    from assembly_gym.envs.gym_env import Action, AssemblyGym, sparse_reward
    trap, hexa = Shape(urdf_file="shapes/trapezoid.urdf"), Shape(urdf_file="shapes/hexagon.urdf")
    env = AssemblyGym(shapes=[trap], targets=[], obstacles=[], reward_fct=sparse_reward, restrict_2d=True,
                      assembly_env=AssemblyEnv(render=False))
    a = Action(-1, 0, 0, 0, -1.0, 0.0)
    b1 = env.create_blocks([a])[0]
    assert env.create_blocks([a])[0] is b1                      # same state: served from the cache
    assert b1.verts_2d.shape[0] == 4 and b1.shape is trap
    env.reset(shapes=[hexa])
    b2 = env.create_blocks([a])[0]
    assert b2 is not b1 and b2.shape is hexa and b2.verts_2d.shape[0] == 6
    env.reset(shapes=[trap])
    b3 = env.create_blocks([a])[0]
    assert b3 is not b1 and b3.shape is trap and np.array_equal(b3.verts_2d, b1.verts_2d)
    env.step(a)
    assert env.create_blocks([a])[0] is not b3                  # the assembly changed: a new candidate set


def test_render_blocks_2d_at_the_reference_default_and_at_non_square_sizes():
    """render_blocks_2d's default img_size is (512, 512) (rendering.py:105); any size goes through bridges_render_blocks,
    pixel for pixel the oracle's image.  A non-square size comes back the way the reference returns it: the [img_size[1],
    img_size[0]] grid of rows re-interpreted (reshape, not transpose) as img_size."""
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, bridge_setup, sparse_reward
    from assembly_gym.utils.rendering import render_blocks_2d
    from oracle import raster as o_raster
    from oracle.env import OracleGym
    from oracle.env import bridge_setup as o_bridge_setup
    env = AssemblyGym(**bridge_setup(num_stories=2, hexagon=True), reward_fct=sparse_reward, restrict_2d=True, max_steps=10,
                      assembly_env=AssemblyEnv(render=False))
    og = OracleGym(**o_bridge_setup(num_stories=2, hexagon=True), max_steps=10)
    acts = [(-1, 0, 0, 3, -1.3333333333333335, 0.0), (0, 1, 1, 0, 0.0, 0.0), (-1, 0, 1, 0, -2.0, 0.0)]
    assert render_blocks_2d([], (-3, 7), (0, 10)).shape == (512, 512) and not render_blocks_2d([], (-3, 7), (0, 10)).any()
    for a in acts:
        obs = env.step(Action(*a))[0]
        og.step(a)
        img = render_blocks_2d(obs["blocks"], xlim=(-3, 7), ylim=(0, 10))                        # the reference's default size
        want = o_raster.render_blocks_2d(og.blocks, (-3, 7), (0, 10), (512, 512))
        assert img.dtype == bool and img.shape == (512, 512) and np.array_equal(img, want) and want.any()
        for size in ((128, 128), (65, 65), (200, 120), (96, 160)):
            got = render_blocks_2d(obs["blocks"], xlim=(-3, 7), ylim=(0, 10), img_size=size)
            ref = o_raster.render_blocks_2d(og.blocks, (-3, 7), (0, 10), size)                       # [size[1], size[0]]
            assert ref.shape == (size[1], size[0]) and got.shape == size
            assert np.array_equal(got, ref.reshape(size))
        # 64 x 64 through the bit rasteriser and through the per-pixel operator: the same image
        from bridges_hip import ops
        assert np.array_equal(ops.render_blocks(obs["blocks"], (-3, 7), (0, 10), (64, 64)).cpu().numpy().astype(bool),
                              render_blocks_2d(obs["blocks"], xlim=(-3, 7), ylim=(0, 10), img_size=(64, 64)))


def test_frozen_and_free_verdicts_of_a_step_come_from_one_operator_call(monkeypatch):
    """AssemblyGym.step solves the state with the new block frozen and stabilities_freezing() then asks for it free
    (gym_env.py:238-243, :325-333 of the reference): both frozen sets ride in ONE bridges_stability call
    (ops.stability_variants, AssemblyEnv._solve_state) -- same verdicts as one call each, and as the oracle."""
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import Action, AssemblyGym, bridge_setup, sparse_reward
    from bridges_hip import ops
    from oracle.env import OracleGym
    from oracle.env import bridge_setup as o_bridge_setup
    calls = []
    inner = ops.stability_variants
    monkeypatch.setattr(ops, "stability_variants", lambda blocks, sets, *a: calls.append(len(sets)) or inner(blocks, sets, *a))
    env = AssemblyGym(**bridge_setup(num_stories=2), reward_fct=sparse_reward, restrict_2d=True, max_steps=10,
                      assembly_env=AssemblyEnv(render=False))
    og = OracleGym(**o_bridge_setup(num_stories=2), max_steps=10)
    for a in [(-1, 0, 0, 3, -1.3333333333333335, 0.0), (0, 1, 0, 3, 0.0, 0.0), (1, 2, 0, 0, 0.0, 0.0), (2, 1, 0, 3, 0.6, 0.0)]:
        before = len(calls)
        obs, *_ = env.step(Action(*a))
        og.step(a)
        pair = env.stabilities_freezing()
        assert len(calls) == before + 1 and calls[-1] == 2, calls             # one call, two frozen sets
        assert pair == og.stabilities_freezing() and obs["stable"] == pair[0]
        ae = env.assembly_env
        n = len(ae.blocks)
        fixed = {i for i, b in enumerate(ae.blocks) if b.is_static}
        single = [ops.stability(ae.blocks, f, ae.mu, ae.density, ae.floor_half_width, ae.floor_depth) for f in (fixed, fixed - {n - 1})]
        both = inner(ae.blocks, [fixed, fixed - {n - 1}], ae.mu, ae.density, ae.floor_half_width, ae.floor_depth)
        assert both == single and (single[0][0], single[1][0]) == pair


@pytest.mark.parametrize("task,episodes", [("tower2", 30), ("mixed", 20), ("hexbridge", 20)])
def test_random_episodes_through_the_drop_in_surface_match_the_numpy_oracle(task, episodes):
    """tests/stress/stress_single_env.py: random-policy episodes through AssemblyGym + generate_actions / filter_actions + the feature
    functions against oracle/env.py (numpy + HiGHS) -- candidate lists, filter masks, candidate rasters, linear rewards, every
    step's stable flag / reward / termination, stabilities_freezing(), state rasters, distance_to_targets."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress", "stress_single_env.py"), "--task", task, "--episodes", str(episodes),
                          "--seed", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    m = re.search(r"RESULT .*: (\d+) env-steps .* (\d+) mismatches", out.stdout)
    assert m and int(m[1]) > 2 * episodes and int(m[2]) == 0, out.stdout[-1000:]

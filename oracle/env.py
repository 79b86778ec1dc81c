"""Oracle: single-environment restatement of the assembly_gym task logic and of
the per-step feature pipeline of ``rollout_episode``.

Follows
* ``AssemblyGym.step / reset / terminated / _update_targets / stabilities_freezing``
  (assembly_gym/assembly_gym/envs/gym_env.py:141-333), ``sparse_reward`` (:11-22),
  the task setups (:25-99);
* ``generate_actions`` / ``filter_actions`` (robotoddler/utils/actions.py:7-52, 71-82)
  with ``collision_on_action`` (gym_env.py:304-323);
* ``get_state_features / get_task_features / get_action_features`` and the
  ``lin_reward`` rule of ``rollout_episode``
  (robotoddler/training/successor_dqn.py:47-94, 395-411).

Pure Python + numpy, one environment, no attempt at speed.
"""
import numpy as np

from . import raster as R
from .geometry import Block, create_block
from .rbe import is_stable_rbe
from .shapes import get_shape

DEFAULT_BOUNDS = ((-3.0, -3.0, -1.0), (7.0, 7.0, 9.0))     # assembly_env.py:168
XLIM = (-3.0, 7.0)                                          # successor_dqn.py:615
YLIM = (0.0, 10.0)                                          # successor_dqn.py:616
X_DISCR_GROUND = np.linspace(-2, 0, 10)                     # successor_dqn.py:611
OFFSET_VALUES = (0.0,)                                      # successor_dqn.py:613


# ---- task setups (gym_env.py:25-99) ---------------------------------------

def _shape_list(trapezoid=True, hexagon=False):
    shapes = []
    if trapezoid:
        shapes.append(get_shape("trapezoid"))
    if hexagon:
        shapes.append(get_shape("hexagon"))
    return shapes


def horizontal_bridge_setup(square_size=0.6, num_obstacles=5, trapezoid=True, hexagon=False):
    reward_x = num_obstacles * square_size + 2.5 * square_size
    targets = [(reward_x, 0, square_size / 2)]
    obstacles = [(i * square_size, 0, square_size / 2) for i in range(1, num_obstacles + 1)]
    return dict(shapes=_shape_list(trapezoid, hexagon), obstacles=obstacles, targets=targets)


def bridge_setup(H=.8, num_stories=1, trapezoid=True, hexagon=False):
    targets = [(0.5, 0, num_stories * H + H / 2)]
    obstacles = [(targets[0][0], 0., i * H + H / 2) for i in range(num_stories)]
    return dict(shapes=_shape_list(trapezoid, hexagon), obstacles=obstacles, targets=targets)


def hard_tower_setup():
    shapes = [get_shape("trapezoid"), get_shape("cube1", receiving_faces_2d=[0], target_faces_2d=[2])]
    return dict(shapes=shapes, obstacles=[[0, 0, 2.0]], targets=[[0, 0, 0.5], [0, 0, 5.5]])


# ---- counter RNG shared with the device policy ----------------------------

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def policy_draw(seed, env_id, counter):
    """Uniform u64 for (seed, env, counter): the synthetic random policy."""
    return splitmix64(splitmix64(((seed & 0xFFFFFFFF) << 32) | (env_id & 0xFFFFFFFF)) ^ (counter & M64))


# ---- the environment -------------------------------------------------------

class OracleGym:
    def __init__(self, shapes, obstacles, targets, mu=0.8, density=1.0, bounds=DEFAULT_BOUNDS,
                 max_steps=None, xlim=XLIM, ylim=YLIM, img_size=(64, 64),
                 x_discr_ground=X_DISCR_GROUND, offset_values=OFFSET_VALUES):
        self.shapes = list(shapes)
        self.obstacles = [tuple(float(v) for v in o) for o in obstacles]
        self.targets = [tuple(float(v) for v in t) for t in targets]
        self.mu, self.density, self.bounds = mu, density, bounds
        self.max_steps = max_steps
        self.xlim, self.ylim, self.img_size = xlim, ylim, img_size
        self.x_discr_ground = [float(v) for v in x_discr_ground]
        self.offset_values = [float(v) for v in offset_values]
        cube06 = get_shape("cube06")                        # gym_env.py:277, successor_dqn.py:73
        self.obstacle_blocks = [Block(cube06, (o[0], o[2])) for o in self.obstacles]
        self.target_blocks = [Block(cube06, (t[0], t[2])) for t in self.targets]
        # task features (successor_dqn.py:67-85)
        self.obstacle_raster = R.render_blocks_2d(self.obstacle_blocks, xlim, ylim, img_size)
        tr = R.render_blocks_2d(self.target_blocks, xlim, ylim, img_size).astype(np.float32)
        self.reward_map = R.convolve_with_gaussian(tr, 101, 16)
        self.reset()

    # gym_env.py:255-289
    def reset(self):
        self.blocks = []
        self.block_graph = {(-1, 0): []}
        self.targets_reached = []
        self.targets_remaining = list(self.targets)
        self.frozen = None
        self.stable = True          # AssemblyEnv.reset -> is_stable_rbe on empty assembly (stability.py:53-56)

    # gym_env.py:218-253
    def step(self, action, honour_frozen_flag=False):
        tb, tf, sh, f = action[:4]
        new_block = create_block(self.shapes, self.blocks, action)
        self.blocks.append(new_block)
        nb = len(self.blocks) - 1
        self.block_graph.setdefault((tb, tf), []).append((nb, f))
        self.block_graph[(nb, f)] = [(tb, tf)]
        # previous frozen block is released (gym_env.py:235-236), new one frozen (:238-240)
        freeze = True
        if honour_frozen_flag:      # the recorded golden table predates the forced freeze (:238)
            freeze = bool(action[6]) if len(action) > 6 else False
        self.frozen = nb if freeze else None
        # gym_env.py:163-169 removes from targets_remaining WHILE iterating over it: CPython's list iterator then skips
        # the element that moves into the freed position, i.e. the open target after each reached one is not tested for
        # this block (it can still be reached by a later block).  Reproduced literally.
        for t in self.targets_remaining:
            if new_block.aabb_contains(t):
                self.targets_reached.append(t)
                self.targets_remaining.remove(t)
        self.stable = self._rbe(self.frozen)
        terminated = (not self.stable) or len(self.targets_remaining) == 0
        truncated = bool(self.max_steps and len(self.blocks) >= self.max_steps)
        return self.stable, self.sparse_reward(), terminated, truncated

    def _rbe(self, frozen):
        fixed = set() if frozen is None else {frozen}
        return is_stable_rbe(self.blocks, fixed, self.mu, self.density, self.bounds)

    # gym_env.py:11-22
    def sparse_reward(self):
        if not self.stable:
            return -1
        n = len(self.targets_reached)
        if len(self.targets_remaining) != 0:
            return -1 + n
        return n

    # gym_env.py:325-333
    def stabilities_freezing(self):
        return self._rbe(len(self.blocks) - 1), self._rbe(None)

    def is_action_stable(self, action):
        """is_action_stable_rbe (assembly_gym/assembly_gym/utils/stability.py:122-130): the candidate block is appended
        (free), boundary conditions of the existing blocks stay (the last placed block is frozen)."""
        blocks = self.blocks + [create_block(self.shapes, self.blocks, action)]
        fixed = set() if self.frozen is None else {self.frozen}
        return is_stable_rbe(blocks, fixed, self.mu, self.density, self.bounds)

    def distance_to_targets(self):          # gym_env.py:154-161
        if not self.blocks:
            return [float("inf")] * len(self.targets)
        return [min(b.distance_to_point(t) for b in self.blocks) for t in self.targets]

    # actions.py:7-52
    def generate_actions(self):
        acts = []
        for si, shape in enumerate(self.shapes):
            for face in shape.target_faces_2d:
                for ox in self.x_discr_ground:
                    acts.append((-1, 0, si, face, ox, 0.0))
                for tb, block in enumerate(self.blocks):
                    # Block(...) drops the shape's receiving_faces_2d (assembly_env.py:153)
                    for tf in range(block.shape.num_faces_2d):
                        if len(self.block_graph.get((tb, tf), ())) >= 1:
                            continue
                        for ox in self.offset_values:
                            acts.append((tb, tf, si, face, ox, 0.0))
        return acts

    # gym_env.py:304-323
    def collision_on_action(self, block):
        eps = 1e-6
        for vx, vz in block.verts:
            if vx < self.xlim[0] - eps or vx > self.xlim[1] + eps or vz < self.ylim[0] - eps or vz > self.ylim[1] + eps:
                return True
        for vx, vz in block.verts:
            if vz < -eps:
                return True
        return False

    def state_raster(self):
        return R.render_blocks_2d(self.blocks, self.xlim, self.ylim, self.img_size)

    def candidates(self):
        """Everything rollout_episode computes for the current state
        (successor_dqn.py:403-411): raw candidate list, their blocks, rasters,
        the filter mask and the linear reward of each candidate."""
        acts = self.generate_actions()
        blocks = [create_block(self.shapes, self.blocks, a) for a in acts]
        X, Y = R.pixel_grid(self.xlim, self.ylim, self.img_size)
        rasters = np.zeros((len(acts),) + (len(Y), len(X)), dtype=bool)
        for i, b in enumerate(blocks):
            rasters[i] = R.contains_2d(b, X, Y)
        state = self.state_raster()
        mask = np.zeros(len(acts), dtype=bool)
        lin = np.zeros(len(acts), dtype=np.float32)
        for i, b in enumerate(blocks):
            ok = not self.collision_on_action(b)
            ok = ok and not (rasters[i] & state).any() and not (rasters[i] & self.obstacle_raster).any()
            mask[i] = ok
            lin[i] = np.float32(self.reward_map[rasters[i]].astype(np.float64).sum())
        return dict(actions=acts, blocks=blocks, rasters=rasters, mask=mask, lin_reward=lin, state=state)


class OracleLockstep:
    """One environment driven by the same lock-step protocol as the device
    VecAssemblyGym (DESIGN.md "Lock-step protocol"): every call of
    ``lockstep(pick)`` is either a real env-step (place the picked candidate,
    both stability variants, reward/termination, auto-reset when done) or a
    reset-only step (when the previous state had no valid candidate)."""

    def __init__(self, gym, capacity=16):
        self.gym = gym
        self.capacity = capacity      # block slots of the device state: a state that fills them is truncated
        self.gym.reset()
        self.cand = self.gym.candidates()
        self.needs_reset = not self.cand["mask"].any()

    def lockstep(self, pick_valid_rank):
        """``pick_valid_rank(n_valid) -> rank`` chooses among the valid candidates."""
        g = self.gym
        out = dict(valid_step=False)
        if self.needs_reset:
            g.reset()
        else:
            valid = np.flatnonzero(self.cand["mask"])
            a = int(valid[pick_valid_rank(len(valid))])
            action = self.cand["actions"][a]
            stable, reward, term, trunc = g.step(action)
            trunc = trunc or len(g.blocks) >= self.capacity
            fs, us = g.stabilities_freezing()
            base = float(self.cand["lin_reward"][a])
            lin = base if us else (np.float32(base) / np.float32(100) if fs else 0.0)   # successor_dqn.py:397-401
            out = dict(valid_step=True, action_index=a, action=action, stable_frozen=fs, stable_unfrozen=us,
                       reward=reward, lin_reward=float(lin), terminated=term, truncated=trunc,
                       done=bool(term or trunc), n_blocks=len(g.blocks),
                       targets_reached=len(g.targets_reached),
                       pose=(g.blocks[-1].pos, g.blocks[-1].cs))
            if out["done"]:
                g.reset()
        self.cand = g.candidates()
        self.needs_reset = not self.cand["mask"].any()
        out["no_actions"] = self.needs_reset
        return out

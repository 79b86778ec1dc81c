"""AssemblyGym, Action, sparse_reward and the task setups with the reference's surface
(assembly_gym/assembly_gym/envs/gym_env.py:11-333 of the reference).  gymnasium itself is not required: the class
keeps the gymnasium calling convention (reset -> (obs, info); step -> (obs, reward, terminated, truncated, info)).
Placement runs in bridges_create_block, stability in bridges_stability."""
from dataclasses import dataclass

import numpy as np

from assembly_gym.envs.assembly_env import AssemblyEnv, Block, Quaternion, Shape
from assembly_gym.utils.geometry import distance_box_point
from bridges_hip import ops


def sparse_reward(gym_env, obs, info):                                   # gym_env.py:11-22
    """-1 for a collapsed / colliding assembly; otherwise the number of targets reached, minus one while any is left."""
    state = gym_env.assembly_env.state_info
    if state['collision'] or not state['stable']:
        return -1
    reached = len(obs['targets_reached'])
    return reached if gym_env.all_targets_reached() else reached - 1


def _shapes(trapezoid, hexagon):
    shapes = []
    if trapezoid:
        shapes.append(Shape(urdf_file='shapes/trapezoid.urdf', name="trapezoid"))
    if hexagon:
        shapes.append(Shape(urdf_file='shapes/hexagon.urdf', name="hexagon"))
    return shapes


def horizontal_bridge_setup(square_size=0.6, num_obstacles=5, trapezoid=True, hexagon=False):   # gym_env.py:25-43
    reward_x = num_obstacles * square_size + 2.5 * square_size
    targets = [(reward_x, 0, square_size / 2)]
    obstacles = [(i * square_size, 0, square_size / 2) for i in range(1, num_obstacles + 1)]
    return dict(shapes=_shapes(trapezoid, hexagon), obstacles=obstacles, targets=targets)


def bridge_setup(H=.8, num_stories=1, trapezoid=True, hexagon=False):                          # gym_env.py:46-61
    targets = [(0.5, 0, num_stories * H + H / 2)]
    obstacles = [(targets[0][0], 0., i * H + H / 2) for i in range(num_stories)]
    return dict(shapes=_shapes(trapezoid, hexagon), obstacles=obstacles, targets=targets)


def tower_setup(num_targets=3, targets=None):                                                   # gym_env.py:64-79
    if targets is None:
        targets = [(np.random.uniform(-4, 4), 0, np.random.uniform(0., 4)) for _ in range(num_targets)]
    return dict(shapes=[Shape(urdf_file='shapes/trapezoid.urdf', name="trapezoid")], obstacles=[], targets=targets)


def hard_tower_setup():                                                                         # gym_env.py:82-88
    trapezoid = Shape(urdf_file='shapes/trapezoid.urdf', name="trapezoid")
    cube = Shape(urdf_file='shapes/cube1.urdf', name="cube", receiving_faces_2d=[0], target_faces_2d=[2])
    return dict(shapes=[trapezoid, cube], targets=[[0, 0, 0.5], [0, 0, 5.5]], obstacles=[[0, 0, 2.0]])


def connecting_setup():                                                                         # gym_env.py:91-99
    rectangle = Shape(urdf_file='shapes/block.urdf', name="rectangle", receiving_faces_2d=[3], target_faces_2d=[0])
    cube = Shape(urdf_file='shapes/cube1.urdf', name="cube", receiving_faces_2d=[3], target_faces_2d=[1])
    u = np.random.uniform
    targets = [[u(0.4, 0.6), 0, 0.175] for _ in range(3)]
    obstacles = [[u(0.4, 0.47), 0, u(0.025, 0.125)], [u(0.53, 0.6), 0, u(0.025, 0.125)]]
    return dict(shapes=[rectangle, cube], obstacles=obstacles, targets=targets)


@dataclass
class Action:                                                                                   # gym_env.py:102-110
    target_block: int
    target_face: int
    shape: int
    face: int
    offset_x: float = 0.
    offset_y: float = 0.
    frozen: bool = False


class AssemblyGym:
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, reward_fct, shapes=None, obstacles=None, targets=None, render_mode=None, assembly_env=None,
                 restrict_2d=False, max_steps=None):
        if not restrict_2d:
            raise NotImplementedError          # gym_env.py:131-133: only the 2-D mode exists
        self.blocks, self.shapes, self.obstacles, self.targets = [], [], [], []
        self.reward_fct, self.render_mode, self.restrict_2d, self.max_steps = reward_fct, render_mode, restrict_2d, max_steps
        self.observation_space = self.action_space = None       # gymnasium attributes, unused like in the reference
        self.action_history = self.block_graph = None
        self._block_cache, self._block_cache_state, self._generation = {}, None, 0
        self.assembly_env = assembly_env if assembly_env is not None else AssemblyEnv(render=render_mode == 'human')
        self.reset(shapes, obstacles, targets)

    def terminated(self, assembly_env):
        """(terminated, truncated); truncated is None / 0 when max_steps is unset (gym_env.py:141-145)."""
        state = assembly_env.state_info
        failed = state['collision'] or not state['stable']
        return failed or self.all_targets_reached(), self.max_steps and len(self.blocks) >= self.max_steps

    @property
    def num_targets(self):
        return len(self.targets)

    @property
    def num_obstacles(self):
        return len(self.obstacles)

    def distance_to_targets(self):
        if len(self.assembly_env.blocks) == 0:
            return self.num_targets * [np.inf]
        return [min(distance_box_point(b.bounding_box, t) for b in self.assembly_env.blocks) for t in self.targets]

    def _update_targets(self, new_block):
        # A target counts as reached once a block's bounding box holds it.  The reference removes from the list it is
        # iterating over (gym_env.py:164-168), which makes CPython skip the open target that follows a reached one (it
        # stays open for later blocks); kept, because reward and termination depend on it.
        still_open, skip = [], False
        for t in self.targets_remaining:
            if not skip and new_block.bounding_box.contains_point(t):
                self.targets_reached.append(t)
                skip = True
            else:
                still_open.append(t)
                skip = False
        self.targets_remaining = still_open

    def all_targets_reached(self):
        return len(self.targets_remaining) == 0

    def _get_obs(self):
        si = self.assembly_env.state_info
        return {
            'blocks': self.blocks,
            'stable': bool(si['stable']),
            'collision': bool(si['collision']),
            'collision_block': bool(si['collision_info']['blocks']),
            'collision_obstacle': bool(si['collision_info']['obstacles']),
            'collision_floor': bool(si['collision_info']['floor']),
            'collision_boundary': bool(si['collision_info']['bounding_box']),
            'frozen_block': self.assembly_env.frozen_block_index,
            'obstacles': self.obstacles,
            'obstacle_blocks': self.assembly_env.obstacles,
            'targets': self.targets,
            'targets_remaining': self.targets_remaining,
            'targets_reached': self.targets_reached,
            'distance_to_targets': self.distance_to_targets(),
        }

    def _get_info(self):
        return {'blocks_initial_state': None, 'blocks_final_state': None}

    def create_block(self, action: Action):
        return self.create_blocks([action])[0]

    def create_blocks(self, actions):
        """create_block (gym_env.py:204-216 of the reference) for a list of actions in one batched operator call.  The
        blocks of the current state's candidates are kept until the assembly changes: the reference creates every
        candidate twice per env-step (get_action_features, successor_dqn.py:69, then collision_on_action through
        filter_actions, actions.py:71-82) -- the second round is served from here."""
        # the cache belongs to ONE state: the block list (by identity, in order), the shape list and the episode (reset()
        # starts a new generation, so a reset to other shapes or a recycled id() can never serve an old episode's blocks)
        state = (self._generation, tuple(id(b) for b in self.assembly_env.blocks), tuple(id(s) for s in self.shapes))
        if self._block_cache_state != state:
            self._block_cache, self._block_cache_state = {}, state
        keys = [(a.target_block, a.target_face, a.shape, a.face, float(a.offset_x), float(a.offset_y)) for a in actions]
        todo = [i for i, k in enumerate(keys) if k not in self._block_cache]
        if todo:
            acts = [actions[i] for i in todo]
            targets = [None if a.target_block == -1 else self.assembly_env.blocks[a.target_block] for a in acts]
            shapes = [self.shapes[a.shape] for a in acts]
            posed = ops.create_blocks(targets, [a.target_face for a in acts], [sh.geometry for sh in shapes],
                                      [a.face for a in acts], [a.offset_x for a in acts], [a.offset_y for a in acts])
            for i, sh, (pose, verts, frames) in zip(todo, shapes, posed):
                self._block_cache[keys[i]] = Block(sh, position=[pose[0], 0.0, pose[1]], _posed=(pose, verts, frames))
        return [self._block_cache[k] for k in keys]

    def step(self, action: Action):
        new_block = self.create_block(action)
        self.assembly_env.add_block(new_block)
        self.action_history.append(action)
        self.blocks.append(new_block)
        new_block_index = len(self.assembly_env.blocks) - 1
        key = (action.target_block, action.target_face)
        self.block_graph.setdefault(key, []).append((new_block_index, action.face))
        self.block_graph[(new_block_index, action.face)] = [key]
        if len(self.assembly_env.blocks) > 1 and self.assembly_env.blocks[-2].is_static:
            self.assembly_env.unfreeze_block(len(self.assembly_env.blocks) - 2)
        action.frozen = True                   # gym_env.py:238: the last block is always frozen
        self.assembly_env.freeze_block(new_block_index)
        self._update_targets(new_block)
        self.assembly_env._update_state_info()
        done = self.terminated(self.assembly_env)
        obs, info = self._get_obs(), self._get_info()
        return (obs, self.reward_fct(self, obs, info), *done, info)

    def reset(self, shapes=None, obstacles=None, targets=None, blocks=None):
        self.assembly_env.reset()
        self._block_cache, self._block_cache_state = {}, None          # candidate blocks of the previous episode are void
        self._generation += 1
        self.action_history, self.targets_reached = [], []
        self.block_graph = {(-1, 0): []}                        # (block, face) -> what is attached there; floor = (-1, 0)
        self.blocks = blocks if blocks is not None else []
        for name, value in (("shapes", shapes), ("obstacles", obstacles), ("targets", targets)):
            if value is not None:
                setattr(self, name, value)
        self.targets_remaining = list(self.targets)
        small_cube = Shape(urdf_file='shapes/cube06.urdf')
        for b in self.blocks:
            self.assembly_env.add_block(Block(self.shapes[b[-1]], b[:3], Quaternion(*b[3:7])))
        for position in self.obstacles:
            self.assembly_env.add_obstacle(Block(shape=small_cube, position=position))
        return self._get_obs(), self._get_info()

    @property
    def num_step(self):
        return len(self.action_history)

    def render(self):
        raise NotImplementedError

    def close(self):
        self.assembly_env.disconnect_client()

    def collision_on_action(self, action, xlim, ylim):                   # gym_env.py:304-323
        block = self.create_block(action)
        eps = 1e-6
        for vx, vz in block.verts_2d:
            if vx < xlim[0] - eps or vx > xlim[1] + eps or vz < ylim[0] - eps or vz > ylim[1] + eps:
                return True
            if vz < -eps:
                return True
        return False

    def stabilities_freezing(self):                                      # gym_env.py:325-333
        n = len(self.assembly_env.blocks) - 1
        self.assembly_env._update_state_info()
        stable = self.assembly_env.is_stable()
        self.assembly_env.unfreeze_block(n)
        self.assembly_env._update_state_info()
        unfreezestable = self.assembly_env.is_stable()
        self.assembly_env.freeze_block(n)
        self.assembly_env._update_state_info()
        return stable, unfreezestable

"""is_stable_rbe / is_action_stable_rbe (assembly_gym/assembly_gym/utils/stability.py:49-71, 122-130 of the
reference) on the HIP contact + simplex kernel (bridges_stability)."""
from bridges_hip import ops


def is_stable_rbe(assembly_env):
    fixed = {i for i, b in enumerate(assembly_env.blocks) if b.is_static}
    stable, info = ops.stability(assembly_env.blocks, fixed, assembly_env.mu, assembly_env.density,
                                 assembly_env.floor_half_width, assembly_env.floor_depth)
    if stable is None:                       # solver error -> (None, {error}) (stability.py:66-68)
        return None, info
    return stable, None


def is_action_stable_rbe(gym_env, action):
    block = gym_env.create_block(action)
    gym_env.assembly_env.blocks.append(block)
    try:
        stable, _ = is_stable_rbe(gym_env.assembly_env)
    finally:
        gym_env.assembly_env.blocks.pop()
    return stable

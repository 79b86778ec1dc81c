"""Stability predicates of assembly_gym/assembly_gym/utils/stability.py on the HIP contact + simplex kernel:

* ``is_stable_rbe`` / ``is_action_stable_rbe`` (stability.py:49-71, 122-130 of the reference) -- the path's predicate,
  pinned by the reference's recorded outputs (tests/golden);
* ``is_stable_rbe_penalty`` (stability.py:75-88: compas_cra ``rbe_solve(penalty=True)`` + ``maximum_tension <= tol``)
  restated as "an equilibrium whose total tension is <= tol exists" on the same LP with one pulling column per contact
  point.  The reference holds no recorded output of this variant: PARITY UNPINNED (checked against this repo's oracle
  only);
* ``is_stable_cra`` / ``is_stable_cra_penalty`` (stability.py:91-119): the coupled rigid-block analysis of compas_cra is a
  non-convex program (virtual displacements, complementarity between contact forces and separation) solved by IPOPT; the
  fork the reference pins is not vendored, nothing records its outputs, and it is not used by the training path.  Not
  built: calling them raises NotImplementedError.
"""
from bridges_hip import ops


def _fixed(assembly_env):
    return {i for i, b in enumerate(assembly_env.blocks) if b.is_static}


def is_stable_rbe(assembly_env):
    stable, info = ops.stability(assembly_env.blocks, _fixed(assembly_env), assembly_env.mu, assembly_env.density,
                                 assembly_env.floor_half_width, assembly_env.floor_depth)
    if stable is None:                       # solver error -> (None, {error}) (stability.py:66-68)
        return None, info
    return stable, None


def is_stable_rbe_penalty(assembly_env, tol=1e-3):
    """(stable, {'max_tension': ...}) as stability.py:75-88.  ``max_tension`` is the largest net tension of a contact
    point in the equilibrium found (<= tol by construction) or None when no equilibrium within the tolerance exists --
    the reference reports the tension of its penalty optimum there, which this feasibility formulation does not produce."""
    from assembly_gym.utils.geometry import maximum_tension
    stable, info = ops.stability(assembly_env.blocks, _fixed(assembly_env), assembly_env.mu, assembly_env.density,
                                 assembly_env.floor_half_width, assembly_env.floor_depth, tension_tol=tol)
    if stable is None:
        return None, info
    return stable, {'max_tension': maximum_tension(info["forces"]) if stable else None}


def is_stable_cra(assembly_env):
    raise NotImplementedError("compas_cra's CRA solve (stability.py:91-104) is not part of this build: see the module docstring")


def is_stable_cra_penalty(assembly_env, tol=1e-3):
    raise NotImplementedError("compas_cra's CRA penalty solve (stability.py:106-119) is not part of this build: see the module docstring")


def is_action_stable_rbe(gym_env, action):
    block = gym_env.create_block(action)
    gym_env.assembly_env.blocks.append(block)
    try:
        stable, _ = is_stable_rbe(gym_env.assembly_env)
    finally:
        gym_env.assembly_env.blocks.pop()
    return stable

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


def _is_stable_rbe_variants(assembly_env, fixed_sets):
    """is_stable_rbe's answers for the environment's blocks under each frozen set of ``fixed_sets`` (one operator call)."""
    res = ops.stability_variants(assembly_env.blocks, fixed_sets, assembly_env.mu, assembly_env.density,
                                 assembly_env.floor_half_width, assembly_env.floor_depth)
    return [(None, info) if stable is None else (stable, None) for stable, info in res]


is_stable_rbe.variants = _is_stable_rbe_variants


def is_stable_rbe_penalty(assembly_env, tol=1e-3):
    """(stable, {'max_tension': ...}) as stability.py:75-88; an assembly without any contact returns (no free block, None)
    like the reference (:77-81).  PARITY UNPINNED -- the reference holds no output of this variant -- and not the same
    predicate near the threshold: the reference solves the penalty QP and tests the LARGEST per-point tension of that one
    optimum against tol; this build asks whether an equilibrium whose TOTAL tension is <= tol exists (a feasibility
    question the same simplex answers).  Total <= tol implies every point <= tol, so a True here is an equilibrium the
    reference's test would accept; several points pulling a little each (each <= tol, sum > tol) pass there and fail here.
    ``max_tension`` is the largest net tension of a contact point in the equilibrium found, or None when none exists within
    the tolerance (the reference reports the tension of its penalty optimum there, which a feasibility solve does not have)."""
    from assembly_gym.utils.geometry import maximum_tension
    stable, info = ops.stability(assembly_env.blocks, _fixed(assembly_env), assembly_env.mu, assembly_env.density,
                                 assembly_env.floor_half_width, assembly_env.floor_depth, tension_tol=tol)
    if stable is None:
        return None, info
    if info.get("n_interfaces", 0) == 0:                 # no edges: (no free node, None), stability.py:77-81
        return stable, None
    return stable, {'max_tension': maximum_tension(info["forces"]) if stable else None}


def is_stable_cra(assembly_env):
    raise NotImplementedError("compas_cra's CRA solve (stability.py:91-104) is out of scope of this build (DESIGN.md section 9: non-convex "
                              "program of an un-pinned fork, no recorded output to pin a restatement to)")


def is_stable_cra_penalty(assembly_env, tol=1e-3):
    raise NotImplementedError("compas_cra's CRA penalty solve (stability.py:106-119) is out of scope of this build (DESIGN.md section 9)")


def is_action_stable_rbe(gym_env, action):
    block = gym_env.create_block(action)
    gym_env.assembly_env.blocks.append(block)
    try:
        stable, _ = is_stable_rbe(gym_env.assembly_env)
    finally:
        gym_env.assembly_env.blocks.pop()
    return stable

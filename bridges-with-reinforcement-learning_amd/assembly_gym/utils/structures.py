"""Known-answer structures (the reference's assembly_gym/utils/structures.py:22-108: seven small assemblies with a
hand-derived stability label per build step), data-driven: the action lists and label rules live in
``assets/structures.json``; every builder returns ``(env, [(Action, expected_is_stable), ...])`` like the reference's.

``replay(env, actions)`` places the blocks one by one honouring each action's ``frozen`` flag (the semantics the
recorded table of notebooks/Stability Evaluation.ipynb was produced with; AssemblyGym.step at HEAD freezes every new
block, gym_env.py:238) and yields the stability verdict after every step."""
import json
import os

from assembly_gym.envs.assembly_env import AssemblyEnv, Shape
from assembly_gym.envs.gym_env import Action, AssemblyGym, sparse_reward
from assembly_gym.utils.stability import is_stable_rbe

_DATA = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "structures.json")))
STRUCTURE_NAMES = [k for k in _DATA if not k.startswith("_")]


def create_env(mu, density, shapes):
    env = AssemblyGym(shapes=shapes, targets=[], obstacles=[], reward_fct=sparse_reward, restrict_2d=True,
                      assembly_env=AssemblyEnv(render=False, mu=mu, density=density, stability=None))
    env.reset()
    return env


def _expected(rule, mu, freeze_last, offset_y):
    (kind, val), = rule.items()
    return {"always": lambda: bool(val), "freeze_last": lambda: bool(freeze_last), "mu_gt": lambda: mu > val,
            "freeze_last_or_mu_gt": lambda: bool(freeze_last) or mu > val,
            "freeze_last_and_mu_gt": lambda: bool(freeze_last) and mu > val,
            "offset_y_lt": lambda: offset_y < val,
            "freeze_last_or_offset_y_lt": lambda: bool(freeze_last) or offset_y < val}[kind]()


def build(name, mu=0.8, density=1.0, **kwargs):
    spec = _DATA[name]
    kw = dict(spec["defaults"], **kwargs)
    unknown = set(kw) - set(spec["defaults"])
    if unknown:
        raise TypeError(f"{name}() got unexpected keyword arguments {sorted(unknown)}")
    freeze_last, offset_y = kw.get("freeze_last", False), kw.get("offset_y", 0.0)
    env = create_env(mu, density, [Shape(urdf_file=f"shapes/{s}.urdf", name=s) for s in spec["shapes"]])
    rows = list(zip(spec["actions"], spec["expected"]))
    if name == "tower":
        rows = rows[:kw["num_blocks"]] if kw["num_blocks"] <= len(rows) else \
            rows + [([i - 1, 0, 0, 3, 0, 0, False], {"always": True}) for i in range(len(rows), kw["num_blocks"])]
    actions = []
    for a, rule in rows:
        oy = offset_y if (name == "levitating_block" and a[0] == -1) else a[5]
        frozen = bool(freeze_last) if a[6] == "FL" else bool(a[6])
        actions.append((Action(a[0], a[1], a[2], a[3], a[4], oy, frozen), _expected(rule, mu, freeze_last, offset_y)))
    return env, actions


def _builder(name):
    def f(mu=0.8, density=1.0, **kwargs):
        return build(name, mu=mu, density=density, **kwargs)
    f.__name__ = name
    f.__doc__ = f"Known-answer structure '{name}' (defaults {_DATA[name]['defaults']})."
    return f


hexagon = _builder("hexagon")
trapezoid_bridge = _builder("trapezoid_bridge")
hexagon_bridge_3 = _builder("hexagon_bridge_3")
hexagon_bridge_5 = _builder("hexagon_bridge_5")
horizontal_bridge = _builder("horizontal_bridge")
tower = _builder("tower")
levitating_block = _builder("levitating_block")


def replay(env, actions, method=is_stable_rbe, **method_kwargs):
    """Yield (step, action, expected, (is_stable, extra)) while building the structure block by block."""
    asm = env.assembly_env
    for step, (action, expected) in enumerate(actions):
        block = env.create_block(action)
        if asm.blocks and asm.blocks[-1].is_static:
            asm.unfreeze_block(len(asm.blocks) - 1)            # the previous frozen block is released (gym_env.py:235-236)
        asm.blocks.append(block)
        if action.frozen:
            asm.freeze_block(len(asm.blocks) - 1)
        yield step, action, expected, method(asm, **method_kwargs)

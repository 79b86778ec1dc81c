"""Regression harness over the known-answer structures (the reference's assembly_gym/utils/test_suite.py): every
structure x build step x stability method -> ``<output_path>/<md5 of the structure>/structure.json`` with the same
keys (``structure``, ``methods``, ``tests``; entries addressed by md5 hashes of their parameters).  Methods: ``rbe`` and
``rbe_penalty`` (test_suite.py:32-38 of the reference also lists pybullet, cra and cra_penalty, which this build does not
have); no plots are written.
    python -m assembly_gym.utils.test_suite --output_path out/ --mu 0.8"""
import argparse
import hashlib
import json
import os
import time

from assembly_gym.utils import structures
from assembly_gym.utils.stability import is_stable_rbe, is_stable_rbe_penalty

STRUCTURES = [("hexagon_bridge_3", dict(freeze_last=True)), ("hexagon_bridge_5", dict(freeze_last=True)),
              ("trapezoid_bridge", dict(freeze_last=True)), ("trapezoid_bridge", dict(freeze_last=False)),
              ("horizontal_bridge", dict(freeze_last=False)), ("horizontal_bridge", dict(freeze_last=True)),
              ("hexagon", dict()), ("tower", dict(num_blocks=10)),
              ("levitating_block", dict()), ("levitating_block", dict(freeze_last=True))]
METHODS = [("rbe", is_stable_rbe, dict()), ("rbe_penalty", is_stable_rbe_penalty, dict(tol=1e-3))]


def compute_hash(**kwargs):
    return hashlib.md5(json.dumps(dict(**kwargs), sort_keys=True).encode("utf-8")).hexdigest()


def run(output_path, mu=0.8, density=1.0, recompute_existing=False):
    written = []
    for name, kw in STRUCTURES:
        env, actions = structures.build(name, mu=mu, density=density, **kw)
        path = os.path.join(output_path, compute_hash(__name__=name, **kw))
        os.makedirs(path, exist_ok=True)
        json_path = os.path.join(path, "structure.json")
        data = json.load(open(json_path)) if os.path.exists(json_path) else \
            dict(structure=dict(name=name, kwargs=kw, plots_env={}, plots_cra={}), methods={}, tests={})
        for mname, _, mkw in METHODS:
            data["methods"][compute_hash(name=mname, **mkw)] = dict(name=mname, kwargs=mkw)
        for step, action, expected, _ in structures.replay(env, actions, method=lambda asm: (None, None)):
            test = data["tests"].setdefault(compute_hash(mu=mu, density=density, step=step),
                                            dict(step=step, is_stable=bool(expected), mu=mu, density=density))
            for mname, method, mkw in METHODS:
                mid = compute_hash(name=mname, **mkw)
                if mid in test and not recompute_existing:
                    continue
                t = time.time()
                res, extra = method(env.assembly_env, **mkw)
                test[mid] = dict(is_stable=res, extra=extra, time=time.time() - t)
        with open(json_path, "w") as f:
            json.dump(data, f, indent=4, default=str)
        written.append(json_path)
    return written


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--output_path", type=str, required=True)
    p.add_argument("--recompute_existing", action="store_true")
    p.add_argument("--density", type=float, default=1.0)
    p.add_argument("--mu", type=float, default=0.8)
    a = p.parse_args(argv)
    for path in run(a.output_path, mu=a.mu, density=a.density, recompute_existing=a.recompute_existing):
        print(path)


if __name__ == "__main__":
    main()

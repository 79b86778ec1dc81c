"""Generates tests/golden/large_assemblies.json with the numpy + HiGHS oracle (nothing from the reference runs):
assemblies of up to 15 blocks grown by a stability-seeking policy, for several tasks / friction values, each with the
oracle's stability boolean and infeasibility value v* for two fixed sets (last block frozen, nothing frozen).
These exercise what random rollouts rarely reach: 30-45 equilibrium rows, 100+ LP columns, the tableau overflow
path, mesh-noise-level infeasibilities.   python tests/golden/make_large_assemblies.py
With --reeval the assemblies already in the file are kept and only their verdicts are recomputed (after a change of
the oracle's predicate)."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.env import OracleGym, bridge_setup, horizontal_bridge_setup      # noqa: E402
from oracle.geometry import create_block                                     # noqa: E402
from oracle.rbe import is_stable_rbe                                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
if "--reeval" in sys.argv:
    from oracle.geometry import Block
    from oracle.shapes import get_shape
    path = os.path.join(HERE, "large_assemblies.json")
    recs = json.load(open(path))
    flips = 0
    for r in recs:
        shapes = [get_shape(n) for n in r["shapes"]]
        blocks = [Block(shapes[s], (p[0], p[1]), (p[2], p[3])) for s, p in zip(r["shape_ids"], r["poses"])]
        for c in r["cases"]:
            st, info = is_stable_rbe(blocks, set(c["fixed"]), r["mu"], return_info=True)
            if bool(st) != c["stable"]:
                flips += 1
                print("verdict changed:", r["task"], len(blocks), "blocks fixed", c["fixed"], c["stable"], "->", bool(st),
                      "v", c["v"], "->", info["v"])
            c.update(stable=bool(st), v=info["v"], n_if=info["n_if"])
    json.dump(recs, open(path, "w"))
    print(len(recs), "assemblies re-evaluated,", flips, "verdicts changed")
    sys.exit(0)
TASKS = [("tower4", bridge_setup, dict(num_stories=4), ["trapezoid"], 0.8, 14),
         ("hexbridge", horizontal_bridge_setup, dict(num_obstacles=5, trapezoid=False, hexagon=True), ["hexagon"], 0.8, 10),
         ("mixed_mu2", horizontal_bridge_setup, dict(num_obstacles=5, trapezoid=True, hexagon=True), ["trapezoid", "hexagon"], 2.0, 8),
         ("bridge_mu05", horizontal_bridge_setup, dict(num_obstacles=5), ["trapezoid"], 0.5, 10)]
out = []
for name, fn, kw, shapes, mu, n_env in TASKS:
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    for e in range(n_env):
        g = OracleGym(**fn(**kw), max_steps=15, mu=mu)
        while len(g.blocks) < 15:
            cand = g.candidates()
            valid = np.flatnonzero(cand["mask"])
            if len(valid) == 0:
                break
            chosen = None
            for a in rng.permutation(valid)[:6]:
                act = cand["actions"][a]
                blocks = g.blocks + [create_block(g.shapes, g.blocks, act)]
                if is_stable_rbe(blocks, {len(blocks) - 1}, mu):
                    chosen = act
                    break
            if chosen is None:
                chosen = cand["actions"][rng.permutation(valid)[0]]        # record one unstable extension too
            g.step(chosen)
            g.targets_remaining = [(99., 0., 99.)]
            nb = len(g.blocks)
            if nb >= 6:
                rec = dict(task=name, shapes=shapes, mu=mu, shape_ids=[g.shapes.index(b.shape) for b in g.blocks],
                           poses=[[b.pos[0], b.pos[1], b.cs[0], b.cs[1]] for b in g.blocks], cases=[])
                for fixed in ([nb - 1], []):
                    st, info = is_stable_rbe(g.blocks, set(fixed), mu, return_info=True)
                    rec["cases"].append(dict(fixed=fixed, stable=bool(st), v=info["v"], n_if=info["n_if"]))
                out.append(rec)
            if not g.stable:
                break
# regression cases found by tests/stress/stress_parity.py are kept across regenerations
path = os.path.join(HERE, "large_assemblies.json")
if os.path.exists(path):
    out += [r for r in json.load(open(path)) if r["task"].startswith("regression_")]
json.dump(out, open(path, "w"))
vs = [c["v"] for r in out for c in r["cases"] if c["v"] is not None]
print(len(out), "assemblies,", sum(len(r["cases"]) for r in out), "cases; blocks max", max(len(r["poses"]) for r in out),
      "n_if max", max(c["n_if"] for r in out for c in r["cases"]), "stable", sum(c["stable"] for r in out for c in r["cases"]))
print("v classes: <=1e-6:", sum(v <= 1e-6 for v in vs), " (1e-6,1e-4):", sum(1e-6 < v < 1e-4 for v in vs), " >=1e-4:", sum(v >= 1e-4 for v in vs))

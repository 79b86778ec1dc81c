"""Large assemblies (6-15 blocks, up to 31 interfaces = 124 LP columns, 45 equilibrium rows) from
tests/golden/large_assemblies.json (made by the numpy + HiGHS oracle, see make_large_assemblies.py).

CPU: the fixture is consistent with the oracle that made it and no case sits in the tolerance gap.
GPU: bridges_stability (contact detection from scratch + simplex incl. the tableau overflow path) reproduces every
boolean."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def fixture(golden_dir):
    return json.load(open(os.path.join(golden_dir, "large_assemblies.json")))


def test_fixture_is_well_separated_and_reproducible(fixture):
    from oracle.geometry import Block
    from oracle.rbe import is_stable_rbe
    from oracle.shapes import get_shape
    assert len(fixture) >= 100
    vs = [c["v"] for r in fixture for c in r["cases"] if c["v"] is not None]
    assert not any(1e-6 < v < 1e-4 for v in vs)            # nothing near the 1e-5 decision threshold
    assert max(len(r["poses"]) for r in fixture) == 15
    assert any(r["task"].startswith("regression_") for r in fixture)      # a noise-pivot case a 1e-7 pivot tolerance got wrong
    for r in fixture[::9]:
        shapes = [get_shape(n) for n in r["shapes"]]
        blocks = [Block(shapes[s], (p[0], p[1]), (p[2], p[3])) for s, p in zip(r["shape_ids"], r["poses"])]
        for c in r["cases"]:
            assert is_stable_rbe(blocks, set(c["fixed"]), r["mu"]) == c["stable"]


@pytest.mark.gpu
def test_gpu_stability_on_large_assemblies(fixture):
    import ctypes as C
    from bridges_hip import abi
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import ShapeTable, _ptr, _stream
    L = abi.require_gpu()
    dev = torch.device("cuda")
    K = abi.MAX_BLOCKS
    by_task = {}
    for r in fixture:
        by_task.setdefault((tuple(r["shapes"]), r["mu"]), []).append(r)
    checked = overflow_cases = 0
    for (shape_names, mu), recs in by_task.items():
        geoms = [load_urdf(f"shapes/{n}.urdf") for n in shape_names]
        table = ShapeTable(geoms)
        rows = [(r, c) for r in recs for c in r["cases"]]
        n = len(rows)
        pose = torch.zeros((n, K, 4), dtype=torch.float64)
        shape = torch.zeros((n, K), dtype=torch.int32)
        nb = torch.zeros(n, dtype=torch.int32)
        fixed = torch.zeros(n, dtype=torch.int32)
        for i, (r, c) in enumerate(rows):
            k = len(r["poses"])
            pose[i, :k] = torch.tensor(r["poses"], dtype=torch.float64)
            shape[i, :k] = torch.tensor(r["shape_ids"], dtype=torch.int32)
            nb[i] = k
            fixed[i] = sum(1 << b for b in c["fixed"])
        pose, shape, nb, fixed = pose.to(dev), shape.to(dev), nb.to(dev), fixed.to(dev)
        verts = torch.zeros((n, K, 6, 2), dtype=torch.float64, device=dev)
        flat_shape = shape.reshape(-1).contiguous()
        abi.check(L.bridges_pose_block(table.ptr, n * K, _ptr(flat_shape), _ptr(pose), _ptr(verts), _stream()))
        ws_stride = abi.lp_ws_stride(K)
        ws = torch.empty((n, ws_stride), dtype=torch.float64, device=dev)
        stable = torch.zeros(n, dtype=torch.uint8, device=dev)
        info = torch.zeros((n, 8), dtype=torch.float64, device=dev)
        abi.check(L.bridges_stability(table.ptr, n, K, _ptr(pose), _ptr(verts), _ptr(shape), _ptr(nb), _ptr(fixed), mu, 1.0,
                                      5.0, 10.0, _ptr(stable), _ptr(info), _ptr(ws), ws_stride, _stream()))
        got, inf = stable.cpu().numpy().astype(bool), info.cpu().numpy()
        for i, (r, c) in enumerate(rows):
            assert inf[i, 3] == 0, "solver error"
            assert int(inf[i, 1]) == c["n_if"], (r["task"], i)              # same contact interfaces as the oracle
            assert got[i] == c["stable"], (r["task"], i, inf[i, 0], c["v"])
            checked += 1
            n_free = len(r["poses"]) - len(c["fixed"])
            overflow_cases += (3 * n_free + 2) * (4 * c["n_if"] + 3) > 2048   # tableau did not fit LDS
    assert checked == sum(len(r["cases"]) for r in fixture)
    assert overflow_cases > 20

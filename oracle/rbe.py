"""Oracle: contact interfaces + rigid-block-equilibrium (RBE) feasibility.

Restates ``AssemblyEnv._reset_cra_assembly`` (assembly_gym/assembly_gym/envs/
assembly_env.py:281-304: floor slab + blocks + ``assembly_interfaces_numpy(amin=0.001)``)
and ``is_stable_rbe`` (assembly_gym/assembly_gym/utils/stability.py:49-71 ->
``compas_cra.equilibrium.rbe_solve(mu, density, penalty=False)``).

compas_cra (git+https://github.com/kirschnj/compas_cra, un-pinned fork,
docker/cscs/requirements.txt:6) is NOT in /root/reference; its published
algorithm (Kao et al. 2022, "Coupled Rigid-Block Analysis") is restated in 2-D:

* interfaces: for every body pair, every pair of faces whose outward normals
  are anti-parallel (n_A.n_B <= -1 + 1e-6), coplanar (|(c_B - c_A).n_A| <= 1e-6)
  and overlap tangentially by area >= amin gives two contact points (the ends
  of the overlap);
* RBE: per contact point a normal force f_n >= 0 and a tangential force with
  |f_t| <= mu f_n; every non-fixed block is in force and moment equilibrium
  under gravity (weight = density * volume).  The reference asks IPOPT whether
  that set is non-empty (ValueError("infeasible") -> unstable).  For y-extruded
  prisms the 3-D set (4 vertices, 8-facet pyramid; shapes pinned by
  notebooks/CRA_Assembly.ipynb cell 3: Aeq (6,12), Afr (32,12)) is non-empty
  iff this 2-D one is (mirror-average in y).

With the cone written in generators, force = a (n + mu t) + b (n - mu t),
a, b >= 0, stability is the standard-form feasibility problem
``exists x >= 0 : M x = w, 1'x <= S_MAX``.  The oracle measures the L1 distance
to feasibility  v* = min 1'(e+ + e-) s.t. M x + e+ - e- = w, 1'x <= S_MAX, x >= 0
with HiGHS and declares stable iff v* <= FEAS_TOL.  (v* is unique even when x
is not, which is what makes the boolean comparable across solvers.)

S_MAX is a budget on the total contact force (sum of the cone-generator
multipliers).  Block weights are O(1..30); genuine equilibria need a total of
1..1e3 (friction wedges near their critical angle form a thin tail up to
~1e4).  The float32 STL vertices however leave ~1e-7 of direction noise in
nominally parallel faces, and an unbounded LP can balance a block on that noise
with forces of 1e5..1e12 x its weight (observed: v* = 0 with max x = 6.4e7 while
v* = 2.95..3.00 for every bound <= 1e6).  That is a numerical artefact, not an
equilibrium, and no finite-precision solver decides it reliably.  With the
budget row the question is well conditioned: over 400 000 decisions of random
rollouts (tests/stress/stress_c_vs_highs.py) no LP has v* in (1e-6, 1e-4) and the two
simplex implementations (plain C, HIP) agree with HiGHS on every one.
"""
import numpy as np

FLOOR = -1
TOL_PARALLEL = 1e-6
TOL_COPLANAR = 1e-6
AMIN = 0.001
S_MAX = 1e4       # budget on the sum of contact-force multipliers at density 1, see module docstring
FEAS_TOL = 1e-5   # float32 meshes leave ~1e-7 geometric noise (observed classes: 0, 3e-7 | 7.8e-4, >= 2.8e-2)
# Both are forces: they are multiplied by the density (the right-hand side w is linear in it, M does not depend on it),
# so the boolean does not depend on the unit of mass -- AssemblyEnv(density=...) is a public input
# (assembly_gym/assembly_gym/envs/assembly_env.py:164).


def floor_body(bounds=((-3.0, -3.0, -1.0), (7.0, 7.0, 9.0))):
    """assembly_env.py:290-296: slab width = bounds span, centred on the origin."""
    width = bounds[1][0] - bounds[0][0]
    depth = bounds[1][1] - bounds[0][1]
    hw = width / 2.0
    va, vb = (-hw, 0.0), (hw, 0.0)
    return dict(faces=[(va, vb, ((0.0, 0.0), (1.0, 0.0), (0.0, 1.0)))], depth=depth)


def block_body(block):
    faces = []
    for (ia, ib), fr in zip(block.shape.faces, block.frames):
        faces.append((block.verts[ia], block.verts[ib], fr))
    return dict(faces=faces, depth=block.shape.depth)


def face_pair_contact(fa, fb, depth, amin=AMIN):
    """Contact segment of face ``fa`` (body A) with face ``fb`` (body B) or None.

    Arithmetic contract (mirrored by the HIP kernel):
      dotn = nA.x*nB.x + nA.z*nB.z                        ; reject if dotn > -1 + 1e-6
      gap  = (cB.x-cA.x)*nA.x + (cB.z-cA.z)*nA.z          ; reject if |gap| > 1e-6
      a0,a1 = ((vaA-cA).tA, (vbA-cA).tA) ; b0,b1 likewise with B's end points
      lo = max(min(a0,a1), min(b0,b1)) ; hi = min(max(a0,a1), max(b0,b1))
      reject if (hi - lo) * depth < amin
      p_lo = cA + lo*tA ; p_hi = cA + hi*tA
    """
    vaA, vbA, (cA, tA, nA) = fa
    vaB, vbB, (cB, _tB, nB) = fb
    dotn = nA[0] * nB[0] + nA[1] * nB[1]
    if dotn > -1.0 + TOL_PARALLEL:
        return None
    gap = (cB[0] - cA[0]) * nA[0] + (cB[1] - cA[1]) * nA[1]
    if abs(gap) > TOL_COPLANAR:
        return None

    def proj(v):
        return (v[0] - cA[0]) * tA[0] + (v[1] - cA[1]) * tA[1]

    a0, a1, b0, b1 = proj(vaA), proj(vbA), proj(vaB), proj(vbB)
    lo = max(min(a0, a1), min(b0, b1))
    hi = min(max(a0, a1), max(b0, b1))
    if (hi - lo) * depth < amin:
        return None
    p_lo = (cA[0] + lo * tA[0], cA[1] + lo * tA[1])
    p_hi = (cA[0] + hi * tA[0], cA[1] + hi * tA[1])
    return p_lo, p_hi, nA, tA


def find_interfaces(blocks, bounds=((-3.0, -3.0, -1.0), (7.0, 7.0, 9.0))):
    """All interfaces, ordered by (body A, body B, face A, face B) with the floor
    as body -1.  Returns [(A, B, p_lo, p_hi, n, t)], n pointing from A into B."""
    bodies = [(FLOOR, floor_body(bounds))] + [(i, block_body(b)) for i, b in enumerate(blocks)]
    out = []
    for ia in range(len(bodies)):
        for ib in range(ia + 1, len(bodies)):
            A, bodyA = bodies[ia]
            B, bodyB = bodies[ib]
            depth = min(bodyA["depth"], bodyB["depth"])
            for fa in bodyA["faces"]:
                for fb in bodyB["faces"]:
                    c = face_pair_contact(fa, fb, depth)
                    if c is not None:
                        out.append((A, B) + c)
    return out


def equilibrium_system(blocks, interfaces, fixed, mu, density):
    """M (3*n_free x 4*n_if), w."""
    free = [i for i in range(len(blocks)) if i not in fixed]
    row = {b: 3 * k for k, b in enumerate(free)}
    M = np.zeros((3 * len(free), 4 * len(interfaces)))
    w = np.zeros(3 * len(free))
    for b in free:
        w[row[b] + 1] = density * blocks[b].weight_per_density
    for k, (A, B, p_lo, p_hi, n, t) in enumerate(interfaces):
        gens = ((n[0] + mu * t[0], n[1] + mu * t[1]), (n[0] - mu * t[0], n[1] - mu * t[1]))
        for ip, p in enumerate((p_lo, p_hi)):
            for ig, g in enumerate(gens):
                col = 4 * k + 2 * ip + ig
                for body, sign in ((B, 1.0), (A, -1.0)):
                    if body in row:
                        r = row[body]
                        gx, gz = sign * g[0], sign * g[1]
                        cx, cz = blocks[body].centroid
                        rx, rz = p[0] - cx, p[1] - cz
                        M[r, col] += gx
                        M[r + 1, col] += gz
                        M[r + 2, col] += rx * gz - rz * gx
    return M, w


def infeasibility(M, w, s_max=S_MAX):
    """v* = min ||M x - w||_1 over x >= 0, sum(x) <= s_max (HiGHS)."""
    from scipy.optimize import linprog
    m, n = M.shape
    if m == 0:
        return 0.0
    if n == 0:
        return float(np.abs(w).sum())
    A = np.hstack([M, np.eye(m), -np.eye(m)])
    c = np.concatenate([np.zeros(n), np.ones(2 * m)])
    budget = np.concatenate([np.ones(n), np.zeros(2 * m)])[None, :]
    res = linprog(c, A_eq=A, b_eq=w, A_ub=budget, b_ub=[s_max], bounds=(0, None), method="highs")
    if res.status != 0:
        raise RuntimeError(f"HiGHS failed on an always-feasible LP: {res.message}")
    return float(res.fun)


def is_stable_rbe(blocks, fixed, mu=0.8, density=1.0, bounds=((-3.0, -3.0, -1.0), (7.0, 7.0, 9.0)),
                  return_info=False):
    """stability.py:49-71.  ``fixed`` = set of block indices with is_static."""
    fixed = set(fixed)
    interfaces = find_interfaces(blocks, bounds)
    n_free = len(blocks) - len([b for b in fixed if 0 <= b < len(blocks)])
    if len(interfaces) == 0:                      # stability.py:53-56
        stable = n_free == 0
        return (stable, dict(v=None, n_if=0)) if return_info else stable
    M, w = equilibrium_system(blocks, interfaces, fixed, mu, density)
    v = infeasibility(M, w, S_MAX * density)
    stable = v <= FEAS_TOL * density
    return (stable, dict(v=v, n_if=len(interfaces))) if return_info else stable


def is_stable_rbe_penalty(blocks, fixed, mu=0.8, density=1.0, tol=1e-3, bounds=((-3.0, -3.0, -1.0), (7.0, 7.0, 9.0)),
                          return_info=False):
    """stability.py:75-88 (compas_cra rbe_solve(penalty=True) + maximum_tension <= tol), restated as a feasibility
    question like is_stable_rbe: every contact point may additionally PULL with t_p >= 0 along -n; stable iff an
    equilibrium exists with  sum x + (S_MAX / tol) sum t <= S_MAX  (i.e. total tension <= tol).  No output of the
    reference pins this variant -- PARITY UNPINNED.  Also returns the true minimum total tension (HiGHS)."""
    from scipy.optimize import linprog
    fixed = set(fixed)
    interfaces = find_interfaces(blocks, bounds)
    n_free = len(blocks) - len([b for b in fixed if 0 <= b < len(blocks)])
    if len(interfaces) == 0:
        stable = n_free == 0
        return (stable, dict(v=None, min_tension=None)) if return_info else stable
    M, w = equilibrium_system(blocks, interfaces, fixed, mu, density)
    m, n = M.shape
    if m == 0:
        return (True, dict(v=0.0, min_tension=0.0)) if return_info else True
    free = [i for i in range(len(blocks)) if i not in fixed]
    row = {b: 3 * k for k, b in enumerate(free)}
    N = np.zeros((m, 2 * len(interfaces)))
    for k, (A, B, p_lo, p_hi, nrm, _t) in enumerate(interfaces):
        for ip, p in enumerate((p_lo, p_hi)):
            for body, sign in ((B, -1.0), (A, 1.0)):                     # tension pulls B towards A
                if body in row:
                    r = row[body]
                    gx, gz = sign * nrm[0], sign * nrm[1]
                    cx, cz = blocks[body].centroid
                    N[r, 2 * k + ip] += gx
                    N[r + 1, 2 * k + ip] += gz
                    N[r + 2, 2 * k + ip] += (p[0] - cx) * gz - (p[1] - cz) * gx
    nt = N.shape[1]
    s_max = S_MAX * density
    Aeq = np.hstack([M, N, np.eye(m), -np.eye(m)])
    cost = np.concatenate([np.zeros(n + nt), np.ones(2 * m)])
    budget = np.concatenate([np.ones(n), np.full(nt, s_max / tol), np.zeros(2 * m)])[None, :]
    res = linprog(cost, A_eq=Aeq, b_eq=w, A_ub=budget, b_ub=[s_max], bounds=(0, None), method="highs")
    if res.status != 0:
        raise RuntimeError(f"HiGHS failed on an always-feasible LP: {res.message}")
    v = float(res.fun)
    stable = v <= FEAS_TOL * density
    if not return_info:
        return stable
    # the true minimum total tension (any budget on the tension removed)
    res2 = linprog(np.concatenate([np.zeros(n), np.ones(nt)]), A_eq=np.hstack([M, N]), b_eq=w,
                   A_ub=np.concatenate([np.ones(n), np.zeros(nt)])[None, :], b_ub=[s_max], bounds=(0, None), method="highs")
    return stable, dict(v=v, min_tension=float(res2.fun) if res2.status == 0 else None)

"""CPU oracle for the assembly_gym / successor-DQN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product tree
(``bridges-with-reinforcement-learning_amd/``) imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may, and there only as the checker / timed CPU baseline.

It is a plain float64 restatement (numpy + scalar Python; HiGHS via
``scipy.optimize.linprog`` for the LP) of the reference's algorithm for the
path SURVEY.md §8 names, plus a second, independent restatement in plain C
(``oracle/c/oracle_env.c``: own simplex, used as the timed CPU baseline) that is
tested bit for bit against the first.  Every function cites the reference file:line it
follows (paths relative to ``/root/reference``).

Parity status
-------------
* Stability booleans: PINNED by the reference's own recorded outputs
  (``notebooks/Stability Evaluation.ipynb`` cell 2: 96 rows, of which 94
  reproduce and 2 are the documented IPOPT false negatives, see
  ``tests/golden/README.md``), plus the ``AssemblyEnv.ipynb`` /
  ``CRA_Assembly.ipynb`` known answers.  The predicate carries a budget on
  the total contact force (``oracle/rbe.py``: S_MAX) so that equilibria which
  exist only through float32 mesh noise are rejected the same way by every
  solver; none of the recorded rows is affected by it.
* Geometry (placement, AABB, distances): PINNED by the float prints in
  ``AssemblyEnv.ipynb`` cells 24-25 (``distance_to_targets``).
* Rasters, action masks, linear rewards of a full rollout: the reference holds
  no fixture for them and its third-party stack (compas, compas_cra, pyomo,
  ipopt) is not installable here -> "parity unpinned"; pinned only against
  this restatement.
* Q-networks / replay / TD target: PINNED by fixtures generated from the
  importable reference modules (``tests/golden/make_net_fixtures.py``).
"""

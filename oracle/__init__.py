"""CPU oracle for the space-time Kronecker hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in NumPy/SciPy (plus one small C file for the
sequential Gauss-Seidel sweep), the algorithms of the reference's hot path
(Jannertje/spacetime-fullgrid-parallel: source/mpi_vector.py, mpi_kron.py,
wavelets.py, multigrid.py, linalg.py, lanczos.py, heateq_mpi.py:126-191, and
the serial wiring heateq.py:18-107 in heat_serial.py).
Each function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  Nothing under
``spacetime-fullgrid-parallel_amd/`` imports it; the product path has no CPU
fallback and fails loudly when the HIP library is missing.

Pinning (see tests/test_oracle_golden.py and tests/golden/make_golden.py):
every function here is checked against golden vectors produced in the build
container by importing the reference's own classes (through a single-rank
mpi4py stand-in; mpi4py itself is not installable here) and against the
known-answer values in the reference's tests (wavelets_test.py:29-51,
mpi_kron_test.py:58-61).  One part is pinned less tightly and says so:
the smoother.  The reference smooths with PETSc ``MatSOR`` (multigrid.py:100-127,
petsc4py, unpinned version, absent here); the golden multigrid vectors were
produced with the reference's own ``MGM`` control flow (multigrid.py:168-193)
driving the reference's own pure-Python ``Smoother`` (multigrid.py:83-97) in
place of PETSc.  Parity at the PETSc boundary itself is unpinned.
heat_serial.py (the serial driver, which needs NGSolve to run in the reference)
is pinned through the rest: its Schur complement is checked to equal the
parallel wiring's, which the goldens pin (test_serial_wiring_equals_parallel_wiring).
"""

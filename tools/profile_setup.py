#!/usr/bin/env python3
"""cProfile of the operator construction (mesh, assembly, hierarchies, ELL copies, uploads)."""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import torch  # noqa: E402,F401
import heateq_mpi as hm  # noqa: E402

J_space = int(sys.argv[1]) if len(sys.argv) > 1 else 9
J_time = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pr = cProfile.Profile()
pr.enable()
h = hm.HeatEquationMPI(J_space=J_space, J_time=J_time)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)

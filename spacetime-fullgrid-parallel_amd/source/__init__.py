"""MI355X-native hot path of the space-time full-grid parabolic solver.

The module names mirror the reference's ``source`` package
(Jannertje/spacetime-fullgrid-parallel), so a driver written against the
reference imports the same classes from here:

    mpi_vector  DofDistributionMPI, KronVectorMPI
    mpi_kron    LinearOperatorMPI family (Kronecker-product operators)
    wavelets    wavelet-in-time transform
    multigrid   MeshHierarchy, MultiGrid (Gauss-Seidel V-cycles)
    linalg      PCG            lanczos  Lanczos
    linop       space operators, KronLinOp, CompositeLinOp, BlockDiagLinOp
    comm        torch.distributed (RCCL) facade replacing mpi4py
    mesh / assembly / problem   build-owned replacement of the NGSolve setup

All arithmetic runs in libstk.so (csrc/, include/stk.h); there is no CPU
fallback.
"""

/* stk.h -- C ABI of libstk, the MI355X (gfx950) hot path of the space-time
 * Kronecker solver.
 *
 * The reference (Jannertje/spacetime-fullgrid-parallel) has no FFI: its
 * boundary is the duck-typed Python protocol LinearOperatorMPI / KronVectorMPI
 * (reference source/mpi_kron.py:13-36, source/mpi_vector.py:41-122).  Every
 * entry point below names the reference method whose arithmetic it replaces;
 * the Python classes in spacetime-fullgrid-parallel_amd/source/ bind them with
 * ctypes and keep the reference's class names and call signatures.
 *
 * Conventions
 *  - extern "C", int status return (0 = ok, nonzero = error; text from
 *    stk_last_error()).  No exceptions cross the ABI.
 *  - All array arguments are DEVICE pointers unless the name ends in _host.
 *    The caller owns every buffer.  float64 values, int32 indices
 *    (reference source/mpi_shared_mem.py:46-48).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    work is enqueued asynchronously on it; nothing synchronises.
 *  - One caller thread per process, one process per GPU.
 *
 * Vector layout ("space-major"): the local slab of a KronVectorMPI, which the
 * reference stores as X_loc[t][i] (mpi_vector.py:62-71), lives on the device
 * as x[i * ld + t], i in [0, M), t in [0, n_loc), ld >= n_loc.  Entries
 * t in [n_loc, ld) are padding and are kept zero by every kernel.  A time
 * column is therefore contiguous: a CSR gather of neighbour j fetches n_loc
 * consecutive doubles.  Ghost time rows t = -1 and t = n_loc (the reference's
 * X_loc_bdr[0], X_loc_bdr[-1], mpi_vector.py:148-175) are separate contiguous
 * arrays of length M.
 */
#ifndef STK_H
#define STK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STK_MAX_TERMS 4

/* ---- library ------------------------------------------------------------ */
const char *stk_last_error(void);
int stk_version(void);
/* Device properties the host side sizes launches with (CUs, wavefront). */
int stk_device_info(int32_t *n_cu, int32_t *wave_size, int64_t *hbm_bytes);
/* Launch-geometry knobs for benchmarking sweeps ("kron_block": 0 = automatic,
 * or 256 / 512 / 1024 threads per workgroup).  Results never depend on them. */
int stk_set_tuning(const char *key, int32_t value);

/* ---- time slices of a slab (BlockDiagMPI._matvec with different operators per
 *      slice, mpi_kron.py:126-131: every distinct operator gets the slices it owns
 *      as one slab) --------------------------------------------------------------
 * gather:  y[i*ld_y + k] = x[i*ld_x + cols[k]], k < n_cols; the padding columns
 *          n_cols .. ld_y-1 of y are written as zero.
 * scatter: y[i*ld_y + cols[k]] = x[i*ld_x + k], k < n_cols; other columns of y
 *          are left alone.  cols: device int32; the caller guarantees that they
 *          are valid, distinct columns of the wider slab. */
int stk_slab_gather_columns(void *stream, int32_t M, int32_t n_cols,
                            const int32_t *cols, const double *x, int32_t ld_x,
                            double *y, int32_t ld_y);
int stk_slab_scatter_columns(void *stream, int32_t M, int32_t n_cols,
                             const int32_t *cols, const double *x, int32_t ld_x,
                             double *y, int32_t ld_y);

/* ---- slab storage for hosts that own no device allocator (KronVectorMPI.__init__
 *      / reset, mpi_vector.py:62-71; scatter / gather, :124-138) ------------------
 * stk_slab_alloc: a zeroed slab of M rows with leading dimension *ld =
 * stk_slab_ld(n_loc) (n_loc rounded up to even: time pairs are 16-byte aligned).
 * upload / download move the reference's time-major host block X_loc[t][i],
 * (n_loc, M) C order, to / from the space-major slab (a transpose on the device
 * through a staging buffer; padding columns are written as zero).  These two
 * calls block until the transfer is complete. */
int stk_slab_ld(int32_t n_loc);
int stk_slab_alloc(int32_t M, int32_t n_loc, int32_t *ld, double **slab);
int stk_slab_free(double *slab);
int stk_slab_upload(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                    const double *x_host, double *slab);
int stk_slab_download(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                      const double *slab, double *x_host);
/* dst[c * ld_dst + r] = src[r * ld_src + c] for r < rows, c < cols (device to
 * device, tiled through LDS); the columns rows .. zero_to-1 of dst are written
 * as zero (0: none).  Time-major block <-> slab, and the block transposes of
 * KronVectorMPI.permute (mpi_vector.py:212-240). */
int stk_transpose(void *stream, int32_t rows, int32_t cols, const double *src,
                  int64_t ld_src, double *dst, int64_t ld_dst, int32_t zero_to);
/* The two time rows a slab sends to its neighbour ranks in communicate_bdr
 * (mpi_vector.py:148-167): first[j * stride_first] = x[j][0], last[j *
 * stride_last] = x[j][n_loc-1], both from one pass over the slab's lines.
 * Stride 1: contiguous send buffers; stride 2: every other double of an
 * interleaved ghost buffer gh[j] = (lo, hi) (stk_kron_pack_apply), e.g. the
 * neighbour's own buffer mapped into this process.  Either pointer may be NULL. */
int stk_halo_pack(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                  const double *x, double *first, int32_t stride_first,
                  double *last, int32_t stride_last);
/* out[k * ld_out + j] = x[j][t_idx[k]], k < n_rows: the time rows communicate_dofs
 * sends (mpi_vector.py:189-203), all from one pass over the slab.  t_idx: device
 * int32, valid local time indices. */
/* stk_halo_pack, and per spatial dof the four entries the boundary steps of a Kronecker
 * apply read once the halo is there (stk_kron_pack_boundary_apply), from the same two
 * lines of the row: records[4*j .. 4*j+3] = (x[j][0], x[j][1], x[j][n_loc-2], x[j][n_loc-1]);
 * first / last may be NULL. */
int stk_halo_pack_records(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                          const double *x, double *first, int32_t stride_first,
                          double *last, int32_t stride_last, double *records);
int stk_slab_extract_time_rows(void *stream, int32_t M, int32_t n_rows,
                               const int32_t *t_idx, const double *x, int32_t ld,
                               double *out, int64_t ld_out);
/* dst[r * ld_dst + c] = src[r * ld_src + c], r < rows, c < cols: a block of rows
 * moved to another leading dimension (the pieces of the all-to-all transposes of
 * KronVectorMPI.permute and of the distributed wavelet transform). */
int stk_copy_block(void *stream, int64_t rows, int32_t cols, const double *src,
                   int64_t ld_src, double *dst, int64_t ld_dst);
/* y[i * ld + t] = u_t[t] * u_x[i] (padding zero): the right-hand side
 * u0_t kron u0_x of heateq_mpi.py:189-191 formed on the device. */
int stk_outer(void *stream, int32_t M, int32_t n_loc, int32_t ld,
              const double *u_t, const double *u_x, double *y);

/* ---- communication of the time-slab decomposition for hosts without
 *      torch.distributed: the reference's mpi4py call sites on
 *      RCCL over xGMI, one process per GPU --------------------------------------
 * RCCL is loaded at run time (librccl.so.1); nothing else in the library needs it.
 * Rank 0 obtains a unique id and the HOST distributes its STK_COMM_ID_BYTES bytes
 * to the other ranks (by whatever launched them: MPI, a file, a socket); every
 * rank then calls stk_comm_create on the device it computes on.  All calls are
 * enqueued on `stream`; buffers are device pointers of float64.
 *   stk_comm_allreduce_sum  KronVectorMPI.dot's allreduce (mpi_vector.py:209), in place
 *   stk_comm_halo_exchange  communicate_bdr (mpi_vector.py:140-187): send_first goes to
 *                           rank-1 (its X_loc_bdr[-1]), send_last to rank+1 (its
 *                           X_loc_bdr[0]); recv_lo / recv_hi receive the neighbours' rows
 *                           (M doubles each); pointers of a side without a neighbour are
 *                           ignored.  Feed it from stk_halo_pack.
 *   stk_comm_exchange       a batch of sends and receives as ONE group (communicate_dofs,
 *                           mpi_vector.py:189-203; the tiles of permute, :224-239);
 *                           messages between a pair of ranks match in posting order. */
#define STK_COMM_ID_BYTES 128
typedef struct stk_comm stk_comm;
typedef struct {
    void *buf;     /* device, float64 */
    int64_t count; /* doubles */
    int32_t peer;  /* rank */
} stk_comm_msg;
int stk_comm_unique_id(void *id_host);
int stk_comm_create(int32_t rank, int32_t size, const void *id_host, stk_comm **out);
int stk_comm_destroy(stk_comm *comm);
int stk_comm_allreduce_sum(stk_comm *comm, void *stream, double *values, int32_t n);
int stk_comm_halo_exchange(stk_comm *comm, void *stream, int32_t M,
                           const double *send_first, const double *send_last,
                           double *recv_lo, double *recv_hi);
int stk_comm_exchange(stk_comm *comm, void *stream, int32_t n_send,
                      const stk_comm_msg *sends_host, int32_t n_recv,
                      const stk_comm_msg *recvs_host);

/* ---- plan construction on the device (the set-up of MultiGrid.__init__,
 *      multigrid.py:130-166, that the reference does with SciPy on the host) --------
 * stk_ell_from_csr: the sliced-ELL copy (stk_ell_rows) of a CSR matrix resident on
 * the device.  ELL row `pos` is CSR row order[pos] (order NULL: pos); K slots per
 * row, unused ones = (pad_col, 0); pad_col < 0: the row's first kept column, or
 * the row itself if it keeps none.  strip_diag: the diagonal entry goes to dia_a /
 * dia_m only (Gauss-Seidel copies, diag_free); otherwise it stays among the slots
 * and dia_a / dia_m (either may be NULL) still receive it.  grp (or NULL): keep
 * only the entries whose column lies in an EARLIER dependency group than the row,
 * grp[col] < grp[row] (the zero-start copies of the first sweep of a level visit).
 * *overflow (device int32, zeroed by the caller) receives the longest row if one
 * has more than K entries.
 * stk_csr_galerkin: C = R A P for CSR matrices on the device, one row of C per
 * thread, every sum accumulated in the order SciPy's (R @ A) @ P accumulates it and
 * with separately rounded products and sums -- C is bit for bit the reference's
 * Galerkin matrix.  Row i: row_counts[i] entries at out_indices / out_data
 * [i*cap ...], columns ascending; cap <= 64; *overflow as above.  P = NULL (all
 * three arrays): C = R A alone, the product the restricted residual uses.
 * stk_gs_depth_step: one relaxation of the depth of every row in the dependency
 * DAG of a Gauss-Seidel sweep in dof order (forward: row i waits for its
 * neighbours j < i; backward: j > i), depth_out[i] = max depth_in[j] + 1;
 * *changed (device int32) is set when any depth moved.  The host repeats it from
 * depth = 0 until nothing changes; rows of equal depth are the groups mg.hip
 * launches together (source/multigrid.py gauss_seidel_schedule). */
int stk_ell_from_csr(void *stream, int32_t n_pos, int32_t K, const int32_t *order,
                     const int32_t *indptr, const int32_t *indices,
                     const double *vals_a, const double *vals_m, int32_t strip_diag,
                     const int32_t *grp, int32_t pad_col, int32_t *idx_out,
                     double *va_out, double *vm_out, double *dia_a, double *dia_m,
                     int32_t *overflow);
int stk_gs_depth_step(void *stream, int32_t n, const int32_t *indptr,
                      const int32_t *indices, int32_t backward,
                      const int32_t *depth_in, int32_t *depth_out, int32_t *changed);
int stk_csr_galerkin(void *stream, int32_t nc, const int32_t *r_indptr,
                     const int32_t *r_indices, const double *r_data,
                     const int32_t *a_indptr, const int32_t *a_indices,
                     const double *a_data, const int32_t *p_indptr,
                     const int32_t *p_indices, const double *p_data, int32_t cap,
                     int32_t *row_counts, int32_t *out_indices, double *out_data,
                     int32_t *overflow);

/* ---- BLAS-1 on flat arrays (KronVectorMPI arithmetic, mpi_vector.py:84-122,
 *      and dot, mpi_vector.py:205-210, local part) -------------------------- */
/* y = a * x + b * y   (b == 0 ignores the old y; x == y allowed) */
int stk_axpby(void *stream, int64_t n, double a, const double *x, double b,
              double *y);
/* z = a * x + b * y   (three-operand form, z may alias x or y) */
int stk_axpbyz(void *stream, int64_t n, double a, const double *x, double b,
               const double *y, double *z);
/* out[0] = sum x[k] * y[k].  Deterministic two-stage reduction; `work` holds
 * at least stk_dot_work_size() doubles. */
int64_t stk_dot_work_size(void);
int stk_dot(void *stream, int64_t n, const double *x, const double *y,
            double *work, double *out);
/* KronVectorMPI.dot (mpi_vector.py:205-210) with a value that does NOT depend on
 * the number of ranks.  The reference (and stk_dot + a scalar all-reduce) adds the
 * products slab by slab, so the last digits of every r.Pr of a solve move with
 * the partition of the time axis (the reference against itself on 8 ranks:
 * 4.6e-11 in the history).  stk_slab_dot sums the products of every TIME STEP over
 * the spatial index in one fixed shape that depends on M alone (csrc/blas1.hip:
 * blocks of 256 rows, 8 interleaved fused-multiply-add chains per block, pairwise
 * tree, fixed tree over the blocks) and writes them to out_steps[t_begin + t], t <
 * n_loc; the other entries of out_steps[0 .. N) are written as zero.  The caller
 * all-reduces the N values (adding zeros is exact in any order) and adds them in
 * increasing t (stk_sum_steps, host): bit for bit the same number on 1, 2, 4 or 8
 * ranks.  x, y: slabs (M rows of ld doubles, time fastest, padding ignored);
 * work: stk_slab_dot_work_size(M, n_loc) device doubles, 16-byte aligned. */
int64_t stk_slab_dot_work_size(int32_t M, int32_t n_loc);
int stk_slab_dot(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                 const double *x, const double *y, double *work, int32_t N,
                 int32_t t_begin, double *out_steps);
double stk_sum_steps(const double *steps_host, int32_t N);
/* ---- time-slab partition (DofDistributionMPI, mpi_vector.py:5-38) ---------
 * Rank p owns the time rows [displs[p], displs[p] + counts[p]): N / size rows
 * each, the N % size extra rows go to the last ranks.  Any output pointer may
 * be NULL; counts / displs hold `size` entries. */
int stk_partition(int32_t N, int32_t size, int32_t rank, int32_t *t_begin,
                  int32_t *t_end, int32_t *counts, int32_t *displs);

/* ---- preconditioned CG (PCG, linalg.py:6-42) ------------------------------
 * Solves T w = b on this rank's slab (n = M*ld doubles, padding zero) with the
 * reference's recurrence and stopping rule: stop when r.Pr < eps^2, at most
 * kmax-1 iterations, return immediately if b = 0 or the initial r.Pr is small.
 * The operators are callbacks (they launch on `stream`; T may communicate);
 * `allreduce` sums host doubles over the ranks (NULL on one rank).  `w` holds
 * the initial guess on entry.  `work`: stk_pcg_work_size(n) device doubles.
 * history (host, kmax entries or NULL) receives r.Pr after the initial
 * residual and after every iteration; *iters the iteration count. */
typedef int (*stk_operator_fn)(void *ctx, void *stream, const double *x,
                               double *y);
typedef int (*stk_allreduce_fn)(void *ctx, double *values, int32_t n);
int64_t stk_pcg_work_size(int64_t n);
int stk_pcg_solve(void *stream, int64_t n, stk_operator_fn T, void *T_ctx,
                  stk_operator_fn P, void *P_ctx, stk_allreduce_fn allreduce,
                  void *allreduce_ctx, const double *b, double *w, double eps,
                  int32_t kmax, double *work, double *history, int32_t *iters);
/* The same loop on slabs (M rows of ld doubles, this rank's time steps [t_begin,
 * t_begin + n_loc) of N) with stk_slab_dot for the inner products: `allreduce` is
 * then called with the N per-step sums, and history / iterate are bit for bit
 * those of a one-rank run, whatever the number of ranks (given operators that
 * have that property, as the Kronecker and multigrid applies of this library do).
 * work: stk_pcg_slab_work_size(M, n_loc, ld, N) device doubles. */
int64_t stk_pcg_slab_work_size(int32_t M, int32_t n_loc, int32_t ld, int32_t N);
int stk_pcg_solve_slab(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                       int32_t N, int32_t t_begin, stk_operator_fn T, void *T_ctx,
                       stk_operator_fn P, void *P_ctx, stk_allreduce_fn allreduce,
                       void *allreduce_ctx, const double *b, double *w, double eps,
                       int32_t kmax, double *work, double *history,
                       int32_t *iters);

/* ---- per-operation device-time counters (LinearOperatorMPI.num_applies /
 *      time_applies, mpi_kron.py:23-36, for hosts without the Python classes) -----
 * While enabled, every apply entry point brackets what it enqueues with two
 * HIP events on its stream.  stk_timing_get waits for the recorded events and
 * returns the number of calls and the summed device seconds of one class:
 * "kron" (stk_kron_*_apply), "space" (stk_csr_spmm, stk_ell_spmm), "time"
 * (stk_time_*_apply), "wavelet", "multigrid" (stk_mg_apply, stk_mg_smooth),
 * "blas1" (stk_axpby(z), stk_dot).  Classes nest where entry points do (the
 * BLAS-1 calls of stk_pcg_solve are counted as blas1).  Applies enqueued on
 * DIFFERENT streams are each timed on their own stream: where they overlap (the two
 * K applies inside S run side by side), the seconds of a class add up to more than
 * the wall time they took.  Communication time is the caller's. */
int stk_timing_enable(int32_t on);
int stk_timing_reset(void);
int stk_timing_get(const char *op_class, int64_t *calls, double *seconds);

/* ---- Lanczos estimate of the extreme eigenvalues of P A (Lanczos,
 *      lanczos.py:87-159; Sturm-sequence bisection bisec / pol, :20-85) ---------
 * The reference's recurrence on device vectors of n doubles: `w` holds the start
 * vector on entry (it is overwritten); A and P are callbacks as in
 * stk_pcg_solve.  Stops when lambda_max and lambda_min both moved by less than
 * `tol` relative (reference default 1e-4; bisection tolerance 1e-6), after at
 * most max_iterations steps (*converged = 0 then), or at a breakdown (invariant
 * Krylov space).  alpha_host / beta_host (max_iterations / max_iterations - 1
 * host doubles, or NULL) receive the recurrence coefficients.  work:
 * stk_lanczos_work_size(n) device doubles. */
int64_t stk_lanczos_work_size(int64_t n);
int stk_lanczos(void *stream, int64_t n, stk_operator_fn A, void *A_ctx,
                stk_operator_fn P, void *P_ctx, stk_allreduce_fn allreduce,
                void *allreduce_ctx, double *w, int32_t max_iterations,
                double tol, double tol_bisec, double *work, double *alpha_host,
                double *beta_host, double *lmax, double *lmin,
                int32_t *iterations, int32_t *converged);
/* On slabs, inner products through stk_slab_dot (see stk_pcg_solve_slab). */
int64_t stk_lanczos_slab_work_size(int32_t M, int32_t n_loc, int32_t ld, int32_t N);
int stk_lanczos_slab(void *stream, int32_t M, int32_t n_loc, int32_t ld, int32_t N,
                     int32_t t_begin, stk_operator_fn A, void *A_ctx,
                     stk_operator_fn P, void *P_ctx, stk_allreduce_fn allreduce,
                     void *allreduce_ctx, double *w, int32_t max_iterations,
                     double tol, double tol_bisec, double *work,
                     double *alpha_host, double *beta_host, double *lmax,
                     double *lmin, int32_t *iterations, int32_t *converged);

/* ---- sum of Kronecker terms  y = beta*y + sum_k (T_k kron X_k) x_k --------
 * Replaces TridiagKronMatMPI._matvec (mpi_kron.py:214-219), i.e.
 * TridiagKronIdentityMPI (:186-201) followed by IdentityKronMatMPI (:143-150),
 * and the accumulation loop of SumMPI._matvec (:77-90), in one pass.
 * All X_k share one CSR pattern (indptr/indices); vals differ per term. */
typedef struct {
    /* 3*n_loc coefficients [sub | diag | super]: output time row t takes
     * sub[t]*z[t-1] + diag[t]*z[t] + super[t]*z[t+1].  NULL = identity. */
    const double *tri;
    const double *vals; /* nnz values of X_k on the shared pattern */
    const double *x;    /* input vector, M x ld */
    const double *x_lo; /* ghost time row t = -1 (length M) or NULL */
    const double *x_hi; /* ghost time row t = n_loc (length M) or NULL */
} stk_kron_term;

/* CSR row `pos` produces output row row_ids[pos] (row_ids == NULL: pos itself).
 * The host lists the rows in the order it wants them processed (a mesh-tile or
 * RCM order keeps the gathers of one XCD inside its L2); the result does not
 * depend on that order.  Requires ld even and 16-byte aligned x, y. */
int stk_kron_sum_apply(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                       const int32_t *indptr, const int32_t *indices,
                       const int32_t *row_ids, int32_t n_terms,
                       const stk_kron_term *terms_host, double beta,
                       double *y);

/* The same operator on a sliced-ELL copy of the pattern, the fast path.
 * Every row owns K slots (K <= 16): ell_idx[pos*K + e] / ell_vals[pos*K + e];
 * unused slots hold the row's own column and the value 0.  Rows are listed in
 * processing order, row pos writes output row row_ids[pos].  Entries beyond K
 * of longer rows are kept in an overflow CSR indexed by pos (NULL if none). */
typedef struct {
    int32_t M, K;
    const int32_t *ell_idx;     /* M*K */
    const int32_t *row_ids;     /* M or NULL */
    const int32_t *ovf_indptr;  /* M+1 or NULL */
    const int32_t *ovf_indices;
} stk_ell_pattern;

typedef struct {
    const double *tri;      /* as in stk_kron_term */
    const double *ell_vals; /* M*K values of X_k */
    const double *ovf_vals; /* overflow values or NULL */
    const double *x, *x_lo, *x_hi;
} stk_kron_ell_term;

/* Tuning key (stk_set_tuning): "ell_wg_per_cu" (persistent workgroups per CU,
 * 0 = default).  K must be one of 5, 7, 9, 12, 16; at most 3 terms.
 * Terms with ghost rows (x_lo / x_hi) are handled as two launches: the
 * slab-local part, then stk_kron_ell_ghost_apply (with a copy of the old
 * boundary entries of y in between when beta != 0).  Every entry of y is
 * rounded as on one rank, wherever the time axis is cut. */
int stk_kron_ell_apply(void *stream, const stk_ell_pattern *pattern_host,
                       int32_t n_loc, int32_t ld, int32_t n_terms,
                       const stk_kron_ell_term *terms_host, double beta,
                       double *y);

/* The first and last local time step of the same operator once the ghost time
 * rows are there: "first/last rows after the halo arrived" of
 * TridiagKronIdentityMPI._matvec (mpi_kron.py:186-201, :199-200).  A caller that
 * overlaps the halo exchange with compute calls stk_kron_ell_apply with x_lo =
 * x_hi = NULL and beta = 0 while the exchange is in flight and this function
 * once it has completed.  The two steps are RECOMPUTED from x, x_lo and x_hi in
 * the order of operations of the main kernel,
 *   y[:, t] = sum_k fma(sup_k[t], z_k[t+1], fma(sub_k[t], z_k[t-1], dia_k[t] z_k[t])),
 * and overwrite what the ghost-less call left there: the result is bit for bit
 * the one-rank apply of the same rows, whatever the partition of the time axis
 * (a share ADDED afterwards, as in rounds 1-5, puts the neighbour's term last in
 * the sum and differs from the one-rank result in the last bits). */
int stk_kron_ell_ghost_apply(void *stream, const stk_ell_pattern *pattern_host,
                             int32_t n_loc, int32_t ld, int32_t n_terms,
                             const stk_kron_ell_term *terms_host, double *y);

/* The same operator with the matrix stream PACKED and the ghost time steps
 * fused into the one pass (the fast path of the metric's operator,
 * SumMPI([TridiagKronMatMPI, TridiagKronMatMPI])._matvec, mpi_kron.py:77-90,
 * with the halo of TridiagKronIdentityMPI._matvec, :186-201).
 * slots[pos*K + e] = code << col_bits | column; `code` indexes the dictionary of
 * distinct value tuples of the union pattern: matrix m has the value
 * dict[m*n_codes + code] in that slot.  Unused slots hold the row's own column
 * and a code whose values are all zero.  Exact: the dictionary holds the
 * doubles themselves.  Every term reads the same x; term k multiplies with
 * matrix t[k].mat.  `ghosts` (NULL on a slab without neighbours) holds the
 * ghost time steps interleaved: ghosts[2*j] = x[j, t = -1] (X_loc_bdr[0],
 * mpi_vector.py:148-175), ghosts[2*j + 1] = x[j, t = n_loc] (X_loc_bdr[-1]);
 * a side without a neighbour is zero-filled.  K one of 5, 7, 9, 12, 16; at most
 * 3 terms; slabs of any size (64-bit addressing).
 * Tuning keys: "pack_wg_per_cu", "pack_flags" (bit 0: non-temporal stores of y,
 * bit 1: non-temporal loads of the slot stream). */
typedef struct {
    int32_t M, K;
    int32_t col_bits; /* 2^col_bits >= M */
    int32_t n_codes;  /* <= 2^(32 - col_bits) */
    int32_t n_mats;
    int32_t rows_per_unit;  /* 1, or 2 = row pairs (below) */
    int32_t n_units;        /* slot rows: M when rows_per_unit = 1 */
    const uint32_t *slots;  /* n_units*K */
    const int32_t *row_ids; /* n_units*rows_per_unit; NULL (index order) only when rows_per_unit = 1 */
    const double *dict;     /* n_mats*n_codes*rows_per_unit */
    const double *vals;     /* NULL, or explicit values (no dictionary; rows_per_unit = 2 only):
                               n_units*K*rows_per_unit*n_mats, the slot words are then bare columns */
} stk_pack_pattern;
/* Explicit values: matrices whose entries do not repeat (an unstructured mesh;
 * the reference takes any CSR, mpi_kron.py:135-150) have no dictionary, but their
 * rows still share COLUMNS.  vals[((u*K + s)*2 + j)*n_mats + m] is the entry of
 * matrix m in row j of slot row u at the column of slot s (zero where that row
 * has none): 36 bytes per slot for two matrices instead of 20 in the one-row
 * plain form (stk_kron_ell_apply), 10 gathers for two rows instead of 14, every
 * row's space-factor sum accumulated in the same order.  A call on such a pattern
 * must use ALL its matrices, term k naming matrix k (n_terms = n_mats): the
 * caller keeps one value array per combination of matrices it applies. */
/* Row pairs (rows_per_unit = 2): slot row u serves the matrix rows
 * row_ids[2u] and row_ids[2u + 1] (-1: none); its K slots (K one of 8, 10, 12)
 * list the UNION of their columns in ascending order, and matrix m has the
 * values dict[(m*n_codes + code)*2 + j] for row j of the pair (zero where that
 * row has no entry in the column).  Two neighbouring vertices of a P1
 * triangulation share 4 of their 7 columns, so a pair costs 10 gathers
 * instead of 14; every row's sum is still accumulated in ascending column
 * order, so the results are bit-identical with rows_per_unit = 1. */

typedef struct {
    const double *tri; /* as in stk_kron_term */
    int32_t mat;       /* which matrix of the pattern */
} stk_kron_pack_term;

/* Host helper of the planners (no device work): groups rows that follow each
 * other in the processing order into units of up to rp rows whose union of
 * columns has at most K_out entries (greedily, left to right).  cols / codes:
 * [M][K], the real entries of position p in the first counts[p] slots, columns
 * ascending; own[p]: matrix row at position p.  Outputs sized for M units:
 * ucols [units][K_out], ucodes [units][K_out][rp] (zero_code = "no entry"),
 * urows [units][rp] (-1 = none).  stk_pack_unit_slots: K_out of the instantiated
 * kernels for rows of K slots (0: none). */
int32_t stk_pack_unit_slots(int32_t K, int32_t rp);
/* Host helper: a processing order in which rows that can share a slot row follow
 * each other, for patterns whose given order lacks that (unstructured meshes).
 * Behind every not yet placed row of the given order goes ONE not yet placed
 * neighbour (a column of the row) whose union of columns with it fits K_out
 * slots -- the nearest in the given order, at most `window` positions away.
 * counts / cols / own as in stk_pack_group_rows; perm[q] = position in the given
 * order of the row at position q of the new order. */
int stk_pack_match_order(int32_t M, int32_t K, const int32_t *counts,
                         const int32_t *cols, const int32_t *own, int32_t K_out,
                         int32_t window, int32_t *perm);
int stk_pack_group_rows(int32_t M, int32_t K, const int32_t *counts,
                        const int32_t *cols, const int32_t *codes,
                        const int32_t *own, int32_t zero_code, int32_t rp,
                        int32_t K_out, int32_t *n_units, int32_t *ucols,
                        int32_t *ucodes, int32_t *urows);

int stk_kron_pack_apply(void *stream, const stk_pack_pattern *pattern_host,
                        int32_t n_loc, int32_t ld, int32_t n_terms,
                        const stk_kron_pack_term *terms_host, const double *x,
                        const double *ghosts, double beta, double *y);

/* The first and last local time step after stk_kron_pack_apply ran with ghosts =
 * NULL and beta = 0 while the halo exchange was in flight (the reference overlaps
 * the exchange with the rows that do not need it, mpi_kron.py:193-200,
 * mpi_vector.py:177-179).  Both steps are recomputed from x and the received rows
 * in the one-pass kernel's order of operations and OVERWRITE y[i][0] and
 * y[i][n_loc-1]: pass + this call, the one-pass form with `ghosts`, and the
 * one-rank apply agree bit for bit (tests: torch.equal).  x: the slab the pass
 * read; x_lo / x_hi: the received rows as they arrive (contiguous, length M; NULL
 * on a side without a neighbour).  One lane per slot row and side on the packed
 * stream. */
int stk_kron_pack_ghost_apply(void *stream, const stk_pack_pattern *pattern_host,
                              int32_t n_loc, int32_t ld, int32_t n_terms,
                              const stk_kron_pack_term *terms_host, const double *x,
                              const double *x_lo, const double *x_hi, double *y);

/* The same two steps from COMPACT operands, one lane per slot row for both sides:
 * records[4*j .. 4*j+3] = (x[j][0], x[j][1], x[j][n_loc-2], x[j][n_loc-1]) as stk_halo_pack_records
 * leaves them (32-byte aligned), ghosts[2*j], ghosts[2*j+1] = (x_lo[j], x_hi[j]) as
 * stk_interleave_ghosts does (a side without a neighbour: zeros, and has_lo / has_hi = 0:
 * its step is not rewritten).  Three 16-byte loads per slot from two lines where
 * stk_kron_pack_ghost_apply gathers six times 8 bytes from the slab: 1.5-2 x faster, the
 * same doubles (tests: torch.equal).  Pass with ghosts = NULL and beta = 0 first, as above. */
int stk_kron_pack_boundary_apply(void *stream, const stk_pack_pattern *pattern_host,
                                 int32_t n_loc, int32_t ld, int32_t n_terms,
                                 const stk_kron_pack_term *terms_host,
                                 const double *records, const double *ghosts,
                                 int32_t has_lo, int32_t has_hi, double *y);

/* The same packed stream with an input slab PER TERM: y = beta*y + sum_k (T_k kron
 * X_k) xs[k] -- the last stage of the Schur complement S = B^T K B + G of
 * heateq_mpi.py:166-181 once K is applied to two right-hand sides instead of four
 * times: (I kron M_x) v1 + (I kron A_x) v2 + (G_t kron M_x) x, i.e. three
 * TridiagKronMatMPI / IdentityKronMatMPI applies and the sum of SumMPI._matvec
 * (mpi_kron.py:77-90, 135-150, 214-219) in one pass.  Every slot row gets a LANE
 * GROUP PER TERM (csrc/kron_pack_multi.hip): a lane gathers K columns of ITS
 * term's slab for one pair of time steps, the sums of all lanes meet in LDS, and
 * the first lane group applies the time factors and adds the terms in term order.
 * A lane whose term's time factor never multiplies its pair of time steps does
 * not gather (G_t has the single entry (0, 0)).  No ghost time steps: a term whose
 * factor couples to the neighbour ranks' rows takes stk_kron_pack_apply.  The
 * pattern must have a dictionary; xs_host: n_terms device pointers (host array),
 * 2 or 3 terms.  Every row's sums are those of stk_kron_pack_apply, term by term,
 * bit for bit.  (Slot rows that would need more than 512 lanes, and tuning key
 * "pack_multi_lanes" = 0: the terms take turns in one lane, round 4's form.) */
int stk_kron_pack_apply_multi(void *stream, const stk_pack_pattern *pattern_host,
                              int32_t n_loc, int32_t ld, int32_t n_terms,
                              const stk_kron_pack_term *terms_host,
                              const double *const *xs_host, double beta, double *y);
/* ... with the caller's knowledge of the time factors (host arrays of n_terms
 * entries, or both NULL): term k's factor has no non-zero entry in a COLUMN
 * outside the local time steps [t_begin_host[k], t_end_host[k]) -- it reads
 * xs[k] at those steps only -- so term k gets lanes for those pairs of steps only
 * and a workgroup holds more slot rows of the others (G_t: [0, 1); an identity or
 * a full tridiagonal factor: [0, n_loc)).  Term 0's lanes also write y and always
 * cover every step.  A range that leaves out a column the factor does reach gives
 * a wrong result: the library cannot see the factors' entries from the host. */
int stk_kron_pack_apply_multi_steps(void *stream, const stk_pack_pattern *pattern_host,
                                    int32_t n_loc, int32_t ld, int32_t n_terms,
                                    const stk_kron_pack_term *terms_host,
                                    const double *const *xs_host,
                                    const int32_t *t_begin_host,
                                    const int32_t *t_end_host, double beta, double *y);

/* ---- plan construction from CSR (no Python needed) ---------------------------
 * Everything the three forms above stream is derived here, on the host side of
 * the library, from the CSR matrices a caller of the reference holds
 * (mpi_shared_mem.py:46-48; TridiagKronMatMPI keeps references to them,
 * mpi_kron.py:209-210): union pattern of the n_mats matrices, K = the smallest
 * instantiated slot count that holds the longest row (overflow CSR beyond 16),
 * rows listed in `row_order_host` (a permutation of 0..M-1, e.g. a mesh-tile or
 * RCM order; NULL = index order), the dictionary of distinct value tuples and
 * the packed slot words when the tuples fit.  All arrays are HOST pointers;
 * the plan owns its device copies.  stk_kron_plan_apply runs
 *   y = beta*y + sum_k (T_k kron X_{t[k].mat}) x
 * on the fastest form the plan has; x_lo / x_hi are the ghost time rows
 * (device, length M, or NULL) and ghost_work 2*M device doubles of scratch,
 * needed when either is given. */
typedef struct stk_kron_plan stk_kron_plan;
int stk_kron_plan_create(int32_t M, int32_t n_mats,
                         const int32_t *const *indptr_host,
                         const int32_t *const *indices_host,
                         const double *const *data_host,
                         const int32_t *row_order_host, stk_kron_plan **out);
int stk_kron_plan_destroy(stk_kron_plan *plan);
/* Any output may be NULL.  K: slots per row of the sliced-ELL copy; packed = 1
 * if the dictionary form was built; rows_per_unit = 2 if a row-pair form was
 * built as well -- stk_kron_plan_apply uses it for slabs of 8 time steps and
 * more, the one-row form below (tuning key "pack_rows" = 1: no pairs in plans
 * created afterwards).  packed = 0 with rows_per_unit = 2: matrices without a
 * dictionary whose rows still share columns -- pairs with explicit values
 * (stk_pack_pattern.vals), the plain sliced-ELL form below 24 time steps. */
int stk_kron_plan_info(const stk_kron_plan *plan, int32_t *K, int32_t *n_codes,
                       int32_t *packed, int64_t *nnz_union,
                       int32_t *rows_per_unit);
int stk_kron_plan_apply(stk_kron_plan *plan, void *stream, int32_t n_loc,
                        int32_t ld, int32_t n_terms,
                        const stk_kron_pack_term *terms_host, const double *x,
                        const double *x_lo, const double *x_hi,
                        double *ghost_work, double beta, double *y);

/* After stk_kron_plan_apply ran with x_lo = x_hi = NULL and beta = 0 while the halo
 * exchange was in flight (mpi_kron.py:193-200): the first / last local time step
 * recomputed with the two received rows, as stk_kron_pack_ghost_apply /
 * stk_kron_ell_ghost_apply do.  x: the slab the apply read. */
int stk_kron_plan_ghost_apply(stk_kron_plan *plan, void *stream, int32_t n_loc,
                              int32_t ld, int32_t n_terms,
                              const stk_kron_pack_term *terms_host, const double *x,
                              const double *x_lo, const double *x_hi, double *y);

/* The same two steps from the compact records stk_halo_pack_records left (4*M doubles) and
 * the received rows (interleaved into ghost_work, 2*M doubles): stk_kron_pack_boundary_apply
 * on the plan's packed form, 2-3 x faster than the slab form above and the same doubles.
 * records = NULL, or a plan without a packed stream: the slab form (needs x). */
int stk_kron_plan_boundary_apply(stk_kron_plan *plan, void *stream, int32_t n_loc,
                                 int32_t ld, int32_t n_terms,
                                 const stk_kron_pack_term *terms_host, const double *x,
                                 const double *records, const double *x_lo,
                                 const double *x_hi, double *ghost_work, double *y);

/* Diagnostic: while `buf` (device, at least 8 * grid * 4 words) is non-NULL,
 * the headline instantiation of the one-row form (2 terms, K = 7, no ghosts;
 * tuning key "pack_rows" = 1) runs a stamped build
 * that leaves, per wavefront, the shader-clock cycles spent in the four
 * segments of a row-group iteration.  NULL switches it off. */
int stk_kron_pack_set_diag(unsigned long long *buf);

/* ghosts[2*j] = lo[j], ghosts[2*j + 1] = hi[j] (a NULL side is written as
 * zero): brings the two received time rows into the layout above. */
int stk_interleave_ghosts(void *stream, int32_t M, const double *lo,
                          const double *hi, double *ghosts);

/* ---- (I_t kron A) for a general, possibly rectangular CSR A ---------------
 * y = alpha * A x + beta * z  on `rows` x n_loc outputs (IdentityKronMatMPI,
 * mpi_kron.py:143-150; residual / restriction / prolongation steps of
 * MultiGrid.MGM, multigrid.py:174-180).  The entries of A may depend on the
 * time slice: a(t) = ca * vals_a + cm[t] * vals_m (vals_m, cm may be NULL).
 * z may be NULL (treated as 0) or alias y. */
int stk_csr_spmm(void *stream, int32_t rows, int32_t n_loc, int32_t ld,
                 const int32_t *indptr, const int32_t *indices,
                 const double *vals_a, double ca, const double *vals_m,
                 const double *cm, const double *x, double alpha, double beta,
                 const double *z, double *y);

/* The same on a sliced-ELL copy (fast path; slot count K one of 2, 4, 5, 6, 7,
 * 9, 12, 16, 20; padding slots: any valid column, value 0).  ELL row `pos`
 * produces output row row_ids[pos]; dia_a / dia_m hold the diagonal entry of
 * that row (Gauss-Seidel only).  diag_free (Gauss-Seidel copies): the slots hold
 * the OFF-diagonal entries only and a row is updated as
 *     u_i = (f_i - sum_{j != i} a_ij u_j) / a_ii
 * -- the form of PETSc's MatSOR, which the reference calls (multigrid.py:116-127);
 * one gather less per row than u_i += (f_i - sum_j a_ij u_j) / a_ii with the
 * diagonal among the slots (diag_free = 0), equal up to rounding. */
typedef struct {
    int32_t n_pos;  /* ELL rows */
    int32_t n_rows; /* rows of the output slab */
    int32_t K;
    const int32_t *idx;     /* n_pos*K columns */
    const double *va, *vm;  /* n_pos*K values; vm may be NULL */
    const int32_t *row_ids; /* n_pos or NULL */
    const double *dia_a, *dia_m; /* n_pos each or NULL */
    int32_t diag_free;
} stk_ell_rows;

/* y = alpha * A(t) x + beta * z; x has x_rows rows, y and z have ell->n_rows.
 * Tuning key "rows_wg_per_cu". */
int stk_ell_spmm(void *stream, const stk_ell_rows *ell_host, int32_t n_loc,
                 int32_t ld, int32_t x_rows, double ca, const double *cm,
                 const double *x, double alpha, double beta, const double *z,
                 double *y);

/* ---- direct inverse of a space matrix (InvLinOp, linop.py:18-26; precond =
 *      'direct', heateq_mpi.py:155-157) ----------------------------------------
 * x = A^-1 b for all time steps of a slab from the factors Pr A Pc = L U of
 * scipy.sparse.linalg.splu (SuperLU), which stays on the host at set-up as in the
 * reference (`self.inv = splu(mat)`); what `self.inv.solve` does per apply -- row
 * permutation, two triangular solves, column permutation -- runs on the device,
 * level-scheduled: wide levels of the elimination tree one launch each, runs of
 * narrow levels inside one workgroup (csrc/sptrsv.hip).  L and U: CSR on the host
 * (sorted columns; L unit lower triangular, U upper with its diagonal), perm_r /
 * perm_c: SuperLU.perm_r / perm_c (NULL = identity).  The sums of a row have one
 * shape whatever the slab length: the result does not depend on the partition of
 * the time axis.  work: M * ld device doubles; b may be x. */
typedef struct stk_lu stk_lu;
int stk_lu_create(int32_t n, const int32_t *L_indptr, const int32_t *L_indices,
                  const double *L_data, const int32_t *U_indptr,
                  const int32_t *U_indices, const double *U_data,
                  const int32_t *perm_r, const int32_t *perm_c, stk_lu **out);
int stk_lu_destroy(stk_lu *lu);
/* Any output may be NULL: dependency levels of the two solves, kernel launches
 * per stk_lu_solve. */
int stk_lu_info(const stk_lu *lu, int32_t *levels_L, int32_t *levels_U,
                int32_t *launches);
int stk_lu_solve(stk_lu *lu, void *stream, int32_t n_loc, int32_t ld,
                 const double *b, double *x, double *work);
/* The top of the elimination tree as a dense block (optional).  The narrow levels are
 * the separators near the root: a few thousand rows, one dependent row after the other
 * (770 of the 802 levels of A_x at J_space = 6 hold 2 578 of its 16 129 rows).  The plan
 * names them -- S = the rows from the first narrow level of L on, ascending
 * (stk_lu_top_rows; n_top = 0: no such block, or U's rows of S reach outside S) -- and a
 * caller that hands over the two blocks L[S, S] and U[S, S] as dense matrices (n_top x
 * n_top, row-major, DEVICE; the plan copies them) whose DIAGONAL BLOCKS of `block` rows
 * are replaced by their inverses turns every solve into
 *   head levels, d_S = (Pr b)_S - L[S, H] y_H, per block k ascending: y_k = inv(L_kk) (d_k - L[k, <k] y_<k);
 *   per block k descending: z_k = inv(U_kk) (y_k - U[k, >k] z_>k), head levels
 * with one launch per block.  block = n_top: the explicit inverses of the whole blocks
 * (two launches per solve, five to ten times the rounding error of substitution); 256
 * keeps the accuracy of substitution.  Without the call the plan walks every level (a
 * run of narrow levels in one workgroup). */
int stk_lu_top_rows(const stk_lu *lu, int32_t *n_top, int32_t *rows_host);
int stk_lu_set_top_inverse(stk_lu *lu, const double *L_blocks_dev,
                           const double *U_blocks_dev, int32_t block);

/* ---- (A_t kron I) for a small sparse time matrix ---------------------------
 * y[., t] = (add_identity ? x[., t] : 0) + sum_e val[e] * src(col[e]) over the
 * CSR row t of the LOCAL rows of the time matrix; col < n_loc addresses the
 * local slab, col >= n_loc addresses row (col - n_loc) of `recv`, a
 * (n_recv x M) array of time rows fetched from other ranks
 * (SparseKronIdentityMPI._matvec, mpi_kron.py:285-317;
 * TridiagKronIdentityMPI._matvec, :186-201). */
int stk_time_csr_apply(void *stream, int32_t M, int32_t n_loc, int32_t ld,
                       const int32_t *t_indptr, const int32_t *t_cols,
                       const double *t_vals, const double *x,
                       const double *recv, int32_t add_identity, double *y);

/* ---- (T kron I) for a dense, possibly rectangular time factor --------------
 * y[i, r] = sum_c T[r*n_in + c] * x[i*ld_in + c] for r < n_out; columns
 * n_out..ld_out-1 of y are set to zero.  The time side of the serial KronLinOp
 * (linop.py:6-15: mat_time.dot(X)) when the factor is not square -- the trial
 * and test spaces in time of heateq.py:37-51 differ -- or is given as a
 * LinearOperator; x and y are slabs with their own row strides. */
int stk_time_dense_apply(void *stream, int32_t M, int32_t n_in, int32_t ld_in,
                         int32_t n_out, int32_t ld_out, const double *T,
                         const double *x, double *y);

/* ---- wavelet transform in time, whole time axis on this GPU ---------------
 * y = (W_t kron I) x or (W_t^T kron I) x in the interleaved numbering, all
 * J levels in one pass (WaveletTransformOp._matmat / _rmatmat,
 * wavelets.py:106-134; same operator as the composite of
 * wavelets.py:172-198 on one rank).  Requires n_loc == 2^J + 1. */
int stk_wavelet_apply(void *stream, int32_t M, int32_t J, int32_t ld,
                      int32_t transposed, const double *x, double *y);

/* ---- multigrid V-cycles with Gauss-Seidel smoothing ------------------------
 * MultiGrid._matvec / MGM (multigrid.py:168-193), PETScSMoother / Smoother
 * (multigrid.py:83-127), batched over the n_loc time slices. */
typedef struct {
    int32_t n; /* dofs on this level */
    const int32_t *indptr, *indices;
    const double *vals_a, *vals_m; /* a(t) = ca*vals_a + cm[t]*vals_m */
    const int32_t *diag;           /* CSR position of the diagonal of row i */
    /* Gauss-Seidel dependency schedule: rows grouped into DAG levels; rows of
     * one group are mutually independent.  *_ptr_host has n_groups+1 entries
     * (host memory), *_rows is a device array of row indices. */
    int32_t n_fwd;
    const int32_t *fwd_ptr_host;
    const int32_t *fwd_rows;
    int32_t n_bwd;
    const int32_t *bwd_ptr_host;
    const int32_t *bwd_rows;
    /* prolongation from the next coarser level (n x n_coarse) and its
     * transpose; NULL on level 0 */
    const int32_t *p_indptr, *p_indices;
    const double *p_vals;
    const int32_t *r_indptr, *r_indices;
    const double *r_vals;
    /* Optional sliced-ELL copies (host structs, copied by stk_mg_create); when
     * present the V-cycle runs on them.  ell_a: the level matrix in a
     * locality order, for the residual.  ell_fwd / ell_bwd: the level matrix
     * with its rows listed group by group of the forward / backward
     * Gauss-Seidel schedule; *_pos_host[g] .. [g+1] is the position range of
     * group g (n_fwd+1 / n_bwd+1 entries).  ell_p / ell_r: the transfers.
     * ell_ra (optional): the product R * (level matrix) on the coarse rows
     * (va = R*vals_a, vm = R*vals_m): the restricted residual is then formed
     * as (R A) u - R f without writing the fine residual. */
    const stk_ell_rows *ell_a, *ell_fwd, *ell_bwd, *ell_p, *ell_r;
    const int32_t *fwd_pos_host, *bwd_pos_host;
    const stk_ell_rows *ell_ra;
    /* ell_fwd0 (optional): n_fwd matrices, one per group of the forward
     * schedule, for the first sweep of a level visit, which starts from u = 0:
     * group g keeps only the entries whose column lies in a group < g (the
     * other products are exact zeros), unused slots point at a row of group 0.
     * With it the plan neither zeroes u before that sweep nor gathers the
     * zeros. */
    const stk_ell_rows *ell_fwd0;
    /* Optional (0 / NULL if absent): the rows of ell_fwd / ell_bwd are listed in
     * an order that follows the geometry; *_tile_row_host[pos] is the index of
     * the tile row (0 .. n_tile_rows-1, ascending inside every group) position
     * pos lies in, and rows in tile row r only couple to tile rows r-1 .. r+1.
     * With it a sweep on a large level runs strip by strip -- all groups on one
     * strip of tile rows before the next strip, each group shifted by one tile
     * row against the previous one so that the order of updates is unchanged --
     * and the strip's vectors stay in the Infinity Cache between the group
     * passes. */
    int32_t n_tile_rows;
    const int32_t *fwd_tile_row_host, *bwd_tile_row_host;
    /* Optional (NULL if absent): ell_fwd / ell_bwd once more in the OTHER row form
     * (stk_ell_rows.diag_free), same rows in the same order: what the sweeps the
     * plan options "fast_until_cycle" / "fast_parts" name run on. */
    const stk_ell_rows *ell_fwd_alt, *ell_bwd_alt;
} stk_mg_level;

typedef struct stk_mg stk_mg;

/* coarse_inv: n_kinds dense (n0 x n0) inverses of the level-0 matrix, row
 * major, one per distinct time-slice coefficient (device).  max_ld bounds the
 * ld of later applies (workspace is allocated here). */
int stk_mg_create(int32_t n_levels, const stk_mg_level *levels_host,
                  int32_t smoothsteps, int32_t vcycles, int32_t n_kinds,
                  const double *coarse_inv, int32_t max_ld, stk_mg **out);
int stk_mg_destroy(stk_mg *mg);
/* Per-plan choices that regroup the reference's arithmetic (results change in
 * the last bits only; the solve's r.Pr history is sensitive to them, DESIGN.md
 * section 5).  "fuse_restrict": 1 = the restricted residual as (R A) u - R f from
 * the precomputed product R A (no fine-level residual is written), 0 = the
 * reference's R (A u - f) (multigrid.py:174-175), -1 = follow the process-wide
 * tuning key "mg_fuse_restrict" (default 1).  "strip_pct": the strips of the
 * strip-wise smoothing of THIS plan as a percentage of the tuning key
 * "mg_strip_mb" (results do not depend on it): plans whose applies run two at a
 * time share the caches and want smaller strips, a plan that runs alone larger
 * ones.
 * Where the reference's forms are used (the history's gap to the CPU path is owned
 * by the LAST V-cycle's restricted residual and post-smoothing on the FINEST level:
 * DESIGN.md section 5): "fuse_restrict_min_level" / "_max_level" (default 0 / no
 * bound): the fused form only on levels inside the window, the two steps outside;
 * "fast_until_cycle" (default 0): the V-cycles with a smaller index take the fast
 * forms throughout -- the alternative sweep copies of levels that have them
 * (stk_mg_level.ell_fwd_alt / ell_bwd_alt), the fused restricted residual on every
 * level; "fast_parts": in the later V-cycles bit 0 lets the pre-smoothing and bit 1
 * the restricted residual take the fast forms as well (the post-smoothing never
 * does). */
int stk_mg_set_option(stk_mg *plan, const char *key, int32_t value);
/* u = MG(f): `vcycles` V-cycles from u = 0.  cm/kind: per-time-slice mass
 * coefficient and coarse-inverse index (device, n_loc) or NULL. */
int stk_mg_apply(stk_mg *mg, void *stream, int32_t n_loc, int32_t ld,
                 double ca, const double *cm, const int32_t *kind,
                 const double *f, double *u);
/* Member matrices for the coarse end of a plan with a second matrix (cm != NULL):
 * by default time slice t runs on ca * A_l + cm[t] * M_l on every level l, the two
 * Galerkin chains combined per slice.  The reference builds one hierarchy per
 * wavelet level from the ASSEMBLED matrix 2^j M + alpha A (heateq_mpi.py:97-98,
 * 147-153; Galerkin products multigrid.py:142-145): its coarse matrices carry the
 * rounding of that one chain, and the difference to the combination grows fourfold
 * per level down (1e-16 at the top, 1e-12 on level 0 of a ten-level hierarchy) --
 * on the smallest levels it owns most of the gap between the two r.Pr histories
 * (profiles/r06_history_by_coarse_level_J7_J10.json).  With member matrices the
 * levels 1 .. stk_mg_coarse_levels(plan) -- the levels of the fused coarse kernel --
 * read, for slice t, the entries of matrix kind[t] itself (kind as in stk_mg_apply:
 * the index of the slice's coarse inverse).  One call per level with the n_kinds
 * matrices of that level as CSR on the host (sorted columns; a NULL indptr for a
 * kind no slice names); members take effect once every level 1..coarse_levels has
 * them.  Entries outside the plan's pattern are ignored. */
int stk_mg_coarse_levels(const stk_mg *plan);
int stk_mg_set_member_matrices(stk_mg *plan, int32_t level, int32_t n_kinds,
                               const int32_t *const *indptr_host,
                               const int32_t *const *indices_host,
                               const double *const *data_host);
/* `its` forward (backward = 0) or backward Gauss-Seidel sweeps on one level
 * (exposed for tests; multigrid.py:116-127). */
int stk_mg_smooth(stk_mg *mg, void *stream, int32_t level, int32_t n_loc,
                  int32_t ld, double ca, const double *cm, int32_t its,
                  int32_t backward, const double *f, double *u);

/* The same plan built inside the library from what a caller of the reference
 * holds when it constructs MultiGrid(mat, hierarchy) (multigrid.py:130-166): the
 * finest-level matrix A (and optionally a second matrix M on the same rows, for
 * families a(t) = ca*A + cm[t]*M, heateq_mpi.py:97-98) and the n_levels - 1
 * prolongation matrices, P[j] from level j to level j + 1 (multigrid.py:39-60),
 * all as HOST CSR.  Computes the Galerkin products R A P (multigrid.py:142-145;
 * same accumulation order as SciPy's csr_matmat, rounding-noise entries dropped),
 * the Gauss-Seidel dependency schedules, every sliced-ELL copy, the bands of the
 * strip-wise smoothing (verified on the pattern) and the exact inverses of the
 * coarsest matrices: kind 0 = A_0, kind 1 + k = ca*A_0 + cms[k]*M_0.
 * coords (optional, HOST): coordinates of the finest-level dofs, row-major
 * (n x dim), level l = the first n_l of them -- only used for cache-friendly row
 * orders and bands; NULL = index order / breadth-first bands.
 * The Gauss-Seidel copies follow the tuning key "mg_gs_diag_free" AS IT STANDS WHEN
 * THE PLAN IS BUILT: 1 (default) = diagonal-free rows, u_i = (f_i - sum_{j != i}
 * a_ij u_j) / a_ii; 0 = the diagonal among the slots and the reference's update
 * u_i += (f_i - row_i u) / a_ii (multigrid.py:89-97); 2 = the latter on the finest
 * level, which also gets the diagonal-free copies as its alternative form, the
 * former below.  Key 2 together with stk_mg_set_option(plan,
 * "fuse_restrict_max_level", n_levels - 2), ("fast_until_cycle", vcycles - 1) and
 * ("fast_parts", 1) is the arithmetic the mirrored driver runs by default (r.Pr
 * histories within 1e-10 of the CPU path for 4 % of the solve); key 0 with
 * ("fuse_restrict", 0) has the reference's forms everywhere.
 * With coordinates, the bands of the strip-wise sweeps hold several mesh rows
 * (tuning key "mg_band_merge" as it stands when the plan is built; 0 = the default:
 * 6 for a family, 1 for a single matrix): inside a thicker band a stage walks mesh
 * tiles instead of the level's full width (a family's P apply 2 % faster on slabs of
 * 17 time steps and more, bit-identical).
 * The host work runs on the threads of the library (STK_HOST_THREADS overrides
 * their number); STK_PLAN_TIMING=1 prints the seconds per stage on stderr. */
typedef struct {
    int32_t n_rows, n_cols;
    const int32_t *indptr, *indices;
    const double *data;
} stk_csr_host;

/* The exact solve on level 0 (multigrid.py:161-165, 169-171) is a dense inverse
 * here.  By default the library inverts the coarsest matrices itself (Gauss-Jordan
 * with partial pivoting); a caller whose other code path inverts them with its own
 * factorisation -- the Python planner of this repository takes numpy.linalg.inv,
 * i.e. LAPACK -- hands that routine over, and plans created from then on carry ITS
 * inverse: the two planners' V-cycles are then equal bit for bit, not to 1e-13.
 * fn(n, a, inv, user): a and inv are n x n, row-major, HOST; returns 0 on success.
 * fn = NULL restores the built-in inverse.  Called on the thread that calls
 * stk_mg_create_from_csr.  The hook and the tuning key "mg_band_merge" are
 * process-wide; a plan construction reads both once, under a mutex, when it starts,
 * so a setter on another thread takes effect for constructions that start later and
 * never half way through one.  A caller that wants a hook for ONE construction while
 * other threads build plans must serialise those constructions itself. */
typedef int (*stk_dense_inverse_fn)(int32_t n, const double *a, double *inv,
                                    void *user);
int stk_mg_set_coarse_inverse(stk_dense_inverse_fn fn, void *user);

int stk_mg_create_from_csr(int32_t n_levels, const stk_csr_host *A_fine,
                           const stk_csr_host *M_fine,
                           const stk_csr_host *P_host,
                           const double *coords_host, int32_t dim,
                           int32_t smoothsteps, int32_t vcycles, double ca,
                           int32_t n_cms, const double *cms_host,
                           int32_t max_ld, stk_mg **out);

/* ---- problem set-up: P1 assembly on the host threads of the library ----------
 * Mass M and stiffness A of a triangulation, restricted to its free dofs, explicit
 * zeros dropped: what the reference gets from NGSolve's BilForm(...).assemble()
 * and source/ngsolve_helper.py:38-45 (heateq_mpi.py:91-96).  HOST arrays:
 * points [nv][2], cells [nc][3] (vertex numbers), boundary [nv] (1 = Dirichlet
 * vertex).  Every row sums the contributions of the triangles around its vertex in
 * ascending triangle number, without fused multiply-adds: the result depends on
 * the mesh alone, not on the number of threads.  Stiffness entries below zero_rel
 * times the largest one are rounding noise of exact zeros and are dropped.
 * The result is read with stk_p1_result_sizes / _copy (which = 0: A, 1: M; CSR with
 * int32 indices, as mpi_shared_mem.py:46-48) and released with stk_p1_result_free. */
typedef struct stk_p1_result stk_p1_result;
int stk_p1_assemble_2d(int64_t nv, int64_t nc, const double *points,
                       const int64_t *cells, const uint8_t *boundary,
                       double zero_rel, stk_p1_result **out);
int stk_p1_result_sizes(const stk_p1_result *r, int32_t *n_free, int64_t *nnz_a,
                        int64_t *nnz_m);
int stk_p1_result_copy(const stk_p1_result *r, int32_t which, int32_t *indptr,
                       int32_t *indices, double *data);
int stk_p1_result_free(stk_p1_result *r);

/* ---- problem set-up: one uniform refinement of a triangulation ---------------
 * What the reference takes from Netgen (source/problem.py:7-41: mesh.Refine())
 * together with NGSolve's parent-vertex table (source/multigrid.py:20-21,
 * GetParentVertices): every triangle cut into four, the midpoints appended to the
 * vertices in the order (colour of the bisected edge, y, x) -- the hierarchical
 * numbering the prolongations and the dof-order sweeps are built on.  HOST arrays:
 * points [nv][2], tris [nt][3], tri_colors [nt][3] (colour of the edge opposite
 * each local vertex; the two triangles of an edge must agree).  Written: the
 * n_new <= capacity new vertices (new_points [.][2], new_parents [.][2] = the ends
 * of the bisected edge, smaller number first, new_colors [.]), and the 4 nt children
 * with their edge colours, child b of triangle t at row b * nt + t.  Integer work
 * and one halving per coordinate on the host threads of the library: the result
 * does not depend on their number. */
int stk_tri_refine(int64_t nv, int64_t nt, const double *points,
                   const int64_t *tris, const int64_t *tri_colors,
                   int64_t capacity, double *new_points, int64_t *new_parents,
                   int64_t *new_colors, int64_t *child_tris,
                   int64_t *child_colors, int64_t *n_new);

/* ---- problem set-up: load vector of a triangulation --------------------------
 * int f phi_i over the mesh with an nq-point rule given in barycentric
 * coordinates (rule_points [nq][3], rule_weights [nq], weights summing to 1):
 * what the reference gets from LinearForm(u0 * v * dx).assemble()
 * (heateq_mpi.py:102-103).  Two calls around the CALLER's evaluation of f:
 * stk_p1_load_points_2d writes the coordinates of the quadrature points
 * (qx, qy [nt][nq]: l0 p0 + l1 p1 + l2 p2, summed from the left);
 * stk_p1_load_sum_2d takes f [nt][nq] at those points and writes vec [nv] (ALL
 * vertices; the caller keeps the free ones): the share of triangle t in the entry
 * of its local vertex a is (sum_q (f_q w_q) l_qa) * |T|, q ascending, and an entry
 * sums its shares in ascending (t, a) -- no fused multiply-adds, the result depends
 * on the mesh alone, not on the number of host threads. */
int stk_p1_load_points_2d(int64_t nv, int64_t nt, const double *points,
                          const int64_t *tris, int32_t nq,
                          const double *rule_points, double *qx, double *qy);
int stk_p1_load_sum_2d(int64_t nv, int64_t nt, const double *points,
                       const int64_t *tris, int32_t nq,
                       const double *rule_weights, const double *rule_points,
                       const double *f, double *vec);

/* ---- plan construction on the host threads: processing order and union pattern ---
 * stk_tile_order: the mesh-tile order of n dofs with coordinates coords [n][d]
 * (d = 2 or 3): the bounding box cut into cubes of edge `side` from the corner `lo`
 * (the caller's choice: source/assembly.py takes about 2 048 dofs per tile), tiles
 * visited lexicographically with the last axis slowest and the dofs of a tile
 * likewise; ties by index.  A performance hint for the gather kernels
 * (stk_kron_*_apply's row_ids): results never depend on it.
 * stk_csr_union_count / _fill: one CSR pattern holding the patterns of n_mats
 * matrices of n rows (each with strictly ascending columns per row), and every
 * matrix's values on it, zeros where it has no entry -- what SumMPI's terms
 * (mpi_kron.py:186-200) become for a fused pass.  _count writes out_indptr [n + 1];
 * the caller allocates out_indices and out_data[k] of out_indptr[n] entries;
 * _fill writes them. */
int stk_tile_order(int64_t n, int32_t d, const double *coords, const double *lo,
                   double side, int32_t *order);
int stk_csr_union_count(int64_t n, int32_t n_mats, const int32_t *const *indptr,
                        const int32_t *const *indices, int32_t *out_indptr);
int stk_csr_union_fill(int64_t n, int32_t n_mats, const int32_t *const *indptr,
                       const int32_t *const *indices, const double *const *data,
                       const int32_t *out_indptr, int32_t *out_indices,
                       double *const *out_data);

#ifdef __cplusplus
}
#endif
#endif /* STK_H */

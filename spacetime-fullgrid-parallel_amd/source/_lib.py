"""ctypes binding of libstk.so (include/stk.h) and the torch plumbing around it.

PyTorch is used for device memory, streams and torch.distributed only; every
arithmetic kernel of the hot path is a hand-written HIP kernel in libstk.
There is NO CPU fallback: if the library is missing, or a tensor does not live
on a GPU, the call raises.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'libstk.so')

c_i32, c_i64, c_f64, c_p = (ctypes.c_int32, ctypes.c_int64, ctypes.c_double,
                            ctypes.c_void_p)


class KronTerm(ctypes.Structure):
    _fields_ = [('tri', c_p), ('vals', c_p), ('x', c_p), ('x_lo', c_p),
                ('x_hi', c_p)]


# callbacks of stk_pcg_solve (include/stk.h)
OPERATOR_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p,
                                ctypes.POINTER(ctypes.c_double), ctypes.c_int32)


class EllPattern(ctypes.Structure):
    _fields_ = [('M', c_i32), ('K', c_i32), ('ell_idx', c_p),
                ('row_ids', c_p), ('ovf_indptr', c_p), ('ovf_indices', c_p)]


class KronEllTerm(ctypes.Structure):
    _fields_ = [('tri', c_p), ('ell_vals', c_p), ('ovf_vals', c_p),
                ('x', c_p), ('x_lo', c_p), ('x_hi', c_p)]


class PackPattern(ctypes.Structure):
    _fields_ = [('M', c_i32), ('K', c_i32), ('col_bits', c_i32),
                ('n_codes', c_i32), ('n_mats', c_i32),
                ('rows_per_unit', c_i32), ('n_units', c_i32), ('slots', c_p),
                ('row_ids', c_p), ('dict', c_p), ('vals', c_p)]


class KronPackTerm(ctypes.Structure):
    _fields_ = [('tri', c_p), ('mat', c_i32)]


class CommMsg(ctypes.Structure):
    _fields_ = [('buf', c_p), ('count', c_i64), ('peer', c_i32)]


class EllRows(ctypes.Structure):
    _fields_ = [('n_pos', c_i32), ('n_rows', c_i32), ('K', c_i32),
                ('idx', c_p), ('va', c_p), ('vm', c_p), ('row_ids', c_p),
                ('dia_a', c_p), ('dia_m', c_p), ('diag_free', c_i32)]


class CsrHost(ctypes.Structure):
    _fields_ = [('n_rows', c_i32), ('n_cols', c_i32), ('indptr', c_p),
                ('indices', c_p), ('data', c_p)]


class MGLevel(ctypes.Structure):
    _fields_ = [('n', c_i32), ('indptr', c_p), ('indices', c_p),
                ('vals_a', c_p), ('vals_m', c_p), ('diag', c_p),
                ('n_fwd', c_i32), ('fwd_ptr_host', c_p), ('fwd_rows', c_p),
                ('n_bwd', c_i32), ('bwd_ptr_host', c_p), ('bwd_rows', c_p),
                ('p_indptr', c_p), ('p_indices', c_p), ('p_vals', c_p),
                ('r_indptr', c_p), ('r_indices', c_p), ('r_vals', c_p),
                ('ell_a', ctypes.POINTER(EllRows)),
                ('ell_fwd', ctypes.POINTER(EllRows)),
                ('ell_bwd', ctypes.POINTER(EllRows)),
                ('ell_p', ctypes.POINTER(EllRows)),
                ('ell_r', ctypes.POINTER(EllRows)),
                ('fwd_pos_host', c_p), ('bwd_pos_host', c_p),
                ('ell_ra', ctypes.POINTER(EllRows)),
                ('ell_fwd0', ctypes.POINTER(EllRows)),
                ('n_tile_rows', c_i32), ('fwd_tile_row_host', c_p),
                ('bwd_tile_row_host', c_p),
                ('ell_fwd_alt', ctypes.POINTER(EllRows)),
                ('ell_bwd_alt', ctypes.POINTER(EllRows))]


_PROTOTYPES = {
    'stk_last_error': (ctypes.c_char_p, []),
    'stk_version': (ctypes.c_int, []),
    'stk_device_info': (ctypes.c_int, [c_p, c_p, c_p]),
    'stk_axpby': (ctypes.c_int, [c_p, c_i64, c_f64, c_p, c_f64, c_p]),
    'stk_axpbyz': (ctypes.c_int, [c_p, c_i64, c_f64, c_p, c_f64, c_p, c_p]),
    'stk_dot_work_size': (c_i64, []),
    'stk_dot': (ctypes.c_int, [c_p, c_i64, c_p, c_p, c_p, c_p]),
    'stk_set_tuning': (ctypes.c_int, [ctypes.c_char_p, c_i32]),
    'stk_partition': (ctypes.c_int, [c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'stk_pcg_work_size': (c_i64, [c_i64]),
    'stk_pcg_solve': (ctypes.c_int, [
        c_p, c_i64, OPERATOR_FN, c_p, OPERATOR_FN, c_p, ALLREDUCE_FN, c_p, c_p,
        c_p, c_f64, c_i32, c_p, c_p, c_p
    ]),
    'stk_pcg_slab_work_size': (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    'stk_pcg_solve_slab': (ctypes.c_int, [
        c_p, c_i32, c_i32, c_i32, c_i32, c_i32, OPERATOR_FN, c_p, OPERATOR_FN, c_p,
        ALLREDUCE_FN, c_p, c_p, c_p, c_f64, c_i32, c_p, c_p, c_p
    ]),
    'stk_slab_dot_work_size': (c_i64, [c_i32, c_i32]),
    'stk_slab_dot': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_i32, c_i32, c_p]),
    'stk_sum_steps': (c_f64, [c_p, c_i32]),
    'stk_slab_gather_columns': (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_i32]),
    'stk_slab_scatter_columns': (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_i32]),
    'stk_slab_ld': (ctypes.c_int, [c_i32]),
    'stk_slab_alloc': (ctypes.c_int, [c_i32, c_i32, c_p, c_p]),
    'stk_slab_free': (ctypes.c_int, [c_p]),
    'stk_slab_upload': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p]),
    'stk_slab_download': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p]),
    'stk_transpose': (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_i64, c_p, c_i64, c_i32]),
    'stk_halo_pack': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_i32]),
    'stk_slab_extract_time_rows': (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_i64]),
    'stk_copy_block': (ctypes.c_int, [c_p, c_i64, c_i32, c_p, c_i64, c_p, c_i64]),
    'stk_outer': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'stk_comm_unique_id': (ctypes.c_int, [c_p]),
    'stk_comm_create': (ctypes.c_int, [c_i32, c_i32, c_p, ctypes.POINTER(c_p)]),
    'stk_comm_destroy': (ctypes.c_int, [c_p]),
    'stk_comm_allreduce_sum': (ctypes.c_int, [c_p, c_p, c_p, c_i32]),
    'stk_comm_halo_exchange': (ctypes.c_int, [c_p, c_p, c_i32, c_p, c_p, c_p, c_p]),
    'stk_comm_exchange': (ctypes.c_int, [c_p, c_p, c_i32, ctypes.POINTER(CommMsg), c_i32,
                                         ctypes.POINTER(CommMsg)]),
    'stk_ell_from_csr': (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_p,
                                        c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'stk_gs_depth_step': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p]),
    'stk_csr_galerkin': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                        c_i32, c_p, c_p, c_p, c_p]),
    'stk_timing_enable': (ctypes.c_int, [c_i32]),
    'stk_timing_reset': (ctypes.c_int, []),
    'stk_timing_get': (ctypes.c_int, [ctypes.c_char_p, c_p, c_p]),
    'stk_lanczos_work_size': (c_i64, [c_i64]),
    'stk_lanczos': (ctypes.c_int, [
        c_p, c_i64, OPERATOR_FN, c_p, OPERATOR_FN, c_p, ALLREDUCE_FN, c_p, c_p,
        c_i32, c_f64, c_f64, c_p, c_p, c_p, c_p, c_p, c_p, c_p
    ]),
    'stk_mg_coarse_levels': (ctypes.c_int, [c_p]),
    'stk_mg_set_member_matrices': (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_p, c_p]),
    'stk_lu_create': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.POINTER(c_p)]),
    'stk_lu_destroy': (ctypes.c_int, [c_p]),
    'stk_lu_info': (ctypes.c_int, [c_p, c_p, c_p, c_p]),
    'stk_lu_top_rows': (ctypes.c_int, [c_p, c_p, c_p]),
    'stk_lu_set_top_inverse': (ctypes.c_int, [c_p, c_p, c_p, c_i32]),
    'stk_lu_solve': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_p, c_p, c_p]),
    'stk_lanczos_slab_work_size': (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    'stk_lanczos_slab': (ctypes.c_int, [
        c_p, c_i32, c_i32, c_i32, c_i32, c_i32, OPERATOR_FN, c_p, OPERATOR_FN, c_p,
        ALLREDUCE_FN, c_p, c_p, c_i32, c_f64, c_f64, c_p, c_p, c_p, c_p, c_p, c_p, c_p
    ]),
    'stk_kron_sum_apply': (ctypes.c_int, [
        c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_i32,
        ctypes.POINTER(KronTerm), c_f64, c_p
    ]),
    'stk_kron_ell_apply': (ctypes.c_int, [
        c_p, ctypes.POINTER(EllPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronEllTerm), c_f64, c_p
    ]),
    'stk_kron_ell_ghost_apply': (ctypes.c_int, [
        c_p, ctypes.POINTER(EllPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronEllTerm), c_p
    ]),
    'stk_pack_unit_slots': (c_i32, [c_i32, c_i32]),
    'stk_pack_match_order': (ctypes.c_int, [c_i32, c_i32, c_p, c_p, c_p, c_i32, c_i32, c_p]),
    'stk_pack_group_rows': (ctypes.c_int, [
        c_i32, c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p
    ]),
    'stk_kron_pack_apply': (ctypes.c_int, [
        c_p, ctypes.POINTER(PackPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronPackTerm), c_p, c_p, c_f64, c_p
    ]),
    'stk_kron_pack_ghost_apply': (ctypes.c_int, [
        c_p, ctypes.POINTER(PackPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronPackTerm), c_p, c_p, c_p, c_p
    ]),
    'stk_kron_plan_boundary_apply': (ctypes.c_int, [
        c_p, c_p, c_i32, c_i32, c_i32, ctypes.POINTER(KronPackTerm), c_p, c_p, c_p, c_p, c_p, c_p
    ]),
    'stk_kron_pack_boundary_apply': (ctypes.c_int, [
        c_p, ctypes.POINTER(PackPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronPackTerm), c_p, c_p, c_i32, c_i32, c_p
    ]),
    'stk_halo_pack_records': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_i32, c_p]),
    'stk_kron_pack_apply_multi': (ctypes.c_int, [
        c_p, ctypes.POINTER(PackPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronPackTerm), ctypes.POINTER(c_p), c_f64, c_p
    ]),
    'stk_kron_pack_apply_multi_steps': (ctypes.c_int, [
        c_p, ctypes.POINTER(PackPattern), c_i32, c_i32, c_i32,
        ctypes.POINTER(KronPackTerm), ctypes.POINTER(c_p), c_p, c_p, c_f64, c_p
    ]),
    'stk_mg_set_coarse_inverse': (ctypes.c_int, [c_p, c_p]),
    'stk_interleave_ghosts': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_p]),
    'stk_kron_plan_create': (ctypes.c_int, [
        c_i32, c_i32, ctypes.POINTER(c_p), ctypes.POINTER(c_p),
        ctypes.POINTER(c_p), c_p, ctypes.POINTER(c_p)
    ]),
    'stk_kron_plan_destroy': (ctypes.c_int, [c_p]),
    'stk_kron_plan_info': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p]),
    'stk_kron_plan_apply': (ctypes.c_int, [
        c_p, c_p, c_i32, c_i32, c_i32, ctypes.POINTER(KronPackTerm), c_p, c_p,
        c_p, c_p, c_f64, c_p
    ]),
    'stk_kron_plan_ghost_apply': (ctypes.c_int, [
        c_p, c_p, c_i32, c_i32, c_i32, ctypes.POINTER(KronPackTerm), c_p, c_p,
        c_p, c_p
    ]),
    'stk_kron_pack_set_diag': (ctypes.c_int, [c_p]),
    'stk_ell_spmm': (ctypes.c_int, [
        c_p, ctypes.POINTER(EllRows), c_i32, c_i32, c_i32, c_f64, c_p, c_p,
        c_f64, c_f64, c_p, c_p
    ]),
    'stk_csr_spmm': (ctypes.c_int, [
        c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_f64, c_p, c_p, c_p, c_f64,
        c_f64, c_p, c_p
    ]),
    'stk_time_csr_apply': (ctypes.c_int, [
        c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_p
    ]),
    'stk_time_dense_apply': (ctypes.c_int, [
        c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p
    ]),
    'stk_wavelet_apply': (ctypes.c_int,
                          [c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    'stk_mg_create': (ctypes.c_int, [
        c_i32,
        ctypes.POINTER(MGLevel), c_i32, c_i32, c_i32, c_p, c_i32,
        ctypes.POINTER(c_p)
    ]),
    'stk_mg_create_from_csr': (ctypes.c_int, [
        c_i32, ctypes.POINTER(CsrHost), ctypes.POINTER(CsrHost),
        ctypes.POINTER(CsrHost), c_p, c_i32, c_i32, c_i32, c_f64, c_i32, c_p,
        c_i32, ctypes.POINTER(c_p)
    ]),
    'stk_p1_assemble_2d': (ctypes.c_int, [c_i64, c_i64, c_p, c_p, c_p, c_f64, ctypes.POINTER(c_p)]),
    'stk_p1_result_sizes': (ctypes.c_int, [c_p, c_p, c_p, c_p]),
    'stk_p1_result_copy': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_p]),
    'stk_p1_result_free': (ctypes.c_int, [c_p]),
    'stk_p1_load_points_2d': (ctypes.c_int, [c_i64, c_i64, c_p, c_p, c_i32, c_p, c_p, c_p]),
    'stk_p1_load_sum_2d': (ctypes.c_int, [c_i64, c_i64, c_p, c_p, c_i32, c_p, c_p, c_p, c_p]),
    'stk_tile_order': (ctypes.c_int, [c_i64, c_i32, c_p, c_p, c_f64, c_p]),
    'stk_csr_union_count': (ctypes.c_int, [c_i64, c_i32, c_p, c_p, c_p]),
    'stk_csr_union_fill': (ctypes.c_int, [c_i64, c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'stk_tri_refine': (ctypes.c_int, [c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    'stk_mg_destroy': (ctypes.c_int, [c_p]),
    'stk_mg_set_option': (ctypes.c_int, [c_p, ctypes.c_char_p, c_i32]),
    'stk_mg_apply': (ctypes.c_int,
                     [c_p, c_p, c_i32, c_i32, c_f64, c_p, c_p, c_p, c_p]),
    'stk_mg_smooth': (ctypes.c_int, [
        c_p, c_p, c_i32, c_i32, c_i32, c_f64, c_p, c_i32, c_i32, c_p, c_p
    ]),
}

EXPORTED_SYMBOLS = sorted(_PROTOTYPES)

_lib = None


class StkError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StkError(
                'libstk.so not found at %s: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C spacetime-fullgrid-parallel_amd/csrc`. '
                'There is no CPU fallback.' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc):
    if rc != 0:
        raise StkError(lib().stk_last_error().decode())


def stream():
    if not torch.cuda.is_available():
        raise StkError('libstk kernels need a GPU (none visible); there is no '
                       'CPU fallback')
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors and
    tensors of another GPU than this process's (a kernel of this rank's stream
    over another rank's memory is a peer access at best, a fault at worst)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise StkError('libstk kernels need device tensors; got a %s tensor '
                       '(no GPU visible? there is no CPU fallback)' % t.device)
    if _process_device is not None and t.device.index != _process_device:
        raise StkError('tensor lives on cuda:%d but this process computes on '
                       'cuda:%d' % (t.device.index, _process_device))
    return t.data_ptr()


# The GPU of this process, fixed ONCE (comm.init_from_env, or the first use):
# torch.cuda.current_device() is thread-local and a new thread starts on device
# 0, so plans built in worker threads (heateq_mpi.py, multigrid.py) would
# otherwise be uploaded to cuda:0 on every rank.
_process_device = None


def set_process_device(index):
    """Pins the device of this process (one process per GPU)."""
    global _process_device
    _process_device = int(index)
    if torch.cuda.is_available():
        torch.cuda.set_device(_process_device)


def compute_device():
    """Device the slab of this process lives on -- the same answer in every
    thread."""
    global _process_device
    if _process_device is None:
        if not torch.cuda.is_available():
            return torch.device('cpu')
        _process_device = torch.cuda.current_device()
    return torch.device('cuda', _process_device)


def in_device_context(fn):
    """Wraps a worker-thread body so that the thread's current device (what
    hipMalloc and torch.cuda.current_stream() see) is the process's."""
    def run(*args, **kw):
        dev = compute_device()
        if dev.type != 'cuda':
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return run


from .host_malloc import give_back, host_heap_for_setup, keep_to_the_heap  # noqa: E402,F401


def to_dev(array, dtype=None):
    """Uploads a NumPy array (used for CSR arrays and small tables)."""
    t = torch.from_numpy(np.ascontiguousarray(array))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(compute_device())


def transpose(src, rows, cols, ld_src, dst, ld_dst, zero_to=0, src_off=0, dst_off=0):
    """dst[c, r] = src[r, c] on the device (stk_transpose); offsets in doubles."""
    check(lib().stk_transpose(stream(), rows, cols, ptr(src) + 8 * src_off, ld_src,
                              ptr(dst) + 8 * dst_off, ld_dst, zero_to))


def copy_block(src, rows, cols, ld_src, dst, ld_dst, src_off=0, dst_off=0):
    """dst[r, c] = src[r, c] between two leading dimensions (stk_copy_block)."""
    if rows and cols:
        check(lib().stk_copy_block(stream(), rows, cols, ptr(src) + 8 * src_off, ld_src,
                                   ptr(dst) + 8 * dst_off, ld_dst))


class DeviceCSR:
    """A CSR matrix resident on the device: int32 pattern + float64 values
    (reference source/mpi_shared_mem.py:46-48 fixes the same dtypes)."""
    def __init__(self, mat):
        import scipy.sparse as sp
        mat = sp.csr_matrix(mat)
        mat.sort_indices()
        self.shape = mat.shape
        self.nnz = mat.nnz
        self.indptr = to_dev(mat.indptr.astype(np.int32))
        self.indices = to_dev(mat.indices.astype(np.int32))
        self.data = to_dev(mat.data.astype(np.float64))

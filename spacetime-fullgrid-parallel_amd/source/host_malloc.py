"""Where the set-up's large host blocks come from (glibc's allocator; imports nothing
heavy, so that a driver can call keep_to_the_heap() before anything starts a thread).

A set-up creates and drops some hundred NumPy temporaries of 10-100 MB in half a dozen
threads.  By default every one of them is an mmap, page faults on first touch and a
munmap, and every pageable upload pins its pages: all of that takes the address-space
lock of the process, and the threads queue on it.  Measured at config 3
(profiles/r06_setup_malloc_ab.log): set-up 0.88-1.05 s by default, 0.67-0.71 s with the
allocator told to keep to the heap; the DEVICE is busy for 0.03 s of a set-up
(profiles/r06_setup_kernel_time.log) and a shorter interpreter switch interval changes
nothing (profiles/r06_setup_switch_interval.log): neither is what the threads wait for.
This is the MALLOC_MMAP_MAX_=0 / MALLOC_TRIM_THRESHOLD_ setting MPI codes are commonly
run with, made from inside the process.  STK_KEEP_MALLOC=1 leaves the allocator alone."""
import ctypes
import os
import threading

# glibc mallopt parameters (malloc.h)
_M_TRIM_THRESHOLD, _M_MMAP_MAX, _M_ARENA_MAX = -1, -4, -8
_lock = threading.Lock()
_users = 0
_permanent = False


def _libc():
    if os.environ.get('STK_KEEP_MALLOC') == '1':
        return None
    try:
        libc = ctypes.CDLL(None)
        libc.mallopt.argtypes, libc.mallopt.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int
        libc.malloc_trim.argtypes, libc.malloc_trim.restype = [ctypes.c_size_t], ctypes.c_int
        return libc
    except (OSError, AttributeError):  # not glibc: nothing to tune
        return None


def keep_to_the_heap():
    """For the rest of the process: one arena, no mmap for large blocks, no trimming.
    The drivers and bench.py call this first thing -- BEFORE anything starts a thread: a
    thread that already owns an arena of its own keeps mapping and unmapping 64 MB heaps
    for its blocks.  The price: the resident host memory of the process stays at the
    peak of its set-up (a few GB at config 3) instead of falling back after it; memory
    freed later is reused, not returned."""
    global _permanent
    libc = _libc()
    if libc is None:
        return False
    with _lock:
        libc.mallopt(_M_ARENA_MAX, 1)
        libc.mallopt(_M_MMAP_MAX, 0)
        libc.mallopt(_M_TRIM_THRESHOLD, -1)  # as a size: never
        if not _permanent:
            _huge_pages_for_the_heap(libc, int(os.environ.get('STK_HEAP_HUGE_GB', '3')))
        _permanent = True
    return True


def _huge_pages_for_the_heap(libc, gigabytes):
    """Where transparent huge pages are on `madvise` (they are on the pool's boxes), the
    next `gigabytes` of the heap are advised for them: the heap is grown by that much --
    address space, no memory until touched --, the range advised, the blocks freed
    (trimming is off: the range stays the top of the heap, where the set-up's blocks
    come from).  What glibc's glibc.malloc.hugetlb=1 tunable does, which can only be
    set when a process starts.  The first set-up of a process touches its 2.5 GB of
    new pages in 2 MB steps instead of 4 KB ones: 0.77-0.82 -> 0.68-0.69 s
    (profiles/r06_setup_thp_ab.log, measured with the tunable)."""
    try:
        if gigabytes <= 0 or '[never]' in open('/sys/kernel/mm/transparent_hugepage/enabled').read():
            return False
        libc.malloc.argtypes, libc.malloc.restype = [ctypes.c_size_t], ctypes.c_void_p
        libc.free.argtypes, libc.free.restype = [ctypes.c_void_p], None
        libc.madvise.argtypes, libc.madvise.restype = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int], ctypes.c_int
    except (OSError, AttributeError):
        return False
    giga, two_mb, madv_hugepage = 1 << 30, 2 << 20, 14
    blocks = [b for b in (libc.malloc(giga) for _ in range(gigabytes)) if b]
    done = False
    if blocks:
        lo, hi = min(blocks), max(blocks) + giga
        lo, hi = (lo + two_mb - 1) & ~(two_mb - 1), hi & ~(two_mb - 1)
        # only a run of blocks that sit side by side is one range of the heap
        if hi - lo <= (len(blocks) + 1) * giga:
            done = libc.madvise(lo, hi - lo, madv_hugepage) == 0
    for b in blocks:
        libc.free(b)
    return done


def give_back():
    """Returns the heap's free pages to the system now (malloc_trim): for a process under
    keep_to_the_heap() that has finished building operators and wants its resident host
    memory down -- three set-ups at config 3 leave 3.6 / 4.3 / 4.7 GB resident
    (profiles/r06_setup_faults.log).  The next set-up then touches its pages anew."""
    libc = _libc()
    if libc is not None:
        libc.malloc_trim(0)


class host_heap_for_setup:
    """The same for the duration of a set-up only (HeatEquationMPI.__init__): on exit the
    defaults are back and the free pages are returned to the system (malloc_trim).
    Helps as far as the planner threads allocate from the main arena (see
    keep_to_the_heap); a no-op once keep_to_the_heap() was called.  The limit of one
    arena for threads started from now on is the one setting that stays."""
    def __enter__(self):
        global _users
        self._libc = None if _permanent else _libc()
        if self._libc is not None:
            with _lock:
                _users += 1
                if _users == 1:
                    self._libc.mallopt(_M_ARENA_MAX, 1)
                    self._libc.mallopt(_M_MMAP_MAX, 0)
                    self._libc.mallopt(_M_TRIM_THRESHOLD, 2**31 - 1)
        return self

    def __exit__(self, *exc):
        global _users
        if self._libc is not None:
            with _lock:
                _users -= 1
                if _users == 0 and not _permanent:
                    self._libc.mallopt(_M_MMAP_MAX, 65536)
                    self._libc.mallopt(_M_TRIM_THRESHOLD, 128 * 1024)
                    self._libc.malloc_trim(0)
        return False

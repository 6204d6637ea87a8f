// Communication of the time-slab decomposition for hosts WITHOUT torch.distributed:
// the mpi4py call sites of the reference on RCCL over xGMI,
// one process per GPU.
//
//   reference (mpi4py)                               here
//   mpi_vector.py:209   allreduce of the local dot   stk_comm_allreduce_sum
//   mpi_vector.py:154-183 Isend/Irecv to rank +-1    stk_comm_halo_exchange (one
//                       (communicate_bdr)            grouped ncclSend/ncclRecv call)
//   mpi_vector.py:189-203, 224-239 arbitrary rows /  stk_comm_exchange (a batch of
//                       tiles between ranks          sends and receives as one group)
//
// RCCL is loaded at run time (dlopen of librccl.so.1, reusing the copy a host such
// as PyTorch has already mapped): libstk itself has no link-time dependency on it,
// and a host that never calls stk_comm_* never loads it.  The Python classes of
// this repository use torch.distributed (backend "nccl" = the same RCCL); these
// entry points are the equivalent for a C / C++ / Fortran caller of the C ABI.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "stk_common.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

template <typename F>
bool sym(void *lib, const char *name, F *out)
{
    *out = reinterpret_cast<F>(dlsym(lib, name));
    return *out != nullptr;
}

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return 0;
    void *lib = nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)  // first: a copy the host process has mapped already
        if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    for (size_t k = 0; !lib && k < sizeof(names) / sizeof(names[0]); ++k) lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        stk_set_error("stk_comm: cannot load librccl.so.1 (%s)", dlerror());
        return 1;
    }
    Rccl r;
    r.lib = lib;
    if (!(sym(lib, "ncclGetUniqueId", &r.GetUniqueId) && sym(lib, "ncclCommInitRank", &r.CommInitRank) &&
          sym(lib, "ncclCommDestroy", &r.CommDestroy) && sym(lib, "ncclAllReduce", &r.AllReduce) &&
          sym(lib, "ncclSend", &r.Send) && sym(lib, "ncclRecv", &r.Recv) && sym(lib, "ncclGroupStart", &r.GroupStart) &&
          sym(lib, "ncclGroupEnd", &r.GroupEnd) && sym(lib, "ncclGetErrorString", &r.GetErrorString))) {
        stk_set_error("stk_comm: librccl lacks an expected symbol (%s)", dlerror());
        return 1;
    }
    g_rccl = r;
    return 0;
}

#define STK_NCCL(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            stk_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_)); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

}  // namespace

struct stk_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
};

extern "C" int stk_comm_unique_id(void *id)
{
    STK_REQUIRE(id != nullptr, "stk_comm_unique_id: null pointer");
    if (load_rccl()) return 1;
    static_assert(sizeof(ncclUniqueId) == STK_COMM_ID_BYTES, "STK_COMM_ID_BYTES must be the size of ncclUniqueId");
    STK_NCCL(g_rccl.GetUniqueId(static_cast<ncclUniqueId *>(id)));
    return 0;
}

extern "C" int stk_comm_create(int32_t rank, int32_t size, const void *id, stk_comm **out)
{
    STK_REQUIRE(out && id && size >= 1 && rank >= 0 && rank < size, "stk_comm_create: bad arguments (rank %d of %d)",
                rank, size);
    if (load_rccl()) return 1;
    ncclUniqueId uid;
    __builtin_memcpy(&uid, id, sizeof(uid));
    stk_comm *c = new stk_comm();
    c->rank = rank, c->size = size;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, size, uid, rank);  // on the calling thread's current device
    if (r != ncclSuccess) {
        stk_set_error("stk_comm_create: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" int stk_comm_destroy(stk_comm *c)
{
    if (!c) return 0;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

extern "C" int stk_comm_allreduce_sum(stk_comm *c, void *stream, double *values, int32_t n)
{
    STK_REQUIRE(c && values && n >= 1, "stk_comm_allreduce_sum: bad arguments");
    STK_NCCL(g_rccl.AllReduce(values, values, (size_t)n, ncclFloat64, ncclSum, c->comm, stk_stream(stream)));
    return 0;
}

extern "C" int stk_comm_exchange(stk_comm *c, void *stream, int32_t n_send, const stk_comm_msg *sends, int32_t n_recv,
                                 const stk_comm_msg *recvs)
{
    STK_REQUIRE(c && n_send >= 0 && n_recv >= 0 && (n_send == 0 || sends) && (n_recv == 0 || recvs),
                "stk_comm_exchange: bad arguments");
    for (int k = 0; k < n_send; ++k)
        STK_REQUIRE(sends[k].buf && sends[k].count >= 0 && sends[k].peer >= 0 && sends[k].peer < c->size,
                    "stk_comm_exchange: send %d is malformed (peer %d of %d)", k, sends[k].peer, c->size);
    for (int k = 0; k < n_recv; ++k)
        STK_REQUIRE(recvs[k].buf && recvs[k].count >= 0 && recvs[k].peer >= 0 && recvs[k].peer < c->size,
                    "stk_comm_exchange: receive %d is malformed (peer %d of %d)", k, recvs[k].peer, c->size);
    if (n_send + n_recv == 0) return 0;
    hipStream_t st = stk_stream(stream);
    // one group: every transfer of the batch progresses together (matching is by
    // posting order per pair of ranks, as in source/comm.py)
    STK_NCCL(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int k = 0; k < n_send && bad == ncclSuccess; ++k)
        bad = g_rccl.Send(sends[k].buf, (size_t)sends[k].count, ncclFloat64, sends[k].peer, c->comm, st);
    for (int k = 0; k < n_recv && bad == ncclSuccess; ++k)
        bad = g_rccl.Recv(recvs[k].buf, (size_t)recvs[k].count, ncclFloat64, recvs[k].peer, c->comm, st);
    ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
        stk_set_error("stk_comm_exchange: %s", g_rccl.GetErrorString(bad != ncclSuccess ? bad : end));
        return 1;
    }
    return 0;
}

extern "C" int stk_comm_halo_exchange(stk_comm *c, void *stream, int32_t M, const double *send_first,
                                      const double *send_last, double *recv_lo, double *recv_hi)
{
    STK_REQUIRE(c && M > 0, "stk_comm_halo_exchange: bad arguments");
    stk_comm_msg sends[2], recvs[2];
    int ns = 0, nr = 0;
    if (c->rank > 0) {  // my first time row is the lower neighbour's X_loc_bdr[-1]; its last row my X_loc_bdr[0]
        STK_REQUIRE(send_first && recv_lo, "stk_comm_halo_exchange: rank %d has a lower neighbour but no buffers for it",
                    c->rank);
        sends[ns++] = stk_comm_msg{const_cast<double *>(send_first), M, c->rank - 1};
        recvs[nr++] = stk_comm_msg{recv_lo, M, c->rank - 1};
    }
    if (c->rank + 1 < c->size) {
        STK_REQUIRE(send_last && recv_hi, "stk_comm_halo_exchange: rank %d has an upper neighbour but no buffers for it",
                    c->rank);
        sends[ns++] = stk_comm_msg{const_cast<double *>(send_last), M, c->rank + 1};
        recvs[nr++] = stk_comm_msg{recv_hi, M, c->rank + 1};
    }
    return stk_comm_exchange(c, stream, ns, sends, nr, recvs);
}

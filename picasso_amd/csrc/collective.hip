// collective.hip — the one exchange of the sharded path: every rank's localization table on every GPU.
//
// The reference has no distributed code (its workers split the movie frame by frame inside one process,
// picasso/localize.py:438-454).  Here every GPU owns a contiguous frame range and the tables are all-gathered at
// the end: RCCL is called directly — the library, not a Python framework, owns the collective, so
// a host without torch can shard.  RCCL is loaded at the first pmi_comm_* call (dlopen), never at import.
//
//   pmi_comm_unique_id   rank 0 makes the 128-byte id; the HOST distributes it (file, socket, MPI, a torch store)
//   pmi_comm_init        every rank: ncclCommInitRank on its current device
//   pmi_allgather_locs   counts (one int64 per rank) and the padded column-major tables, one grouped submission on
//                        the caller's stream; asynchronous
//   pmi_compact_gathered_dev   rank-major padded tables -> one contiguous column-major table, row counts read on the
//                        device (no host round trip); contiguous frame shards keep it frame-sorted, the order
//                        picasso/gaussmle.py:1036 produces
#include <dlfcn.h>

#include <algorithm>

#include "pmi_common.h"

namespace pmi {

namespace {

// the few RCCL entry points used, with the ABI of rccl.h (ncclUniqueId is 128 bytes, passed by value)
struct UniqueId { char internal[128]; };
typedef void *Comm;
enum { RCCL_INT32 = 2, RCCL_INT64 = 4 };      // ncclInt32, ncclInt64
struct Rccl {
    int (*GetUniqueId)(UniqueId *);
    int (*CommInitRank)(Comm *, int, UniqueId, int);
    int (*CommDestroy)(Comm);
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*CommCount)(Comm, int *);
    int (*CommUserRank)(Comm, int *);
    const char *(*GetErrorString)(int);
    bool ok = false;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.ok) return PMI_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error("RCCL not found: %s", dlerror()); return PMI_ERR_HIP; }
#define PMI_SYM(field, name)                                                                  \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                  \
    if (!g_rccl.field) { set_error("RCCL lacks %s", name); return PMI_ERR_HIP; }
    PMI_SYM(GetUniqueId, "ncclGetUniqueId")
    PMI_SYM(CommInitRank, "ncclCommInitRank")
    PMI_SYM(CommDestroy, "ncclCommDestroy")
    PMI_SYM(AllGather, "ncclAllGather")
    PMI_SYM(GroupStart, "ncclGroupStart")
    PMI_SYM(GroupEnd, "ncclGroupEnd")
    PMI_SYM(CommCount, "ncclCommCount")
    PMI_SYM(CommUserRank, "ncclCommUserRank")
    PMI_SYM(GetErrorString, "ncclGetErrorString")
#undef PMI_SYM
    g_rccl.ok = true;
    return PMI_OK;
}

int rccl_fail(int rc, const char *what)
{
    set_error("RCCL error %d (%s) in %s", rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", what);
    return PMI_ERR_HIP;
}
#define PMI_RCCL(call)                                      \
    do {                                                    \
        int rc_ = (call);                                   \
        if (rc_ != 0) return rccl_fail(rc_, #call);         \
    } while (0)

struct CommBox { Comm comm; int world, rank; };

// out column c = rows of rank 0, rank 1, ... ; src = world tables of ncols x cap cells, counts on the device
__global__ void compact_gathered_kernel(const int32_t *__restrict__ src, const int64_t *__restrict__ counts, int world,
                                        int ncols, int64_t cap, int32_t *__restrict__ dst, int64_t dst_cap,
                                        int64_t *__restrict__ d_total)
{
    int64_t total = 0;
    for (int r = 0; r < world; r++) total += counts[r] < cap ? counts[r] : cap;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && d_total) *d_total = total;
    const int c = blockIdx.y;
    int64_t base = 0;
    for (int r = 0; r < world; r++) {
        const int64_t n = counts[r] < cap ? counts[r] : cap;
        const int32_t *s = src + ((int64_t)r * ncols + c) * cap;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            if (base + i < dst_cap) dst[(int64_t)c * dst_cap + base + i] = s[i];
        base += n;
    }
}

}  // namespace

}  // namespace pmi

extern "C" {

int pmi_comm_available(void)
{
    return pmi::load_rccl();      // PMI_OK when librccl and the entry points used here resolve; nothing is created
}

int pmi_comm_unique_id(void *id128)
{
    using namespace pmi;
    int rc = load_rccl();
    if (rc != PMI_OK) return rc;
    if (!id128) { set_error("null pointer"); return PMI_ERR_ARG; }
    PMI_RCCL(g_rccl.GetUniqueId(reinterpret_cast<UniqueId *>(id128)));
    return PMI_OK;
}

int pmi_comm_init(const void *id128, int world, int rank, void **comm)
{
    using namespace pmi;
    int rc = load_rccl();
    if (rc != PMI_OK) return rc;
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator arguments (world %d, rank %d)", world, rank); return PMI_ERR_ARG; }
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    CommBox *box = new CommBox{nullptr, world, rank};
    int r = g_rccl.CommInitRank(&box->comm, world, id, rank);
    if (r != 0) { delete box; return rccl_fail(r, "ncclCommInitRank"); }
    *comm = box;
    return PMI_OK;
}

int pmi_comm_destroy(void *comm)
{
    using namespace pmi;
    if (!comm) return PMI_OK;
    CommBox *box = static_cast<CommBox *>(comm);
    int r = g_rccl.ok ? g_rccl.CommDestroy(box->comm) : 0;
    delete box;
    if (r != 0) return rccl_fail(r, "ncclCommDestroy");
    return PMI_OK;
}

int pmi_comm_info(void *comm, int *world, int *rank)
{
    if (!comm) { pmi::set_error("null communicator"); return PMI_ERR_ARG; }
    pmi::CommBox *box = static_cast<pmi::CommBox *>(comm);
    // what RCCL itself says about the communicator (not what pmi_comm_init was told)
    using namespace pmi;
    int w = -1, r = -1;
    PMI_RCCL(g_rccl.CommCount(box->comm, &w));
    PMI_RCCL(g_rccl.CommUserRank(box->comm, &r));
    if (w != box->world || r != box->rank) { pmi::set_error("communicator reports world %d rank %d, made as %d / %d", w, r, box->world, box->rank); return PMI_ERR_HIP; }
    if (world) *world = w;
    if (rank) *rank = r;
    return PMI_OK;
}

int pmi_comm_library_path(char *path, size_t path_len)
{
    // the shared object the collective entry points resolved to (torch bundles an RCCL of its own next to /opt/rocm's)
    using namespace pmi;
    int rc = load_rccl();
    if (rc != PMI_OK) return rc;
    Dl_info di;
    if (!path || !path_len) { set_error("null pointer"); return PMI_ERR_ARG; }
    if (!dladdr(reinterpret_cast<void *>(g_rccl.AllGather), &di) || !di.dli_fname) { set_error("dladdr cannot place ncclAllGather"); return PMI_ERR_HIP; }
    strncpy(path, di.dli_fname, path_len - 1);
    path[path_len - 1] = 0;
    return PMI_OK;
}

int pmi_allgather_locs(void *comm, const void *d_table, int ncols, int64_t cap, const int64_t *d_n,
                       void *d_all_tables, int64_t *d_all_counts, void *stream)
{
    using namespace pmi;
    if (!comm || !d_table || !d_n || !d_all_tables || !d_all_counts || ncols < 1 || cap < 1) { set_error("bad all-gather arguments"); return PMI_ERR_ARG; }
    CommBox *box = static_cast<CommBox *>(comm);
    hipStream_t s = (hipStream_t)stream;
    PMI_RCCL(g_rccl.GroupStart());
    int r1 = g_rccl.AllGather(d_n, d_all_counts, 1, RCCL_INT64, box->comm, s);
    int r2 = g_rccl.AllGather(d_table, d_all_tables, (size_t)ncols * (size_t)cap, RCCL_INT32, box->comm, s);
    int r3 = g_rccl.GroupEnd();
    if (r1 != 0) return rccl_fail(r1, "ncclAllGather(counts)");
    if (r2 != 0) return rccl_fail(r2, "ncclAllGather(tables)");
    if (r3 != 0) return rccl_fail(r3, "ncclGroupEnd");
    return PMI_OK;
}

int pmi_compact_gathered_dev(const void *d_all_tables, const int64_t *d_all_counts, int world, int ncols, int64_t cap,
                             void *d_table, int64_t table_cap, int64_t *d_total, void *stream)
{
    using namespace pmi;
    if (!d_all_tables || !d_all_counts || !d_table || world < 1 || ncols < 1 || cap < 1 || table_cap < 1) { set_error("bad compaction arguments"); return PMI_ERR_ARG; }
    const unsigned bx = (unsigned)std::min<int64_t>((cap + 255) / 256, 1024);
    hipLaunchKernelGGL(compact_gathered_kernel, dim3(bx, (unsigned)ncols), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t *)d_all_tables, d_all_counts, world, ncols, cap, (int32_t *)d_table, table_cap, d_total);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

}  // extern "C"

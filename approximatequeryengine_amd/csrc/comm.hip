// comm.hip — the collective of the multi-GPU path behind the C ABI: RCCL all-reduce of the moment vectors over xGMI.
//
// The reference merges its workers inside one process — a mutex-guarded vector, future.get() concatenation, a CAS loop
// on atomic<double> (custom_bplus_db.cpp:948-951, 966-967, 2031-2036).  Across GPUs the same merge is ONE all-reduce
// SUM of (n, S - c n, Q) x {leader, others} per convergence step — or per batch of queries (aqe_batch_run_sharded).
// The library has no link-time dependency on RCCL: librccl is opened on first use (the copy already loaded into the
// process, e.g. PyTorch's, is taken when there is one), so single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "host.hpp"

using namespace aqe;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    int version = 0;
    std::string error;
};

Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char* e = std::getenv("AQE_RCCL_LIB")) names.push_back(e);
        for (const char* n : {"librccl.so.1", "librccl.so"}) names.push_back(n);
        names.push_back("/opt/rocm/lib/librccl.so.1");
        // a copy already in the process (PyTorch-ROCm bundles one beside its own HIP runtime) comes first
        for (const std::string& n : names)
            if (!R.handle) R.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD);
        for (const std::string& n : names)
            if (!R.handle) R.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!R.handle) {
            const char* why = dlerror();
            R.error = std::string("librccl not found (set AQE_RCCL_LIB): ") + (why ? why : "");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(R.handle, name);
            if (!p && R.error.empty()) R.error = std::string("librccl lacks ") + name;
            return p;
        };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
        R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        R.GetVersion = reinterpret_cast<decltype(R.GetVersion)>(sym("ncclGetVersion"));
        R.CommAbort = reinterpret_cast<decltype(R.CommAbort)>(sym("ncclCommAbort"));
        // The function-pointer types above come from THIS build's <rccl/rccl.h>; the library found at run time may be
        // another copy (PyTorch's).  The calls used here have kept their signatures across NCCL 2.x: a library of another
        // major version is refused rather than called through the wrong prototypes.
        if (R.error.empty() && R.GetVersion) {
            if (R.GetVersion(&R.version) != ncclSuccess) R.error = "ncclGetVersion failed";
            else {
                const int major = R.version >= 10000 ? R.version / 10000 : R.version / 1000;  // NCCL_VERSION(): X*10000 + Y*100 + Z from 2.9 on
                if (major != NCCL_MAJOR)
                    R.error = "librccl reports NCCL version " + std::to_string(R.version) + ", this library was built against major version " + std::to_string(NCCL_MAJOR);
            }
        }
    });
    return R;
}

int rccl_ready(aqe_ctx* c) {
    Rccl& R = rccl();
    if (!R.error.empty()) return fail(c, AQE_ERR_UNSUPPORTED, R.error);
    return AQE_OK;
}

#define RCCLCHK(ctx, expr)                                                                          \
    do {                                                                                            \
        ncclResult_t r__ = (expr);                                                                  \
        if (r__ != ncclSuccess) return fail(ctx, AQE_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r__)); \
    } while (0)

}  // namespace

struct aqe_comm {
    aqe_ctx* ctx = nullptr;
    aqe_mailbox* mailbox = nullptr;  // aqe_comm_create_mailbox: SUMs go through the peer-mapped mailbox, no RCCL involved
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    int device = 0;  // (kept beside ctx: a communicator may be destroyed after its context)
};

extern "C" {

static_assert(AQE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as RCCL's ncclUniqueId");

int aqe_comm_unique_id(void* id) {
    if (!id) return AQE_ERR_INVALID;
    int rc = rccl_ready(nullptr);
    if (rc != AQE_OK) return rc;
    ncclUniqueId u;
    RCCLCHK(nullptr, rccl().GetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return AQE_OK;
}

int aqe_comm_create(aqe_ctx* c, const void* id, int nranks, int rank, aqe_comm** out) {
    if (!c || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return c ? fail(c, AQE_ERR_INVALID, "bad communicator arguments") : AQE_ERR_INVALID;
    int rc = rccl_ready(c);
    if (rc != AQE_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    std::unique_ptr<aqe_comm> m(new aqe_comm);
    m->ctx = c;
    m->device = c->device;
    m->nranks = nranks;
    m->rank = rank;
    RCCLCHK(c, rccl().CommInitRank(&m->comm, nranks, u, rank));
    *out = m.release();
    return AQE_OK;
}

int aqe_comm_create_mailbox(aqe_ctx* c, aqe_mailbox* mb, aqe_comm** out) {
    if (!c || !mb || !out) return AQE_ERR_INVALID;
    std::unique_ptr<aqe_comm> m(new aqe_comm);
    m->ctx = c;
    m->device = c->device;
    m->mailbox = mb;
    int rc = aqe_mailbox_info(mb, &m->nranks, &m->rank);
    if (rc != AQE_OK) return rc;
    *out = m.release();
    return AQE_OK;
}

int aqe_comm_create_all(aqe_ctx* const* ctxs, int n, aqe_comm** out_n) {
    if (!ctxs || !out_n || n < 1) return AQE_ERR_INVALID;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return AQE_ERR_INVALID;
    aqe_ctx* c0 = ctxs[0];
    int rc = rccl_ready(c0);
    if (rc != AQE_OK) return rc;
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return fail(c0, AQE_ERR_INVALID, "aqe_comm_create_all: one context per GPU (two share a device)");
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    RCCLCHK(c0, rccl().CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        aqe_comm* m = new aqe_comm;
        m->ctx = ctxs[i];
        m->device = ctxs[i]->device;
        m->comm = comms[i];
        m->nranks = n;
        m->rank = i;
        out_n[i] = m;
    }
    return AQE_OK;
}

void aqe_comm_destroy(aqe_comm* m) {
    if (!m) return;
    if (m->comm && rccl().CommDestroy) {
        (void)hipSetDevice(m->device);
        (void)rccl().CommDestroy(m->comm);
    }
    delete m;
}

// A rank that fails between two collectives of a sharded run must not simply return: its peers would block in the next
// all-reduce for ever.  The communicator is aborted instead (ncclCommAbort: peers' pending and later collectives fail)
// and the handle is dead from then on; the caller tears the group down.
static int abort_after(aqe_comm* m, int rc) {
    if (rc != AQE_OK && m && m->comm && m->nranks > 1 && rccl().CommAbort) {
        (void)hipSetDevice(m->device);
        (void)rccl().CommAbort(m->comm);
        m->comm = nullptr;
    }
    return rc;
}

int aqe_comm_info(const aqe_comm* m, int* nranks, int* rank) {
    if (!m) return AQE_ERR_INVALID;
    if (nranks) *nranks = m->nranks;
    if (rank) *rank = m->rank;
    return AQE_OK;
}

static int all_reduce(aqe_comm* m, double* buf, uint64_t count, ncclRedOp_t op, void* stream) {
    if (!m || !buf) return AQE_ERR_INVALID;
    aqe_ctx* c = m->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (count == 0) return AQE_OK;
    if (m->mailbox) {
        if (op != ncclSum) return fail(c, AQE_ERR_UNSUPPORTED, "a mailbox communicator sums only");
        return aqe_mailbox_all_reduce_sum(m->mailbox, buf, count, s);
    }
    if (!m->comm) return fail(c, AQE_ERR_INVALID, "the communicator was aborted after a failure on this rank");
    RCCLCHK(c, rccl().AllReduce(buf, buf, static_cast<size_t>(count), ncclFloat64, op, m->comm, s));
    return AQE_OK;
}

int aqe_comm_all_reduce_sum(aqe_comm* m, double* dev_buf, uint64_t count, void* stream) { return all_reduce(m, dev_buf, count, ncclSum, stream); }
int aqe_comm_all_reduce_max(aqe_comm* m, double* dev_buf, uint64_t count, void* stream) { return all_reduce(m, dev_buf, count, ncclMax, stream); }

int aqe_comm_group_start(void) {
    int rc = rccl_ready(nullptr);
    if (rc != AQE_OK) return rc;
    RCCLCHK(nullptr, rccl().GroupStart());
    return AQE_OK;
}

int aqe_comm_group_end(void) {
    int rc = rccl_ready(nullptr);
    if (rc != AQE_OK) return rc;
    RCCLCHK(nullptr, rccl().GroupEnd());
    return AQE_OK;
}

// One step of a batch over the ranks of a communicator: this shard's round totals of every plan (ONE launch), ONE
// all-reduce SUM of the whole buffer, ONE launch that replays every plan's stop rules on the reduced totals.
int aqe_batch_run_sharded(aqe_batch* b, aqe_comm* m, double* dev_totals, uint64_t row_stride, uint32_t n_plans, void* stream) {
    if (!b || !m || !dev_totals) return AQE_ERR_INVALID;
    int rc = aqe_batch_enqueue_sweeps(b, dev_totals, row_stride);
    if (rc == AQE_OK) rc = aqe_batch_join(b, stream);
    if (rc == AQE_OK) rc = aqe_comm_all_reduce_sum(m, dev_totals, static_cast<uint64_t>(n_plans) * row_stride, stream);
    if (rc == AQE_OK) rc = aqe_batch_enqueue_replays(b, dev_totals, row_stride, stream);
    return abort_after(m, rc);
}

// One query over the ranks of a communicator, in the form the plan offers: batched (one collective for the whole
// query) when it has a totals form, else one collective per convergence step; a due top-up (DB.cpp:1031-1040) is
// finished with the stepwise step.  dev_vec: max(AQE_MOMENT_VEC, totals_len) doubles of device memory.
int aqe_plan_run_sharded(aqe_plan* p, aqe_comm* m, double* dev_vec, void* stream, aqe_result* out) {
    if (!p || !m || !dev_vec || !out) return AQE_ERR_INVALID;
    uint32_t len = 0, rounds = 0;
    int32_t has_topup = 0;
    int rc = aqe_plan_totals_len(p, &len);
    if (rc == AQE_OK) rc = aqe_plan_rounds(p, &rounds, &has_topup);
    if (rc != AQE_OK) return rc;
    aqe_ctx* c = m->ctx;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    auto step = [&](uint32_t r) {
        int e = hipMemsetAsync(dev_vec, 0, sizeof(double) * AQE_MOMENT_VEC, s) == hipSuccess ? AQE_OK : AQE_ERR_HIP;  // a launch that leaves early writes nothing
        if (e == AQE_OK) e = aqe_plan_enqueue_round(p, r, dev_vec, s);
        if (e == AQE_OK) e = aqe_comm_all_reduce_sum(m, dev_vec, AQE_MOMENT_VEC, s);
        if (e == AQE_OK) e = aqe_plan_enqueue_update(p, r, dev_vec, s);
        return e;
    };
    if (len) {
        rc = aqe_plan_enqueue_sweep_totals(p, dev_vec, s);
        if (rc == AQE_OK) rc = aqe_comm_all_reduce_sum(m, dev_vec, len, s);
        if (rc == AQE_OK) rc = aqe_plan_enqueue_replay(p, dev_vec, s);
        if (rc == AQE_OK) rc = aqe_plan_fetch(p, out, s);
        if (rc != AQE_OK || !out->topup_pending) return abort_after(m, rc);
        rc = step(rounds);  // every rank reads the same mark: every rank comes here
    } else {
        rc = aqe_plan_reset(p, s);
        for (uint32_t r = 0; rc == AQE_OK && r < rounds + (has_topup ? 1u : 0u); ++r) rc = step(r);
    }
    if (rc == AQE_OK) rc = aqe_plan_enqueue_finalize(p, s);
    if (rc == AQE_OK) rc = aqe_plan_fetch(p, out, s);
    return abort_after(m, rc);
}

}  // extern "C"

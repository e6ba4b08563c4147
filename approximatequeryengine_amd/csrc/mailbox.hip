// mailbox.hip — the one-shot peer-mapped all-reduce of SURVEY §5 / §8e: for the moment vectors of the multi-GPU path
// (tens to a few hundred doubles) a collective is pure latency, and on ≤ 8 GPUs joined point-to-point by xGMI the shortest
// path is for every rank to WRITE its vector straight into every peer's memory and for every rank to add up what arrived.
//
// Each rank owns one mailbox in its own HBM: two parities x nranks slots of kMailboxDoubles doubles + one flag line per
// slot.  An all-reduce of epoch e is ONE launch of ONE workgroup per rank:
//     (1) store my vector into slot [e & 1][my rank] of EVERY rank's mailbox (system-scope stores over xGMI; mine locally),
//     (2) drain them, then store e into the matching flag of every mailbox,
//     (3) spin on MY mailbox's nranks flags of this parity until each holds e (bounded: a rank that never shows up ends the
//         launch with the timeout mark set instead of hanging the GPU),
//     (4) add the nranks slots up IN RANK ORDER — every rank computes the same floating-point sum, bit for bit, which is what
//         lets every rank take the same stop decision without a broadcast — and write it over the caller's vector.
// Two parities are enough: a rank can enter epoch e + 2 (and overwrite parity e's slots) only after it finished e + 1, which
// needs every rank's e + 1 flag, which a rank raises only after its own epoch-e launch — reads included — has ended.
//
// The reference has no counterpart across devices; inside its one process this is the CAS loop on atomic<double> and the
// mutex-guarded merges of custom_bplus_db.cpp:948-951, 966-967, 2031-2036.
// Peers in other processes are mapped through HIP IPC handles (aqe_mailbox_handle / aqe_mailbox_connect); peers of the
// same process (one context per GPU) are connected directly (aqe_mailbox_connect_local).
#include "host.hpp"

using namespace aqe;

namespace {

constexpr uint32_t kMailboxDoubles = 4096;  // capacity of one message
constexpr uint32_t kMailboxMaxRanks = 16;
constexpr uint32_t kFlagStride = 8;          // flags on 64-byte lines of their own
constexpr unsigned long long kSpinTicks = 200000000ull;  // ~2 s of the 100 MHz wall clock

static_assert(kMailboxDoubles == AQE_MAILBOX_MAX_DOUBLES && kMailboxMaxRanks == AQE_MAILBOX_MAX_RANKS, "include/aqe_hip.h");

struct MailboxPeers {
    double* base[kMailboxMaxRanks];
};

__host__ __device__ inline size_t mailbox_slot_doubles(uint32_t nranks) { return 2ull * nranks * kMailboxDoubles; }
__host__ __device__ inline size_t mailbox_bytes(uint32_t nranks) { return (mailbox_slot_doubles(nranks) + 2ull * nranks * kFlagStride) * sizeof(double); }

__device__ inline double* slot_of(double* base, uint32_t nranks, uint32_t parity, uint32_t rank) {
    return base + (static_cast<size_t>(parity) * nranks + rank) * kMailboxDoubles;
}
__device__ inline unsigned long long* flag_of(double* base, uint32_t nranks, uint32_t parity, uint32_t rank) {
    return reinterpret_cast<unsigned long long*>(base + mailbox_slot_doubles(nranks)) + (static_cast<size_t>(parity) * nranks + rank) * kFlagStride;
}

__global__ __launch_bounds__(256) void k_mailbox_all_reduce(MailboxPeers peers, double* __restrict__ vec, uint32_t count, uint32_t nranks, uint32_t rank,
                                                             unsigned long long epoch, unsigned* __restrict__ status) {
    const uint32_t parity = static_cast<uint32_t>(epoch & 1ull), t = threadIdx.x;
    // (1) my vector into my slot of every mailbox
    for (uint32_t p = 0; p < nranks; ++p) {
        double* dst = slot_of(peers.base[p], nranks, parity, rank);
        for (uint32_t i = t; i < count; i += blockDim.x) __hip_atomic_store(dst + i, vec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (2) every store of this workgroup has left before any flag does
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t < nranks) __hip_atomic_store(flag_of(peers.base[t], nranks, parity, rank), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // (3) wait for every rank's flag in MY mailbox
    __shared__ unsigned late;
    if (t == 0) late = 0;
    __syncthreads();
    if (t < nranks) {
        unsigned long long* f = flag_of(peers.base[rank], nranks, parity, t);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
            if (wall_clock64() - t0 > kSpinTicks) { atomicOr(&late, 1u << t); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (late) {  // (the vector is left as it was: the caller reads the mark, aqe_mailbox_status)
        if (t == 0) __hip_atomic_store(status, late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // (4) the sum, in rank order
    double* mine = peers.base[rank];
    for (uint32_t i = t; i < count; i += blockDim.x) {
        double s = 0.0;
        for (uint32_t r = 0; r < nranks; ++r) s += __hip_atomic_load(slot_of(mine, nranks, parity, r) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        vec[i] = s;
    }
}

}  // namespace

struct aqe_mailbox {
    aqe_ctx* ctx = nullptr;
    int nranks = 0, rank = -1;
    double* mine = nullptr;
    MailboxPeers peers{};
    bool opened[kMailboxMaxRanks] = {false};  // mapped through an IPC handle (to be closed)
    bool connected = false;
    unsigned long long epoch = 0;
    unsigned* status = nullptr;       // pinned, mapped: ranks that were late, by bit
    unsigned* status_dev = nullptr;
};

extern "C" {

int aqe_mailbox_create(aqe_ctx* c, int nranks, int rank, aqe_mailbox** out) {
    if (!c || !out || nranks < 1 || nranks > static_cast<int>(kMailboxMaxRanks) || rank < 0 || rank >= nranks) return fail(c, AQE_ERR_INVALID, "aqe_mailbox_create: 1 <= nranks <= 16, 0 <= rank < nranks");
    HIPCHK(c, hipSetDevice(c->device));
    std::unique_ptr<aqe_mailbox> m(new aqe_mailbox());
    m->ctx = c; m->nranks = nranks; m->rank = rank;
    // (fine-grained where the runtime gives it — the peers' stores must be seen while a kernel of this device is polling —
    // otherwise ordinary device memory: every access of the kernel is a system-scope one either way)
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&m->mine), mailbox_bytes(nranks), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&m->mine), mailbox_bytes(nranks)));
    }
    hipError_t e = hipMemset(m->mine, 0, mailbox_bytes(nranks));
    if (e == hipSuccess) e = hipDeviceSynchronize();  // zero before anyone can be handed the address
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&m->status), 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { *m->status = 0; e = hipHostGetDevicePointer(reinterpret_cast<void**>(&m->status_dev), m->status, 0); }
    if (e != hipSuccess) {
        (void)hipFree(m->mine);
        if (m->status) (void)hipHostFree(m->status);
        return fail(c, AQE_ERR_HIP, std::string("aqe_mailbox_create: ") + hipGetErrorString(e));
    }
    m->peers.base[rank] = m->mine;
    m->connected = nranks == 1;
    *out = m.release();
    return AQE_OK;
}

int aqe_mailbox_handle(aqe_mailbox* m, void* handle64) {
    if (!m || !handle64) return AQE_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == AQE_MAILBOX_HANDLE_BYTES, "IPC handle size");
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    hipIpcMemHandle_t h;
    HIPCHK(m->ctx, hipIpcGetMemHandle(&h, m->mine));
    std::memcpy(handle64, &h, sizeof h);
    return AQE_OK;
}

int aqe_mailbox_connect(aqe_mailbox* m, const void* handles) {
    if (!m || (!handles && m->nranks > 1)) return AQE_ERR_INVALID;
    if (m->connected) return AQE_OK;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    const char* h = static_cast<const char*>(handles);
    for (int r = 0; r < m->nranks; ++r) {
        if (r == m->rank) continue;
        hipIpcMemHandle_t ih;
        std::memcpy(&ih, h + static_cast<size_t>(r) * AQE_MAILBOX_HANDLE_BYTES, sizeof ih);
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(m->ctx, AQE_ERR_HIP, "aqe_mailbox_connect: mapping rank " + std::to_string(r) + "'s mailbox: " + hipGetErrorString(e));
        m->peers.base[r] = static_cast<double*>(p);
        m->opened[r] = true;
    }
    m->connected = true;
    return AQE_OK;
}

int aqe_mailbox_connect_local(aqe_mailbox* const* ms, int n) {
    if (!ms || n < 1) return AQE_ERR_INVALID;
    for (int i = 0; i < n; ++i)
        if (!ms[i] || ms[i]->nranks != n || ms[i]->rank != i) return fail(ms[0] ? ms[0]->ctx : nullptr, AQE_ERR_INVALID, "aqe_mailbox_connect_local: mailbox i must be rank i of n");
    for (int i = 0; i < n; ++i) {
        aqe_mailbox* m = ms[i];
        HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
        for (int r = 0; r < n; ++r) {
            if (r == i) continue;
            if (ms[r]->ctx->device != m->ctx->device) {
                int can = 0;
                HIPCHK(m->ctx, hipDeviceCanAccessPeer(&can, m->ctx->device, ms[r]->ctx->device));
                if (!can) return fail(m->ctx, AQE_ERR_UNSUPPORTED, "aqe_mailbox_connect_local: no peer access between the devices");
                hipError_t e = hipDeviceEnablePeerAccess(ms[r]->ctx->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(m->ctx, AQE_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
            }
            m->peers.base[r] = ms[r]->mine;
        }
        m->connected = true;
    }
    return AQE_OK;
}

int aqe_mailbox_all_reduce_sum(aqe_mailbox* m, double* dev_buf, uint64_t count, void* stream) {
    if (!m || (!dev_buf && count)) return AQE_ERR_INVALID;
    if (!m->connected) return fail(m->ctx, AQE_ERR_INVALID, "aqe_mailbox_all_reduce_sum: connect the mailbox first");
    if (count > kMailboxDoubles) return fail(m->ctx, AQE_ERR_INVALID, "aqe_mailbox_all_reduce_sum: at most " + std::to_string(kMailboxDoubles) + " doubles per call");
    if (count == 0) return AQE_OK;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->ctx->stream;
    m->epoch += 1;  // (every rank counts its calls: the same collective has the same epoch everywhere)
    hipLaunchKernelGGL(k_mailbox_all_reduce, dim3(1), dim3(256), 0, s, m->peers, dev_buf, static_cast<uint32_t>(count), static_cast<uint32_t>(m->nranks),
                       static_cast<uint32_t>(m->rank), m->epoch, m->status_dev);
    HIPCHK(m->ctx, hipGetLastError());
    return AQE_OK;
}

int aqe_mailbox_info(const aqe_mailbox* m, int* nranks, int* rank) {
    if (!m) return AQE_ERR_INVALID;
    if (nranks) *nranks = m->nranks;
    if (rank) *rank = m->rank;
    return AQE_OK;
}

int aqe_mailbox_status(aqe_mailbox* m, uint32_t* late_ranks) {
    if (!m || !late_ranks) return AQE_ERR_INVALID;
    *late_ranks = *const_cast<volatile unsigned*>(m->status);
    return AQE_OK;
}

void aqe_mailbox_destroy(aqe_mailbox* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < m->nranks; ++r)
        if (m->opened[r]) (void)hipIpcCloseMemHandle(m->peers.base[r]);
    if (m->mine) (void)hipFree(m->mine);
    if (m->status) (void)hipHostFree(m->status);
    delete m;
}

}  // extern "C"

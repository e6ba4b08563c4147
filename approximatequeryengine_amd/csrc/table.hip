// table.hip — the table in HBM: context lifecycle, staging from host rows and files, synthetic tables, the lazily
// built side structures (key columns, zone variances, the amount-sorted column).  Host code; kernels are in kernels.hip,
// grouped.hip and sort.hip.
#include "host.hpp"

using namespace aqe;

namespace aqe {

thread_local std::string g_create_error;

int fail(aqe_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

// Shift c of the shifted moments: the mean of the table's first rows (up to 1024), so that one outlying
// first row cannot push c outside the data's range.  Every shard of a table must use the same value.
constexpr uint64_t kShiftRows = 1024;
double shift_of_rows(const aqe_record* rows, uint64_t n) {
    const uint64_t m = std::min<uint64_t>(n, kShiftRows);
    double s = 0.0;
    for (uint64_t i = 0; i < m; ++i) s += rows[i].amount;
    return m ? s / static_cast<double>(m) : 0.0;
}
// Coefficient of variation of the same rows: a rough idea of how many samples a CLT query needs before its error
// rule can hold, used only to choose between two equivalent launch forms (plans.hip create_plan).
double cv_of_values(const double* x, uint64_t m) {
    if (m < 2) return 0.0;
    double mean = 0.0;
    for (uint64_t i = 0; i < m; ++i) mean += x[i];
    mean /= static_cast<double>(m);
    double ss = 0.0;
    for (uint64_t i = 0; i < m; ++i) ss += (x[i] - mean) * (x[i] - mean);
    return mean != 0.0 ? std::sqrt(ss / static_cast<double>(m - 1)) / std::fabs(mean) : 0.0;
}
double cv_of_rows(const aqe_record* rows, uint64_t n) {
    const uint64_t m = std::min<uint64_t>(n, kShiftRows);
    std::vector<double> x(m);
    for (uint64_t i = 0; i < m; ++i) x[i] = rows[i].amount;
    return cv_of_values(x.data(), m);
}

void free_table(aqe_ctx* c) {
    if (c->owns_table) {
        if (c->amount) (void)hipFree(c->amount);
        if (c->aos) (void)hipFree(c->aos);
    }
    if (c->sorted_amount) (void)hipFree(c->sorted_amount);
    if (c->sorted_row) (void)hipFree(c->sorted_row);
    for (int k = 0; k < 2; ++k) {
        if (c->keycol[k]) (void)hipFree(c->keycol[k]);
        c->keycol[k] = nullptr;
    }
    for (auto& kv : c->stride_views) {
        (void)hipFree(kv.second.amount);
        for (int32_t* k : kv.second.keys)
            if (k) (void)hipFree(k);
    }
    c->stride_views.clear();
    c->view_bytes = 0;
    c->view_evictions = c->view_fallbacks = 0;
    c->synthetic = false;
    c->sorted_amount = nullptr;
    c->sorted_row = nullptr;
    c->zone_var_valid = false;
    c->amount = nullptr;
    c->aos = nullptr;
    c->owns_table = true;
    c->staged = false;
    c->ids_dense = false;
    c->first_id = 0;
    c->head_cv = 0.0;
    c->n_global = c->shard_lo = c->n_local = 0;
    c->hbm_bytes = 0;
    c->table_epoch++;
}

int alloc_table(aqe_ctx* c, uint64_t n_local, bool keep_aos) {
    free_table(c);
    drop_cache(c);
    if (n_local == 0) return AQE_OK;
    // one spare double behind the column: the 16-byte dense loads park masked lanes on rows 0..1
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->amount), (n_local + 1) * sizeof(double)));
    c->hbm_bytes = (n_local + 1) * sizeof(double);
    c->dense16 = true;
    if (keep_aos) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->aos), n_local * sizeof(aqe_record)));
        c->hbm_bytes += n_local * sizeof(aqe_record);
    }
    return AQE_OK;
}

// GROUP BY needs the key column as SoA int32 on the device: from the resident 32-byte rows, or — for a table made
// by aqe_generate_synthetic — from the row number.  Built on first use, kept until the table changes.
int ensure_keys(aqe_ctx* c, int column) {
    const int k = column - 1;
    if (c->keycol[k] || c->n_local == 0) return AQE_OK;
    if (!c->aos && !c->synthetic)
        return fail(c, AQE_ERR_UNSUPPORTED, "grouped reduction needs the key columns: stage the table with AQE_STAGE_KEEP_AOS");
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->keycol[k]), c->n_local * sizeof(int32_t)));
    c->hbm_bytes += c->n_local * sizeof(int32_t);
    if (c->aos) HIPCHK(c, launch_extract_key(c->aos, c->keycol[k], c->n_local, column, c->stream));
    else HIPCHK(c, launch_synth_key(c->keycol[k], c->n_local, c->shard_lo, column, c->stream));
    int32_t* d_range = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_range), 2 * sizeof(int32_t)));
    int32_t init[2] = {std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::min()};
    hipError_t e = hipMemcpyAsync(d_range, init, sizeof init, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_key_range(c->keycol[k], c->n_local, d_range, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(init, d_range, sizeof init, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_range);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("key range: ") + hipGetErrorString(e));
    c->key_min[k] = init[0];
    c->key_max[k] = init[1];
    return AQE_OK;
}

// adaptive_block_sample's pre-pass (DB.cpp:1291-1308): population variance of each of the ten zones from raw
// moments, var = Q/n - (S/n)^2 — ten exact window scans on the device, kept until the table changes.
// (rows, sum, sum of squares) of the rows of each zone this context holds: an exact window scan per zone.
static int zone_moments(aqe_ctx* c, double* out30) {
    const uint64_t zone_size = c->n_global / 10;
    if (zone_size == 0) return fail(c, AQE_ERR_INVALID, "adaptive_block_sample: needs at least 10 rows");
    const uint64_t lo = c->shard_lo, hi = c->shard_lo + c->n_local;
    for (uint64_t z = 0; z < 10; ++z) {
        double* o = out30 + 3 * z;
        o[0] = o[1] = o[2] = 0.0;
        const uint64_t a = z * zone_size, b = std::min(a + zone_size, c->n_global);
        if (std::max(a, lo) >= std::min(b, hi)) continue;  // none of the zone's rows are here
        aqe_query q;
        aqe_query_defaults(&q);
        q.method = AQE_M_EXACT;
        q.sample_percent = 100.0;
        q.row_lo = a;
        q.row_hi = b;
        aqe_plan* p = nullptr;
        aqe_result r;
        int rc = cached_plan(c, &q, &p);
        if (rc == AQE_OK) rc = enqueue_all(p, c->stream, false);
        if (rc == AQE_OK) rc = fetch(p, &r, c->stream);
        if (rc != AQE_OK) return rc;
        o[0] = static_cast<double>(r.n);
        o[1] = r.sum;
        o[2] = r.sumsq;
    }
    return AQE_OK;
}

int ensure_zone_variances(aqe_ctx* c) {
    if (c->zone_var_valid) return AQE_OK;  // (of a sharded table: what its ranks agreed on, aqe_set_zone_variances)
    double m[30];
    int rc = zone_moments(c, m);
    if (rc != AQE_OK) return rc;
    for (int z = 0; z < 10; ++z) {
        const double cnt = m[3 * z], mean = m[3 * z + 1] / cnt;
        c->zone_var[z] = (m[3 * z + 2] / cnt) - (mean * mean);
    }
    c->zone_var_valid = true;
    return AQE_OK;
}

// stratified_block_sample's pre-pass (DB.cpp:1342-1345): the amount column sorted ascending + its row permutation.
int ensure_sorted(aqe_ctx* c) {
    if (c->sorted_amount || c->n_local == 0) return AQE_OK;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->sorted_amount), (c->n_local + 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->sorted_row), c->n_local * sizeof(uint32_t)));
    hipError_t e = sort_amounts(c->amount, c->n_local, c->sorted_amount, c->sorted_row, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(c->sorted_amount); (void)hipFree(c->sorted_row);
        c->sorted_amount = nullptr; c->sorted_row = nullptr;
        return fail(c, AQE_ERR_HIP, std::string("sorting the amount column: ") + hipGetErrorString(e));
    }
    c->hbm_bytes += (c->n_local + 1) * sizeof(double) + c->n_local * sizeof(uint32_t);
    return AQE_OK;
}

// A CLT pointer reads rows row0, row0 + s, row0 + 2 s, ...: in the column that is 8 bytes of every 8 s, and the
// memory system moves whole lines (at s = 5, with the fast and the slow pointer, 40 % of every line is wanted and
// 100 % is moved).  HBM capacity is the cheap resource on this part, so the column is kept a second time in
// stride-major order per step in use: the same rows, now contiguous — a dense stream, traffic = the sampled bytes.
namespace {

void free_view(aqe_ctx* c, StrideView& v) {
    (void)hipFree(v.amount);
    for (int32_t* k : v.keys)
        if (k) (void)hipFree(k);
    c->hbm_bytes -= v.bytes;
    c->view_bytes -= v.bytes;
}

// Makes room for one more view: the least recently used view that no plan of the caller is laid out over goes, and
// the reduce cache's plans that use it go with it.  False when every view is held by live plans.
bool evict_one_view(aqe_ctx* c) {
    auto victim = c->stride_views.end();
    for (auto it = c->stride_views.begin(); it != c->stride_views.end(); ++it)
        if (it->second.refs == it->second.cache_refs && (victim == c->stride_views.end() || it->second.last_use < victim->second.last_use)) victim = it;
    if (victim == c->stride_views.end()) return false;
    const uint64_t step = victim->first;
    (void)hipDeviceSynchronize();  // ONE wait for whatever still reads the view; the plans below then go without one each
    for (size_t i = 0; i < c->cache.size();) {
        aqe_plan* p = c->cache[i].second;
        if (p->view_step_rounds == step || p->view_step_topup == step) {
            destroy_plan(p, true);  // (releases its references)
            c->cache.erase(c->cache.begin() + static_cast<long>(i));
        } else {
            ++i;
        }
    }
    victim = c->stride_views.find(step);
    free_view(c, victim->second);
    c->stride_views.erase(victim);
    c->view_evictions++;
    return true;
}

}  // namespace

int ensure_stride_view(aqe_ctx* c, uint64_t step, const double** view, uint64_t* M_out, uint64_t* q0_out) {
    const uint64_t M = c->n_local / step + 2, q0 = c->shard_lo / step;
    *M_out = M;
    *q0_out = q0;
    auto it = c->stride_views.find(step);
    if (it != c->stride_views.end()) {
        it->second.last_use = ++c->view_clock;
        *view = it->second.amount;
        return AQE_OK;
    }
    *view = nullptr;
    StrideView v;
    v.M = M;
    v.q0 = q0;
    v.bytes = (static_cast<size_t>(step) * M + 2) * sizeof(double);  // + the spare rows the 16-byte loads park on
    // A table may hold kMaxStrideViews views whatever their size, and more while all of them fit the view budget (16 GiB;
    // AQE_VIEW_BUDGET_MB, read per call): HBM capacity is the cheap resource here — a 10 M-row table holds a view per
    // step anybody asks for and never falls back to sweeping in place, a 1 B-row table (8 GB per view) stops at eight.
    uint64_t budget = 16ull << 30;
    if (const char* e = std::getenv("AQE_VIEW_BUDGET_MB")) budget = std::strtoull(e, nullptr, 10) << 20;
    while (c->stride_views.size() >= kMaxStrideViews && c->view_bytes + v.bytes > budget) {
        if (!evict_one_view(c)) {
            c->view_fallbacks++;  // every view is held by a live plan: this step is swept in place
            return AQE_OK;
        }
    }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&v.amount), v.bytes));
    hipError_t e = hipMemsetAsync(v.amount, 0, v.bytes, c->stream);
    if (e == hipSuccess) e = launch_stride_view(c->amount, c->n_local, c->shard_lo, step, M, q0, v.amount, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(v.amount); return fail(c, AQE_ERR_HIP, std::string("stride-major view: ") + hipGetErrorString(e)); }
    v.last_use = ++c->view_clock;
    c->hbm_bytes += v.bytes;
    c->view_bytes += v.bytes;
    *view = v.amount;
    c->stride_views[step] = v;
    return AQE_OK;
}

void release_stride_view(aqe_ctx* c, uint64_t step, bool cached) {
    auto it = c->stride_views.find(step);
    if (it == c->stride_views.end()) return;  // (the table was replaced: its views went with it)
    if (it->second.refs) it->second.refs--;
    if (cached && it->second.cache_refs) it->second.cache_refs--;
}

// The key column of a GROUP BY in the slot order of an existing view: the sampled rows' keys are contiguous beside
// their amounts (12 bytes per sampled row instead of a 128-byte line of each column).
int ensure_key_view(aqe_ctx* c, int column, uint64_t step, const int32_t** view) {
    *view = nullptr;
    auto it = c->stride_views.find(step);
    if (it == c->stride_views.end()) return fail(c, AQE_ERR_INVALID, "internal: key view without its amount view");
    StrideView& v = it->second;
    const int k = column - 1;
    if (!v.keys[k]) {
        int rc = ensure_keys(c, column);
        if (rc != AQE_OK) return rc;
        const size_t bytes = (static_cast<size_t>(step) * v.M + 2) * sizeof(int32_t);
        int32_t* kv = nullptr;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&kv), bytes));
        hipError_t e = hipMemsetAsync(kv, 0, bytes, c->stream);
        if (e == hipSuccess) e = launch_stride_view_keys(c->keycol[k], c->n_local, c->shard_lo, step, v.M, v.q0, kv, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(kv); return fail(c, AQE_ERR_HIP, std::string("stride-major key view: ") + hipGetErrorString(e)); }
        v.keys[k] = kv;
        v.bytes += bytes;
        c->hbm_bytes += bytes;
        c->view_bytes += bytes;
    }
    v.last_use = ++c->view_clock;
    *view = v.keys[k];
    return AQE_OK;
}

namespace {

// Worker threads that fill the pinned bounce buffers: one host core reads rows at ~13 GB/s, a fifth of what the
// PCIe link takes, so the fill of every chunk is split over the host's cores (the loader).
class FillPool {
  public:
    explicit FillPool(unsigned n) {
        for (unsigned i = 0; i < n; ++i)
            workers_.emplace_back([this, i, n] {
                unsigned seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m_);
                    wake_.wait(lk, [&] { return stop_ || gen_ != seen; });
                    if (stop_) return;
                    seen = gen_;
                    lk.unlock();
                    job_(i, n);
                    lk.lock();
                    if (--pending_ == 0) done_.notify_one();
                }
            });
    }
    ~FillPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // runs f(part, parts) on every worker and returns when all are done
    void run(std::function<void(unsigned, unsigned)> f) {
        std::unique_lock<std::mutex> lk(m_);
        job_ = std::move(f);
        pending_ = static_cast<unsigned>(workers_.size());
        ++gen_;
        wake_.notify_all();
        done_.wait(lk, [&] { return pending_ == 0; });
    }

  private:
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    std::function<void(unsigned, unsigned)> job_;
    unsigned gen_ = 0, pending_ = 0;
    bool stop_ = false;
};

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void free_ring(aqe_ctx* c) {
    for (int b = 0; b < kStageRing; ++b) {
        if (c->ring.buf[b]) (void)hipHostFree(c->ring.buf[b]);
        if (c->ring.done[b]) (void)hipEventDestroy(c->ring.done[b]);
        c->ring.buf[b] = nullptr;
        c->ring.done[b] = nullptr;
    }
    c->ring.bytes_each = 0;
}

int ensure_ring(aqe_ctx* c, size_t bytes_each) {
    if (c->ring.bytes_each >= bytes_each) return AQE_OK;
    free_ring(c);
    for (int b = 0; b < kStageRing; ++b) {
        if (hipHostMalloc(&c->ring.buf[b], bytes_each, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->ring.done[b], hipEventDisableTiming) != hipSuccess) {
            free_ring(c);
            return fail(c, AQE_ERR_HIP, "staging: cannot allocate pinned bounce buffers");
        }
    }
    c->ring.bytes_each = bytes_each;
    return AQE_OK;
}

// Where the rows of a staging come from: host memory (aqe_stage_records) or a file in the reference's format, read with
// pread straight into the pinned ring — the kernel copies out of the page cache without a page fault per 4 KB page, which
// is what capped a copy out of a fresh mmap at 11 GB/s (a fault per page against ~0.1 us of copying).
struct RowSource {
    const aqe_record* rows = nullptr;  // memory
    int fd = -1;                       // else: file, row i of the staging at byte file_off + 32 i
    uint64_t file_off = 0;
};

bool read_fully(int fd, void* dst, size_t bytes, uint64_t off) {
    char* p = static_cast<char*>(dst);
    while (bytes) {
        const ssize_t got = pread(fd, p, bytes, static_cast<off_t>(off));
        if (got <= 0) return false;
        p += got; off += static_cast<uint64_t>(got); bytes -= static_cast<size_t>(got);
    }
    return true;
}

int stage_rows(aqe_ctx* c, const RowSource& src, uint64_t n_local, uint64_t shard_lo, uint64_t n_global, uint32_t flags, int64_t id0) {
    const auto t_all = std::chrono::steady_clock::now();
    aqe_stage_stats st{};
    if (shard_lo + n_local > n_global) return fail(c, AQE_ERR_INVALID, "shard exceeds the table");
    const bool keep = flags & AQE_STAGE_KEEP_AOS;
    auto t0 = std::chrono::steady_clock::now();
    int rc = alloc_table(c, n_local, keep);
    if (rc != AQE_OK) return rc;
    st.device_alloc_ms = ms_since(t0);
    c->n_global = n_global;
    c->shard_lo = shard_lo;
    c->n_local = n_local;
    c->staged = true;
    c->stage_stats = st;
    if (n_local == 0) return AQE_OK;
    // A ring of pinned buffers: host threads fill buffer b while the copy engine drains the ones before it.
    // Without KEEP_AOS only the amount column crosses PCIe (8 of every 32 bytes).
    const size_t row_bytes = keep ? sizeof(aqe_record) : sizeof(double);
    // (a table of up to 1 GiB of rows takes half-size chunks: pinning the ring — ~0.19 ms per MiB — is then a quarter of the load, not half)
    const uint64_t chunk_rows = std::min<uint64_t>(n_local <= (1ull << 25) ? kStageChunkRows / 2 : kStageChunkRows, n_local);
    t0 = std::chrono::steady_clock::now();
    const bool had_ring = c->ring.bytes_each >= chunk_rows * row_bytes;
    rc = ensure_ring(c, chunk_rows * row_bytes);
    if (rc != AQE_OK) { free_table(c); return rc; }
    st.pinned_alloc_ms = had_ring ? 0.0 : ms_since(t0);
    int status = AQE_OK;
    std::atomic<bool> dense_ids{true}, io_ok{true};
    const unsigned hw = std::thread::hardware_concurrency();
    FillPool pool(n_local < (1u << 18) ? 1u : std::max(1u, std::min(16u, hw ? hw : 4u)));
    st.fill_threads = n_local < (1u << 18) ? 1u : std::max(1u, std::min(16u, hw ? hw : 4u));
    for (uint64_t off = 0, k = 0; off < n_local && status == AQE_OK; off += chunk_rows, ++k) {
        const int b = static_cast<int>(k % kStageRing);
        const uint64_t m = std::min<uint64_t>(chunk_rows, n_local - off);
        if (k >= static_cast<uint64_t>(kStageRing)) {
            t0 = std::chrono::steady_clock::now();
            if (hipEventSynchronize(c->ring.done[b]) != hipSuccess) { status = fail(c, AQE_ERR_HIP, "event sync"); break; }
            st.wait_ms += ms_since(t0);
        }
        hipError_t e;
        void* const dst_buf = c->ring.buf[b];
        t0 = std::chrono::steady_clock::now();
        pool.run([&, dst_buf](unsigned part, unsigned parts) {  // rows [lo, hi) of the chunk: fill + dense-id check
            const uint64_t lo = m * part / parts, hi = m * (part + 1) / parts;
            bool dense = true;
            if (src.rows) {
                const aqe_record* rows = src.rows;
                for (uint64_t i = lo; i < hi; ++i) dense = dense && rows[off + i].id == id0 + static_cast<int64_t>(off + i);
                if (keep) {
                    std::memcpy(static_cast<aqe_record*>(dst_buf) + lo, rows + off + lo, (hi - lo) * sizeof(aqe_record));
                } else {
                    double* dst = static_cast<double*>(dst_buf);
                    for (uint64_t i = lo; i < hi; ++i) dst[i] = rows[off + i].amount;
                }
            } else if (keep) {
                aqe_record* dst = static_cast<aqe_record*>(dst_buf) + lo;
                if (!read_fully(src.fd, dst, (hi - lo) * sizeof(aqe_record), src.file_off + (off + lo) * sizeof(aqe_record))) { io_ok.store(false); return; }
                for (uint64_t i = lo; i < hi; ++i) dense = dense && dst[i - lo].id == id0 + static_cast<int64_t>(off + i);
            } else {
                constexpr uint64_t kPiece = 4096;  // rows: 128 KiB of file through a cache-resident buffer, the amounts picked out of it
                std::vector<aqe_record> tmp(std::min<uint64_t>(kPiece, hi - lo));
                double* dst = static_cast<double*>(dst_buf);
                for (uint64_t i = lo; i < hi; i += kPiece) {
                    const uint64_t cnt = std::min<uint64_t>(kPiece, hi - i);
                    if (!read_fully(src.fd, tmp.data(), cnt * sizeof(aqe_record), src.file_off + (off + i) * sizeof(aqe_record))) { io_ok.store(false); return; }
                    for (uint64_t j = 0; j < cnt; ++j) {
                        dense = dense && tmp[j].id == id0 + static_cast<int64_t>(off + i + j);
                        dst[i + j] = tmp[j].amount;
                    }
                }
            }
            if (!dense) dense_ids.store(false, std::memory_order_relaxed);
        });
        st.fill_ms += ms_since(t0);
        if (!io_ok.load()) { status = fail(c, AQE_ERR_IO, "staging: short read from the database file"); break; }
        if (keep) {
            e = hipMemcpyAsync(c->aos + off, dst_buf, m * sizeof(aqe_record), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = launch_split_amount(c->aos + off, c->amount + off, m, c->stream);
        } else {
            e = hipMemcpyAsync(c->amount + off, dst_buf, m * sizeof(double), hipMemcpyHostToDevice, c->stream);
        }
        if (e == hipSuccess) e = hipEventRecord(c->ring.done[b], c->stream);
        if (e != hipSuccess) status = fail(c, AQE_ERR_HIP, std::string("staging: ") + hipGetErrorString(e));
        st.chunks++;
    }
    t0 = std::chrono::steady_clock::now();
    hipError_t e = hipStreamSynchronize(c->stream);
    st.wait_ms += ms_since(t0);
    if (e != hipSuccess && status == AQE_OK) status = fail(c, AQE_ERR_HIP, std::string("staging sync: ") + hipGetErrorString(e));
    if (status != AQE_OK) { free_table(c); return status; }
    c->ids_dense = dense_ids.load();
    c->first_id = id0 - static_cast<int64_t>(shard_lo);  // id of global row 0 when the ids are dense
    st.host_bytes = n_local * sizeof(aqe_record);
    st.link_bytes = n_local * row_bytes;
    st.total_ms = ms_since(t_all);
    c->stage_stats = st;
    return status;
}

int stage_from_host(aqe_ctx* c, const aqe_record* rows, uint64_t n_local, uint64_t shard_lo, uint64_t n_global,
                    uint32_t flags) {
    if (n_local && !rows) return fail(c, AQE_ERR_INVALID, "null rows");
    RowSource src;
    src.rows = rows;
    int rc = stage_rows(c, src, n_local, shard_lo, n_global, flags, n_local ? rows[0].id : 0);
    if (rc != AQE_OK) return rc;
    c->shift = shift_of_rows(rows, n_local);  // shards other than the first are given the table's value (aqe_set_shift)
    c->head_cv = cv_of_rows(rows, n_local);
    return AQE_OK;
}

struct MappedFile {
    void* base = MAP_FAILED;
    size_t bytes = 0;
    int fd = -1;
    ~MappedFile() {
        if (base != MAP_FAILED) munmap(base, bytes);
        if (fd >= 0) close(fd);
    }
};

int open_db_file(aqe_ctx* c, const char* path, MappedFile& mf, uint64_t& count) {
    mf.fd = open(path, O_RDONLY);
    if (mf.fd < 0) return fail(c, AQE_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(mf.fd, &st) != 0 || st.st_size < 24) return fail(c, AQE_ERR_IO, std::string("not an aqe database file: ") + path);
    mf.bytes = static_cast<size_t>(st.st_size);
    mf.base = mmap(nullptr, mf.bytes, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if (mf.base == MAP_FAILED) return fail(c, AQE_ERR_IO, std::string("mmap failed: ") + path);
    uint64_t hdr[3];  // size_t total | size_t height | size_t count, DB.cpp:669-676
    std::memcpy(hdr, mf.base, sizeof hdr);
    count = hdr[2];
    // (the count is untrusted: compared without multiplying, a header of 2^59 rows must not wrap past the check)
    if (count > (mf.bytes - 24) / sizeof(aqe_record)) return fail(c, AQE_ERR_IO, std::string("truncated database file: ") + path);
    return AQE_OK;
}

}  // namespace
}  // namespace aqe

extern "C" {

int aqe_create(int device_id, aqe_ctx** out) {
    if (!out) return fail(nullptr, AQE_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, AQE_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (device_id < 0 || device_id >= n) return fail(nullptr, AQE_ERR_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, AQE_ERR_NO_DEVICE, "cannot query device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, AQE_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    std::unique_ptr<aqe_ctx> c(new aqe_ctx());
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, AQE_ERR_HIP, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(nullptr, AQE_ERR_HIP, "stream creation failed");
    // one 16-wave workgroup per CU, rounded down to a power of two (the wave->tile map uses masks)
    c->persist_grid = 16;
    while (c->persist_grid * 2 <= static_cast<unsigned>(prop.multiProcessorCount) && c->persist_grid * 2 <= kMaxPersistGrid) c->persist_grid *= 2;
    if (std::getenv("AQE_PERSIST_STAMPS") &&
        hipMalloc(reinterpret_cast<void**>(&c->d_stamps), 8 * (8 * static_cast<size_t>(c->persist_grid) * kPersistWaves + 8 * kMaxPersistRounds)) != hipSuccess)
        return fail(nullptr, AQE_ERR_HIP, "stamp buffer allocation failed");
    *out = c.release();
    return AQE_OK;
}

void aqe_destroy(aqe_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_cache(c);
    for (const PlanScratch& sc : c->scratch_pool) {
        (void)hipFree(sc.partials);
        (void)hipFree(sc.counter);
        (void)hipFree(sc.d_state);
        (void)hipHostFree(sc.h_result);
        (void)hipEventDestroy(sc.ev0);
        (void)hipEventDestroy(sc.ev1);
        (void)hipFree(sc.d_ctl);
        (void)hipFree(sc.d_rehearsal);
        (void)hipFree(sc.d_fams_small);
    }
    c->scratch_pool.clear();
    free_table(c);
    free_ring(c);
    if (c->d_stamps) (void)hipFree(c->d_stamps);
    if (c->grp_partial) (void)hipFree(c->grp_partial);
    if (c->grp_out_host) (void)hipHostFree(c->grp_out_host);
    if (c->grp_check_host) (void)hipHostFree(c->grp_check_host);
    if (c->grp_acc) (void)hipFree(c->grp_acc);
    if (c->grp_ticket) (void)hipFree(c->grp_ticket);
    for (hipStream_t s : c->lanes) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int aqe_stage_records(aqe_ctx* c, const void* aos32, uint64_t n_local, uint64_t shard_lo, uint64_t n_global, uint32_t flags) {
    if (!c) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    return stage_from_host(c, static_cast<const aqe_record*>(aos32), n_local, shard_lo, n_global, flags);
}

int aqe_last_stage_stats(const aqe_ctx* c, aqe_stage_stats* out) {
    if (!c || !out) return AQE_ERR_INVALID;
    *out = c->stage_stats;
    return AQE_OK;
}

int aqe_file_rows(const char* path, uint64_t* n_rows) {
    if (!path || !n_rows) return AQE_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(nullptr, AQE_ERR_IO, std::string("cannot open ") + path);
    uint64_t hdr[3];
    size_t got = std::fread(hdr, sizeof hdr, 1, f);
    std::fclose(f);
    if (got != 1) return fail(nullptr, AQE_ERR_IO, std::string("not an aqe database file: ") + path);
    *n_rows = hdr[2];
    return AQE_OK;
}

int aqe_stage_file(aqe_ctx* c, const char* path, uint64_t shard_lo, uint64_t n_local, uint32_t flags) {
    if (!c || !path) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    MappedFile mf;
    uint64_t count = 0;
    int rc = open_db_file(c, path, mf, count);
    if (rc != AQE_OK) return rc;
    if (shard_lo > count) return fail(c, AQE_ERR_INVALID, "shard_lo beyond the end of the file");
    if (n_local == 0) n_local = count - shard_lo;
    if (shard_lo + n_local > count) return fail(c, AQE_ERR_INVALID, "shard exceeds the file");
    // (the header and the table's head are read through the mapping; the rows themselves with pread into the pinned ring)
    const aqe_record* rows = reinterpret_cast<const aqe_record*>(static_cast<const char*>(mf.base) + 24);
    RowSource src;
    src.fd = mf.fd;
    src.file_off = 24 + shard_lo * sizeof(aqe_record);
    rc = stage_rows(c, src, n_local, shard_lo, count, flags, n_local ? rows[shard_lo].id : 0);
    if (rc == AQE_OK) {  // from the table's head: identical on every shard
        c->shift = shift_of_rows(rows, count);
        c->head_cv = cv_of_rows(rows, count);
    }
    return rc;
}

int aqe_save_file(aqe_ctx* c, const char* path) {
    if (!c || !path) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_local && !c->aos) return fail(c, AQE_ERR_UNSUPPORTED, "save needs the rows resident (AQE_STAGE_KEEP_AOS)");
    if (c->shard_lo != 0 || c->n_local != c->n_global) return fail(c, AQE_ERR_UNSUPPORTED, "save needs the whole table in this context");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(c, AQE_ERR_IO, std::string("cannot create ") + path);
    uint64_t height = 1;  // informational: the reference rebuilds its tree on load (DB.cpp:703-710)
    for (uint64_t cap = 254; c->n_global > cap; cap *= 128) ++height;
    uint64_t hdr[3] = {c->n_global, height, c->n_global};
    bool ok = std::fwrite(hdr, sizeof hdr, 1, f) == 1;
    std::vector<aqe_record> buf(std::min<uint64_t>(kStageChunkRows, std::max<uint64_t>(c->n_local, 1)));
    for (uint64_t off = 0; ok && off < c->n_local; off += buf.size()) {
        uint64_t m = std::min<uint64_t>(buf.size(), c->n_local - off);
        if (hipMemcpy(buf.data(), c->aos + off, m * sizeof(aqe_record), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = std::fwrite(buf.data(), sizeof(aqe_record), m, f) == m;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? AQE_OK : fail(c, AQE_ERR_IO, std::string("write failed: ") + path);
}

int aqe_generate_synthetic(aqe_ctx* c, uint64_t n_local, uint64_t shard_lo, uint64_t n_global, uint64_t seed, uint32_t flags) {
    if (!c) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (shard_lo + n_local > n_global) return fail(c, AQE_ERR_INVALID, "shard exceeds the table");
    int rc = alloc_table(c, n_local, flags & AQE_STAGE_KEEP_AOS);
    if (rc != AQE_OK) return rc;
    c->n_global = n_global;
    c->shard_lo = shard_lo;
    c->n_local = n_local;
    c->staged = true;
    {   // shift from the table's first rows, evaluated with the expression the kernel uses (every shard agrees)
        const uint64_t m = std::min<uint64_t>(n_global, kShiftRows);
        double acc = 0.0;
        std::vector<double> head(m);
        for (uint64_t i = 0; i < m; ++i) {
            uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            head[i] = 1.0 + 999.0 * (static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0));
            acc += head[i];
        }
        c->shift = m ? acc / static_cast<double>(m) : 0.0;
        c->head_cv = cv_of_values(head.data(), m);
    }
    c->ids_dense = true;  // id = row + 1
    c->first_id = 1;
    c->synthetic = true;
    HIPCHK(c, launch_synth(c->aos, c->amount, n_local, shard_lo, seed, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AQE_OK;
}

int aqe_attach_device(aqe_ctx* c, const double* dev_amount, const void* dev_aos32, uint64_t n_local, uint64_t shard_lo,
                      uint64_t n_global, double shift) {
    if (!c) return AQE_ERR_INVALID;
    if (n_local && !dev_amount) return fail(c, AQE_ERR_INVALID, "null amount column");
    if (shard_lo + n_local > n_global) return fail(c, AQE_ERR_INVALID, "shard exceeds the table");
    free_table(c);
    drop_cache(c);
    c->owns_table = false;
    c->dense16 = n_local >= 2;  // caller-owned memory has no spare row
    c->staged = true;
    c->amount = const_cast<double*>(dev_amount);
    c->aos = static_cast<aqe_record*>(const_cast<void*>(dev_aos32));
    c->n_local = n_local;
    c->shard_lo = shard_lo;
    c->n_global = n_global;
    c->shift = shift;
    return AQE_OK;
}

int aqe_set_shift(aqe_ctx* c, double shift) {
    if (!c) return AQE_ERR_INVALID;
    c->shift = shift;
    drop_cache(c);
    return AQE_OK;
}

int aqe_table_info_get(const aqe_ctx* c, aqe_table_info* out) {
    if (!c || !out) return AQE_ERR_INVALID;
    out->global_rows = c->n_global;
    out->shard_lo = c->shard_lo;
    out->local_rows = c->n_local;
    out->shift = c->shift;
    out->has_aos = c->aos != nullptr;
    out->device_id = c->device;
    out->hbm_bytes = c->hbm_bytes;
    out->view_bytes = c->view_bytes;
    out->n_views = static_cast<uint32_t>(c->stride_views.size());
    out->view_evictions = static_cast<uint32_t>(c->view_evictions);
    out->view_fallbacks = static_cast<uint32_t>(c->view_fallbacks);
    out->reserved = 0;
    return AQE_OK;
}

// ---- the variance-aware samplers over a sharded table (include/aqe_hip.h) ----
int aqe_zone_moments(aqe_ctx* c, double* out30) {
    if (!c || !out30) return AQE_ERR_INVALID;
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    HIPCHK(c, hipSetDevice(c->device));
    return zone_moments(c, out30);
}

int aqe_set_zone_variances(aqe_ctx* c, const double* var10) {
    if (!c || !var10) return AQE_ERR_INVALID;
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    for (int z = 0; z < 10; ++z)
        if (!(var10[z] == var10[z])) return fail(c, AQE_ERR_INVALID, "zone variance is NaN");
    HIPCHK(c, hipSetDevice(c->device));
    // plans made from other variances must not be served again
    bool changed = !c->zone_var_valid;
    for (int z = 0; z < 10 && !changed; ++z) changed = c->zone_var[z] != var10[z];
    if (changed) {
        bool any = false;
        for (auto& kv : c->cache) any = any || kv.second->q.method == AQE_M_ADAPTIVE_BLOCK;
        if (any) {
            (void)hipDeviceSynchronize();
            for (size_t i = 0; i < c->cache.size();) {
                if (c->cache[i].second->q.method == AQE_M_ADAPTIVE_BLOCK) {
                    destroy_plan(c->cache[i].second, true);
                    c->cache.erase(c->cache.begin() + static_cast<long>(i));
                } else {
                    ++i;
                }
            }
        }
    }
    std::memcpy(c->zone_var, var10, sizeof c->zone_var);
    c->zone_var_valid = true;
    return AQE_OK;
}

int aqe_sorted_counts(aqe_ctx* c, const double* values, uint32_t n, uint64_t* n_less, uint64_t* n_less_equal) {
    if (!c || (n && (!values || !n_less || !n_less_equal))) return AQE_ERR_INVALID;
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_sorted(c);
    if (rc != AQE_OK) return rc;
    hipError_t e = sorted_counts(c->sorted_amount, c->n_local, values, n, n_less, n_less_equal, c->stream);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("aqe_sorted_counts: ") + hipGetErrorString(e));
    return AQE_OK;
}

int aqe_key_range_rows(aqe_ctx* c, int64_t id_min, int64_t id_max, uint64_t* row_lo, uint64_t* row_hi) {
    if (!c || !row_lo || !row_hi) return AQE_ERR_INVALID;
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    HIPCHK(c, hipSetDevice(c->device));
    *row_lo = *row_hi = 0;
    if (id_max < id_min || c->n_global == 0) return AQE_OK;
    if (c->ids_dense) {  // id = first_id + row
        const int64_t last = c->first_id + static_cast<int64_t>(c->n_global) - 1;
        if (id_max < c->first_id || id_min > last) return AQE_OK;
        *row_lo = static_cast<uint64_t>(std::max(id_min, c->first_id) - c->first_id);
        *row_hi = static_cast<uint64_t>(std::min(id_max, last) - c->first_id) + 1;
        return AQE_OK;
    }
    if (c->shard_lo != 0 || c->n_local != c->n_global) return fail(c, AQE_ERR_UNSUPPORTED, "key bounds on a shard need dense ids");
    if (!c->aos) return fail(c, AQE_ERR_UNSUPPORTED, "key bounds on non-dense ids need the rows resident (AQE_STAGE_KEEP_AOS)");
    uint64_t* d_out = nullptr;
    uint64_t h_out[2] = {0, 0};
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_out), sizeof h_out));
    hipError_t e = launch_id_bounds(c->aos, c->n_local, id_min, id_max, d_out, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("key bounds: ") + hipGetErrorString(e));
    *row_lo = h_out[0];
    *row_hi = std::max(h_out[0], h_out[1]);
    return AQE_OK;
}

// Key bounds of a SHARD: how many of its rows come before id_min / up to id_max.  Ids ascend in leaf order over the whole table,
// so the counts of all shards add up to the global row window [sum n_below, sum n_upto) — one all-reduce SUM of two numbers.
int aqe_key_range_counts(aqe_ctx* c, int64_t id_min, int64_t id_max, uint64_t* n_below, uint64_t* n_upto) {
    if (!c || !n_below || !n_upto) return AQE_ERR_INVALID;
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    HIPCHK(c, hipSetDevice(c->device));
    *n_below = *n_upto = 0;
    if (c->n_local == 0) return AQE_OK;
    if (c->ids_dense) {  // id = first_id + global row: rows below x are the global rows [0, x - first_id), clipped to the shard
        auto rows_below = [&](int64_t x) -> uint64_t {  // local rows with id < x
            if (x <= c->first_id) return 0;
            const unsigned __int128 g = static_cast<unsigned __int128>(static_cast<__int128>(x) - static_cast<__int128>(c->first_id));
            const uint64_t hi = c->shard_lo + c->n_local;
            const uint64_t gg = g > static_cast<unsigned __int128>(hi) ? hi : static_cast<uint64_t>(g);
            return gg <= c->shard_lo ? 0 : gg - c->shard_lo;
        };
        *n_below = rows_below(id_min);
        *n_upto = id_max == INT64_MAX ? c->n_local : rows_below(id_max + 1);
        if (*n_upto < *n_below) *n_upto = *n_below;
        return AQE_OK;
    }
    if (!c->aos) return fail(c, AQE_ERR_UNSUPPORTED, "key bounds on non-dense ids need the rows resident (AQE_STAGE_KEEP_AOS)");
    uint64_t* d_out = nullptr;
    uint64_t h_out[2] = {0, 0};
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_out), sizeof h_out));
    hipError_t e = launch_id_bounds(c->aos, c->n_local, id_min, id_max, d_out, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("key bounds: ") + hipGetErrorString(e));
    *n_below = h_out[0];
    *n_upto = std::max(h_out[0], h_out[1]);
    return AQE_OK;
}

int aqe_release_table(aqe_ctx* c) {
    if (!c) return AQE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_cache(c);
    free_table(c);
    return AQE_OK;
}

}  // extern "C"

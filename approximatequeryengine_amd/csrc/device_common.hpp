// device_common.hpp — device-side building blocks shared by kernels.hip and persist.hip: wave64 reduction,
// the Welford fold with the CLT rules, the estimator, and the tile sweep.
//
// Reference lines restated here (DB.cpp = /root/reference/src/aqe_backend/core/custom_bplus_db.cpp):
//   reducer loops       DB.cpp:285-294, 324-335, 2024-2036  -> sweep_tile
//   CLT error rule      DB.cpp:936-961                      -> fold(), rule A on the leader's own triple (group a)
//   CLT cross-check     DB.cpp:993-1016                     -> fold(), rule B
//   top-up              DB.cpp:1031-1040                    -> fold() with FoldParams::is_topup
//   estimators + CI     enhanced_aqe_cli.py:189-200, 277-291; DB.cpp:303-315 -> finalize()
#pragma once

#include "kernels.hpp"

namespace aqe {
namespace {

typedef unsigned long long u64;

// kernel arguments read through the kernarg segment pointer (constant address space): dynamic indexing stays in
// scalar loads instead of making hipcc copy the by-value parameter's arrays to scratch
#define AQE_KARG __attribute__((address_space(4)))

__device__ __forceinline__ u64 uniform64(u64 x) {
    unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(x));
    unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(x >> 32));
    return (static_cast<u64>(hi) << 32) | lo;
}

// Sum SEVEN per-lane values over the wave with 10 exchanges instead of 42: a butterfly that halves the
// number of components a lane carries at each of the first three steps (lane bits 5, 4, 3) and then finishes
// the one remaining component over the low three lane bits.  On return lane L holds the wave total of
// component (L >> 3) for (L >> 3) < 7 — in all eight lanes of that class.  Fixed exchange pattern: bitwise
// reproducible.  Every exchange is a VALU cross-lane move (gfx950 v_permlane32_swap / v_permlane16_swap, DPP
// row_ror / quad_perm / row_half_mirror): no LDS round trips — this sits on the critical path of every
// hand-off.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// lanes < 32: a[L] + a[L + 32];  lanes >= 32: b[L - 32] + b[L]
__device__ __forceinline__ double swap_add32(double a, double b) {
    const u32x2 lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2loint(a)), static_cast<unsigned>(__double2loint(b)), false, false);
    const u32x2 hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2hiint(a)), static_cast<unsigned>(__double2hiint(b)), false, false);
    return __hiloint2double(static_cast<int>(hi.x), static_cast<int>(lo.x)) + __hiloint2double(static_cast<int>(hi.y), static_cast<int>(lo.y));
}
// even rows of 16 lanes: a[L] + a[L + 16];  odd rows: b[L - 16] + b[L]
__device__ __forceinline__ double swap_add16(double a, double b) {
    const u32x2 lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(a)), static_cast<unsigned>(__double2loint(b)), false, false);
    const u32x2 hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(a)), static_cast<unsigned>(__double2hiint(b)), false, false);
    return __hiloint2double(static_cast<int>(hi.x), static_cast<int>(lo.x)) + __hiloint2double(static_cast<int>(hi.y), static_cast<int>(lo.y));
}
template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum7(const double (&v)[7], int lane) {
    // bit 5: lanes < 32 go on with components 0..3, lanes >= 32 with 4..7 (component 7 is a zero pad)
    const double u0 = swap_add32(v[0], v[4]), u1 = swap_add32(v[1], v[5]), u2 = swap_add32(v[2], v[6]), u3 = swap_add32(v[3], 0.0);
    // bit 4: even rows go on with the lower two of their four, odd rows with the upper two
    const double w0 = swap_add16(u0, u2), w1 = swap_add16(u1, u3);  // rows: components {0,2,4,6} and {1,3,5,7}
    // bit 3: each half row keeps one component and receives its partner's share of it
    const bool b3 = (lane & 8) != 0;
    double k = b3 ? w1 : w0;
    const double s = b3 ? w0 : w1;
    k += dpp_f64<0x128>(s);  // row_ror:8 = lane ^ 8 within a row of 16
    // bits 1, 0, 2 within the half row
    k += dpp_f64<0xB1>(k);   // quad_perm [1,0,3,2] = lane ^ 1
    k += dpp_f64<0x4E>(k);   // quad_perm [2,3,0,1] = lane ^ 2
    k += dpp_f64<0x141>(k);  // row_half_mirror: the other quad of the half row (all four of its lanes agree)
    return k;
}

// value of a wave-uniform source lane, through the scalar path (v_readlane) instead of an LDS permute
__device__ __forceinline__ double read_lane_f64(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// One lane of a wave has written LDS words that the other lanes of the SAME wave read next (a result parked for a
// wave-wide store).  The hardware keeps a wave's LDS accesses in order, but the compiler orders each thread on its own:
// without this it is free to run the readers' side of the branch before the writer's (it did, in k_round).  The
// convergent wave barrier pins the join; the wavefront-scope fences pin the accesses to their side of it.
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Acc {
    double sa = 0.0, qa = 0.0, sb = 0.0, qb = 0.0;
    unsigned na = 0, nb = 0, nv = 0;
};

// mean and centred second moment of a group from its shifted sums (n, S_d = sum(x-c), Q_d = sum (x-c)^2):
// mean = c + S_d/n, M2 = Q_d - S_d^2/n.  With c inside the data's range the subtraction is benign; this is
// the shifted-data form of the running mean/variance (Chan, Golub & LeVeque 1983), and unlike Welford's
// recurrence its partial states merge by plain addition — which is what lets workgroups, rounds and GPUs
// (one all-reduce SUM) combine in any grouping.
__device__ __forceinline__ void mean_m2(double n, double sd, double qd, double c, double& mean, double& m2) {
    mean = c + sd / n;
    m2 = qd - sd * sd / n;
    if (m2 < 0.0) m2 = 0.0;
}

// The monitor's decision on the moments gathered so far.  Group a is the LEADER — fast worker 0 — group b every other
// worker.  1 = error rule on the leader's OWN samples (DB.cpp:936-961: a fast thread judges the samples it took itself),
// 2 = cross-validation of the others' mean against the leader's, once the leader holds base/2 rows (DB.cpp:993-1016:
// the gate is the publishing fast thread's own count), 0 = go on.  The reference's own expressions:
__device__ __forceinline__ int clt_rules_exact(double n_a, double sd_a, double qd_a, double n_b, double sd_b, double qd_b,
                                               const FoldParams& p) {
    const double n = n_a;
    if (n >= 30.0) {
        double mean, m2;
        mean_m2(n, sd_a, qd_a, p.shift, mean, m2);
        const double var = m2 / (n - 1.0);
        const double se = sqrt(var / n);
        const double err = (p.z * se / mean) * 100.0;
        if (err <= p.e && n >= 50.0) return 1;
    }
    if (n_b >= 20.0 && n_a >= 30.0) {
        const double mean_a = p.shift + sd_a / n_a, mean_b = p.shift + sd_b / n_b;
        if (mean_a > 0.0) {
            const double diff = fabs(mean_b - mean_a) / mean_a;
            if (diff <= p.e / 100.0 && n_a >= static_cast<double>(p.base / 2)) return 2;
        }
    }
    return 0;
}

// The decision sits on the critical path of every query (the monitor wave's tail), and the expressions above are a
// chain of four f64 divisions and a square root — about 0.6 us on this part.  Both rules are comparisons, so they
// are first evaluated cleared of divisions and roots:
//   rule A   100 z sqrt(m2 / ((n-1) n)) / mean <= e   <=>   1e4 z^2 Q <= e^2 M^2 (n-1),   M = n mean = n c + Sd,  Q = n m2 = n Qd - Sd^2
//   rule B   |mean_b - mean_a| / mean_a <= e/100      <=>   |M_b n_a - M_a n_b| <= (e/100) M_a n_b
// (valid for mean > 0, e >= 0: both sides non-negative).  Only when a comparison falls within a guard band of its
// threshold — far wider than the rounding of either side — or outside that domain do the reference's expressions
// decide; the decision is therefore the same one, a dozen multiplications away instead of five divisions.
__device__ __forceinline__ int clt_rules(double n_a, double sd_a, double qd_a, double n_b, double sd_b, double qd_b,
                                         const FoldParams& p) {
    const double n = n_a, sd = sd_a, qd = qd_a, c = p.shift;  // rule A: the leader's own triple
    int a = 0, b = 0;  // 1/2: holds, 0: does not, -1: too close to call
    if (n >= 30.0) {
        const double M = n * c + sd, Q = qd * n - sd * sd;
        if (M > 0.0 && Q >= 0.0 && p.e >= 0.0) {
            const double zz = 1e4 * p.z * p.z;
            const double A = zz * Q, B = p.e * p.e * M * M * (n - 1.0);
            const double tol = 1e-6 * fmax(A, B) + zz * 1e-12 * (qd * n);
            a = A < B - tol ? (n >= 50.0 ? 1 : 0) : A > B + tol ? 0 : -1;
        } else {
            a = -1;
        }
    }
    if (a == 1) return 1;
    if (n_b >= 20.0 && n_a >= 30.0 && n_a >= static_cast<double>(p.base / 2)) {
        const double Ma = n_a * c + sd_a, Mb = n_b * c + sd_b;
        if (Ma > 0.0 && p.e >= 0.0) {
            const double L = fabs(Mb * n_a - Ma * n_b), R = (p.e / 100.0) * Ma * n_b;
            const double tol = 1e-6 * fmax(L, R) + 1e-12 * (fabs(Mb) * n_a + Ma * n_b);
            b = L < R - tol ? 2 : L > R + tol ? 0 : -1;
        } else {
            b = -1;
        }
    }
    if (a < 0 || b < 0) return clt_rules_exact(n_a, sd_a, qd_a, n_b, sd_b, qd_b, p);
    return b;
}

// Fold one launch's reduced vector into the query state and take the CLT decision.
__device__ __forceinline__ void fold(QueryState& s, const double (&vec)[kVec], const FoldParams& p) {
    if (p.is_topup) {  // DB.cpp:1031-1040: systematic rows appended to the sample
        s.n_p += vec[0]; s.sd_p += vec[1]; s.qd_p += vec[2];
        s.topup += vec[0];
        s.visited += vec[6];
        return;
    }
    s.n_a += vec[0]; s.sd_a += vec[1]; s.qd_a += vec[2];
    s.n_b += vec[3]; s.sd_b += vec[4]; s.qd_b += vec[5];
    s.n_p += vec[0] + vec[3]; s.sd_p += vec[1] + vec[4]; s.qd_p += vec[2] + vec[5];
    s.visited += vec[6];
    s.rounds += 1;
    if (!p.is_clt) return;
    const int code = clt_rules(s.n_a, s.sd_a, s.qd_a, s.n_b, s.sd_b, s.qd_b, p);
    if (code) {
        s.converged = code;
        s.stop = 1;
    }
}

// Estimate + interval from the folded state (CLI:189-200, 277-291; DB.cpp:303-315).
__device__ __forceinline__ aqe_result make_result(const QueryState& s, const FinalizeParams& p) {
    aqe_result r;
    const double n = s.n_p, visited = s.visited, c = p.shift;
    double mean = 0.0, m2 = 0.0;
    if (n > 0.0) mean_m2(n, s.sd_p, s.qd_p, c, mean, m2);
    const double N = static_cast<double>(p.n_global);
    const double S = s.sd_p + n * c;
    r.sum = S;
    r.sumsq = s.qd_p + 2.0 * c * s.sd_p + n * c * c;
    r.mean = mean;
    r.m2 = m2;
    r.n = static_cast<uint64_t>(n);
    r.visited = static_cast<uint64_t>(visited);
    r.topup = static_cast<uint64_t>(s.topup);
    r.converged = s.converged;
    r.rounds = s.rounds;
    r.kernel_ms = 0.0;
    r.bytes_algorithmic = r.visited * 8ull;
    r.device_status = s.error;
    r.topup_pending = 0;

    double moe = 0.0;  // CLI:279-282: two-pass variance, fixed 1.96
    if (n > 1.0) moe = 1.96 * sqrt(m2 / ((n - 1.0) * n));  // = 1.96 sqrt(var) / sqrt(n), one division and one root shorter
    double value = 0.0, margin = 0.0;
    if (p.is_exact) {  // DB.cpp:242-274
        value = p.agg == AQE_SUM ? S : p.agg == AQE_AVG ? (N > 0.0 ? S / N : 0.0) : (visited > n ? n : N);
    } else if (p.convention == AQE_EST_CLI) {  // CLI:189-200; interval CLI:284-291
        const double scale = visited > 0.0 ? N / visited : 0.0;
        if (p.agg == AQE_SUM) { value = S * scale; margin = moe * scale; }
        else if (p.agg == AQE_COUNT) { value = visited > n ? n * scale : (visited > 0.0 ? N : 0.0); }
        else { value = n > 0.0 ? S / n : 0.0; margin = moe; }
    } else if (p.convention == AQE_EST_CPP) {  // DB.cpp:303-315; interval scaled as executor.cpp:192-197
        const double scale = 100.0 / p.pct;
        if (p.agg == AQE_SUM) { value = S * scale; margin = moe * scale; }
        else if (p.agg == AQE_AVG) { value = N > 0.0 ? S * scale / N : 0.0; margin = moe; }
        else { value = static_cast<double>(static_cast<uint64_t>(visited * scale)); }
    } else {  // raw sample aggregate, DB.cpp:2046
        if (p.agg == AQE_SUM) { value = S; margin = moe * n; }
        else if (p.agg == AQE_AVG) { value = mean; margin = moe; }
        else { value = visited; }
    }
    r.value = value;
    r.margin = margin;
    r.ci_lower = value - margin;
    r.ci_upper = value + margin;
    return r;
}
__device__ __forceinline__ void finalize(const QueryState& s, const FinalizeParams& p, aqe_result* out) { *out = make_result(s, p); }

// Per-tile accumulator (one pointer group per tile), merged into the lane's two groups with selects:
// a data-dependent choice of WHICH accumulator to update makes hipcc index the struct in scratch.
struct TileAcc {
    double s = 0.0, q = 0.0;
    unsigned n = 0, nv = 0;
};

__device__ __forceinline__ void accumulate(TileAcc& t, double x, bool ok, const SweepCommon& a) {
    const bool pass = ok && (!a.has_where || (x >= a.wmin && x <= a.wmax));  // inclusive both ends, DB.cpp:329
    const double d = pass ? x - a.shift : 0.0;
    t.nv += ok ? 1u : 0u;
    t.n += pass ? 1u : 0u;
    t.s += d;
    t.q += d * d;
}

__device__ __forceinline__ void merge_tile(Acc& acc, const TileAcc& t, bool group_b) {
    acc.nv += t.nv;
    acc.na += group_b ? 0u : t.n;
    acc.nb += group_b ? t.n : 0u;
    acc.sa += group_b ? 0.0 : t.s;
    acc.sb += group_b ? t.s : 0.0;
    acc.qa += group_b ? 0.0 : t.q;
    acc.qb += group_b ? t.q : 0.0;
}

// Family tables are tiny (1-64 entries) and every tile decode walks them: keep a copy in LDS so the
// decode costs LDS reads instead of a chain of dependent global loads.  Returns the table to use.
constexpr unsigned kMaxLdsFams = 64;
__device__ __forceinline__ const DevFamily* stage_families(const SweepCommon& a, DevFamily* lds) {
    if (a.nfam > kMaxLdsFams) return a.fams;
    constexpr unsigned kWords = sizeof(DevFamily) / 8;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.fams);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(lds);
    for (unsigned i = threadIdx.x; i < a.nfam * kWords; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    return lds;
}

// One wave folds tile `t` of the launch's family table: 64 * kTileUnroll ordinals of one segment of one
// family (both pointers of a PAIR family).  All indices are wave-uniform except the lane's ordinal.
// Every load of the tile is issued before the first use; out-of-window lanes load row 0 of the shard
// instead of branching around the load (a per-element branch would serialise the round trips:
// cdna_hip_programming.md §5 item 4c).
// the family that owns tile t (tile_begin ascending)
template <typename FamPtr>
__device__ __forceinline__ unsigned find_family(FamPtr fams, unsigned nfam, u64 t) {
    unsigned lo = 0, hi = nfam;
    while (hi - lo > 1) {
        unsigned mid = (lo + hi) >> 1;
        if (fams[mid].tile_begin <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// kNT: the dense path loads non-temporally.  A column (or view) beyond the 256 MiB Infinity Cache is read once per
// query and nothing of it is found again: streaming it through the caches with the default policy costs 9-11 % of the
// bandwidth (1 B rows, exact scan: 1 300 -> 1 180 us = 6.8 TB/s; A/B on one box).  Tables that fit the cache keep the
// default policy: there the next query — or the neighbour group of a batch — finds the lines again (10 M rows, the
// bench batch: 140 us default, 155 us non-temporal).  The host picks the instantiation by the size of what is swept.
template <bool kNT = false, typename Fam>
__device__ __forceinline__ void sweep_family(const SweepCommon& a, const Fam& F, u64 t, int lane, u64 ord_limit, Acc& acc) {
    const u64 lt = t - F.tile_begin;
    u64 seg, j;
    if (F.tiles_per_seg == 0) { seg = F.seg_lo; j = F.j_lo + lt; }
    else { seg = F.seg_lo + lt / F.tiles_per_seg; j = lt % F.tiles_per_seg; }
    const u64 seg_len = F.seg_len, step = F.step;
    const u64 seg_ord0 = seg * seg_len;
    const u64 ord_lo = F.ord_lo;
    const u64 ord_hi = (F.flags & AQE_F_TOPUP) ? (F.ord_hi < ord_limit ? F.ord_hi : ord_limit) : F.ord_hi;
    const double* base = a.amount + (F.row0 + seg * F.pitch - a.shard_lo);

    if (a.dense16 && is_dense16(step, F.flags, seg_len)) {
        // Dense segment (blocks, pages, exact scans): two consecutive rows per lane per 16-byte load, eight loads in
        // flight — 1 KiB per wave instruction, the widest coalesced access.  Rows are only 8-byte aligned, which
        // global loads allow.  An ordinal pair never straddles the segment end unless masked.
        struct __attribute__((packed, aligned(8))) Row2 { double x, y; };
        const bool group_bd = F.group != 0;
        // INTERIOR tile — all of its ordinals inside the segment and inside the window, which is every tile but the
        // one or two at a window's edges: no per-element masks, no address selects.  The general form below spends
        // ~500 vector instructions per tile on them — with four waves per SIMD that, not memory, paces a wave
        // (tools/exp_latency.hip: a lean sweep of 32 MB ends after 3.7 us, this loop took 8.3) — this form ~100.
        // Same operations on the same values in the same order: the sums are bitwise those of the general form.
        const u64 tile_lo = uniform64(j * kDenseTileOrdinals), o_lo = uniform64(seg_ord0 + tile_lo);
        if (tile_lo + kDenseTileOrdinals <= uniform64(seg_len) && o_lo >= uniform64(ord_lo) && o_lo + kDenseTileOrdinals <= uniform64(ord_hi)) {
            const Row2* const p = reinterpret_cast<const Row2*>(base + tile_lo) + lane;
            Row2 v2[kTileUnroll];
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                if (kNT) {
                    v2[k].x = __builtin_nontemporal_load(&p[k * 64].x);
                    v2[k].y = __builtin_nontemporal_load(&p[k * 64].y);
                } else {
                    v2[k] = p[k * 64];
                }
            }
            TileAcc ta;
            ta.nv = 2u * kTileUnroll;
            if (a.has_where) {
#pragma unroll
                for (int k = 0; k < kTileUnroll; ++k) {
                    const double x = v2[k].x, y = v2[k].y;
                    const bool px = x >= a.wmin && x <= a.wmax, py = y >= a.wmin && y <= a.wmax;  // inclusive both ends, DB.cpp:329
                    const double dx = px ? x - a.shift : 0.0, dy = py ? y - a.shift : 0.0;
                    ta.n += (px ? 1u : 0u) + (py ? 1u : 0u);
                    ta.s += dx; ta.q += dx * dx;
                    ta.s += dy; ta.q += dy * dy;
                }
            } else {
                ta.n = 2u * kTileUnroll;
#pragma unroll
                for (int k = 0; k < kTileUnroll; ++k) {
                    const double dx = v2[k].x - a.shift, dy = v2[k].y - a.shift;
                    ta.s += dx; ta.q += dx * dx;
                    ta.s += dy; ta.q += dy * dy;
                }
            }
            merge_tile(acc, ta, group_bd);
            return;
        }
        const u64 oi0d = j * kDenseTileOrdinals + 2 * static_cast<u64>(lane);
        Row2 v2[kTileUnroll];
        bool ok0[kTileUnroll], ok1[kTileUnroll];
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const u64 oi = oi0d + static_cast<u64>(k) * 128;
            const u64 o = seg_ord0 + oi;
            ok0[k] = oi < seg_len && o >= ord_lo && o < ord_hi;
            ok1[k] = oi + 1 < seg_len && o + 1 >= ord_lo && o + 1 < ord_hi;
            // both rows must be readable: fall back to row 0/1 of the shard when the pair leaves the window
            const bool both = ok0[k] && ok1[k];
            const Row2* p = reinterpret_cast<const Row2*>(both ? base + oi : a.amount);
            if (kNT) {
                v2[k].x = __builtin_nontemporal_load(&p->x);  // (the two halves leave as ONE global_load_dwordx4 ... nt)
                v2[k].y = __builtin_nontemporal_load(&p->y);
            } else {
                v2[k] = *p;
            }
            if (!both) {  // window edge (at most one lane per tile side): single 8-byte reads
                v2[k].x = ok0[k] ? base[oi] : 0.0;
                v2[k].y = ok1[k] ? base[oi + 1] : 0.0;
            }
        }
        TileAcc ta;
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) { accumulate(ta, v2[k].x, ok0[k], a); accumulate(ta, v2[k].y, ok1[k], a); }
        merge_tile(acc, ta, group_bd);
        return;
    }
    const u64 oi0 = j * kTileOrdinals + lane;

    if (F.flags & AQE_F_PAIR) {
        // fast + slow pointer of one region in one sweep: the two rows of an ordinal sit in the same or the
        // neighbouring cache line, so the region's lines are fetched once for both groups.
        const double* base_b = a.amount + (F.row0_b - a.shard_lo);
        const u64 lo_b = F.ord_lo_b, hi_b = F.ord_hi_b;
        double va[kTileUnroll], vb[kTileUnroll];
        bool oka[kTileUnroll], okb[kTileUnroll];
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const u64 o = oi0 + static_cast<u64>(k) * 64;
            oka[k] = o >= ord_lo && o < ord_hi;
            okb[k] = o >= lo_b && o < hi_b;
            const double* pa = oka[k] ? base + o * step : a.amount;
            const double* pb = okb[k] ? base_b + o * step : a.amount;
            va[k] = *pa;
            vb[k] = *pb;
        }
        TileAcc ta, tb;
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) { accumulate(ta, va[k], oka[k], a); accumulate(tb, vb[k], okb[k], a); }
        merge_tile(acc, ta, F.group != 0);  // (the first pointer is the leader's only in the leader's region)
        merge_tile(acc, tb, true);
        return;
    }
    const bool group_b = F.group != 0;
    double v[kTileUnroll];
    bool ok[kTileUnroll];
    if (F.flags & kFamLinear) {
        // short segments, tiled along the ordinal axis: ordinal o = T0 + x sits in segment (T0 + x) / seg_len.  One
        // 64-bit division per tile for T0; x + (T0 mod seg_len) stays below 1024, where a float reciprocal is exact
        const u64 T0 = j * kTileOrdinals;
        const u64 seg0 = T0 / seg_len;
        const unsigned r0 = static_cast<unsigned>(T0 - seg0 * seg_len), sl = static_cast<unsigned>(seg_len);
        const float inv = 1.0f / static_cast<float>(sl);
        const double* const col = a.amount + (F.row0 - a.shard_lo);
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const unsigned x = r0 + static_cast<unsigned>(lane) + 64u * static_cast<unsigned>(k);
            const unsigned qx = static_cast<unsigned>((static_cast<float>(x) + 0.5f) * inv);
            const u64 o = T0 + static_cast<unsigned>(lane) + 64u * static_cast<unsigned>(k);
            ok[k] = o >= ord_lo && o < ord_hi;
            const double* p = ok[k] ? col + (seg0 + qx) * F.pitch + static_cast<u64>(x - qx * sl) * step : a.amount;
            v[k] = *p;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const u64 oi = oi0 + static_cast<u64>(k) * 64;
            const u64 o = seg_ord0 + oi;
            ok[k] = oi < seg_len && o >= ord_lo && o < ord_hi;
            const double* p = ok[k] ? base + oi * step : a.amount;
            v[k] = *p;
        }
    }
    TileAcc ta;
#pragma unroll
    for (int k = 0; k < kTileUnroll; ++k) accumulate(ta, v[k], ok[k], a);
    merge_tile(acc, ta, group_b);
}

template <bool kNT = false, typename FamPtr>
__device__ __forceinline__ void sweep_tile(const SweepCommon& a, FamPtr fams, u64 t, int lane, u64 ord_limit, Acc& acc) {
    sweep_family<kNT>(a, fams[find_family(fams, a.nfam, t)], t, lane, ord_limit, acc);
}

}  // namespace
}  // namespace aqe

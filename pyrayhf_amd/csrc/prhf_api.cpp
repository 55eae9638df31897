// prhf_api.cpp - the C ABI of libprhf.so (include/prhf.h): contexts, launch planning,
// host<->device staging, timing and error reporting.  No compute happens on the host;
// there is no CPU fallback: without a GPU every compute entry point fails with PRHF_EHIP.

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <xmmintrin.h>

#include "prhf.h"
#include "prhf_kernels.h"
#include "prhf_plan.h"        // Knobs, plan_slice, validate_work_list: the HIP-free part of the launch planning

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(PRHF_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

constexpr size_t kPackBytes = 1u << 20;
constexpr size_t kSlabMinBytes = 16u << 20;    // host-buffer batches from this many input bytes on go in slabs (run_host_slabs)
constexpr size_t kDirectBytes = 128u << 10;   // inputs up to this size are written by the CPU through the BAR (direct_upload)

constexpr int kStatusWords = 12;           // prhf_ctx::d_status: block queues and scratch words of the launches
constexpr long long kMaxAlt = 1400;        // nodes + hints must fit 160 KiB of LDS
constexpr long long kMaxAltTall = 65535;   // taller profiles are staged in global memory (vfo_tall_kernel); level
                                           // indices travel as uint16 in the hint table

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Every entry point works on its context's device and leaves the calling thread's current HIP device as it
// found it (torch's current device is the same per-thread state: a library call must not move it).
class DeviceScope {
  public:
    explicit DeviceScope(int device) {
        err_ = hipGetDevice(&prev_);
        if (err_ == hipSuccess && prev_ != device) {
            err_ = hipSetDevice(device);
            switched_ = err_ == hipSuccess;
        }
    }
    ~DeviceScope() {
        if (switched_) (void)hipSetDevice(prev_);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    hipError_t error() const { return err_; }

  private:
    int prev_ = -1;
    bool switched_ = false;
    hipError_t err_ = hipSuccess;
};
#define ENTER_DEVICE(dev)            \
    DeviceScope device_scope_(dev);  \
    HIP_TRY(device_scope_.error())

hipError_t create_events_untimed(hipEvent_t* ev, int n) {
    for (int i = 0; i < n; ++i) {
        const hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t create_events(hipEvent_t* ev, int n) {
    for (int i = 0; i < n; ++i) {
        const hipError_t e = hipEventCreate(&ev[i]);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

struct prhf_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream = nullptr;  // mixed lists: the short-grid launch runs beside the general one (fork / join by events)
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    hipStream_t up_stream = nullptr, down_stream = nullptr;   // host-buffer batches in slabs: uploads / downloads beside the kernels
    hipEvent_t slab_up[3] = {}, slab_done[3] = {}, slab_free = nullptr;
    hipStream_t stream = nullptr;
    // start / stop events of the most recent launches, a ring: a caller that enqueues many launches without
    // synchronising can still read every one's device time afterwards (prhf_recent_kernel_ms)
    static constexpr int kTimingRing = 64;
    hipEvent_t ring0[kTimingRing] = {}, ring1[kTimingRing] = {};
    unsigned long long n_timed = 0;    // launches timed so far
    int slot = 0;                      // ring slot of the last launch that was enqueued completely
    int pending = 0;                   // ring slot of the launch being enqueued: published by mark_timed() only, so that
                                       // a launch that fails half way leaves `slot` on the last good pair of events
    bool timed = false;
    hipEvent_t begin_ev() { pending = (int)(n_timed % kTimingRing); return ring0[pending]; }
    hipEvent_t pending_end_ev() const { return ring1[pending]; }
    hipEvent_t end_ev() const { return ring1[slot]; }
    void mark_timed() { slot = pending; ++n_timed; timed = true; }
    int math = PRHF_MATH_AUTO;
    Knobs knobs;
    int cu_count = 256;
    int large_bar = 0;   // hipDeviceAttributeIsLargeBar: device memory is mapped into the host's address space
    DevBuf arena;     // staged host inputs + output
    DevBuf partial;   // chunk sums
    DevBuf altmin;    // per-profile min(alt) for chunked slices
    DevBuf pairs;     // (m_i, m_i+1 - m_i) table of the fast tier's main loop
    DevBuf ftab;      // per-frequency scalars of a long launch
    DevBuf levels;    // level table of a grouped tracer launch
    DevBuf ptab;      // tracers: per-profile table of the frequency-independent parts of a level's mu, mu'
    DevBuf leftover;  // short-grid launches: the profiles left to the general kernel (count + block indices)
    DevBuf leftover_x;   // ... of the X-mode short-grid launch
    DevBuf leftover_tall;   // compact short-grid launch: the profiles whose bottomside needs the full-size arrays
    DevBuf leftover_tall_x; // ... of the X-mode short-grid launch
    DevBuf order;           // short-grid launch: its blocks by cost class (short_order_kernel)
    DevBuf tall;         // profiles of more than kMaxAlt levels: one slab of staged levels per resident workgroup
    const double* pairs_src = nullptr;   // PRHF_FLAG_GRID_STABLE: multiplier array the table was built from
    int64_t pairs_len = 0;
    // PRHF_FLAG_GRID_STABLE with HOST buffers: the stretched grid at a host address that keeps its contents is
    // uploaded, and its pair table built, once (a 20 000-point grid is 160 KB: more than the rest of a
    // single-profile call's inputs together)
    struct HostGrid {
        const double* host = nullptr;
        int64_t len = 0;
        DevBuf mult, pairs;
        bool pairs_ready = false;
    };
    static constexpr int kHostGrids = 16;
    HostGrid host_grid[kHostGrids];
    int n_host_grids = 0;
    bool order_clean = false;       // the class counters at the head of `order` are zero (short_order_kernel needs them so; the
                                    // follow-up kernel of the launch that used them leaves them so)
    unsigned* d_status = nullptr;   // device words [1..5]: block queues of persistent launches (general, short-grid O and its
                                    // follow-up, short-grid X and its follow-up); [6]: ray queue of the per-ray tracer launch;
                                    // [0] unused
    unsigned* h_status = nullptr;   // PRHF_STATUS_WORDS words of pinned host memory mapped into the device: word b = status
    unsigned* h_status_dev = nullptr;   // bit b (post_status) - nothing to copy back or reset on the device; its device address
    double* h_pack = nullptr;       // pinned, kPackBytes: inputs of a small host-buffer call, sent in one piece; its upper
    double* h_pack_dev = nullptr;   // half, mapped into the device (h_pack_dev), takes a small result straight from the kernel
    unsigned long long* d_words = nullptr;   // 2 words: nanmax|Y| bits, any-not-NaN
    unsigned long long* h_words = nullptr;   // pinned
    bool status_pending = false;
};

namespace {

int ensure(prhf_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return PRHF_OK;
    if (b.p) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(PRHF_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return PRHF_OK;
}

// Optional second stage of a launch: residual rows against one observed trace (prhf_vfo_residual_f64).
struct Residual {
    const double* vh_obs;   // (n_freq)
    double* residual;       // (n_prof, n_freq) or null
    double* cost;           // (n_prof) or null
};

int status_to_code(unsigned bits) {
    if (bits & PRHF_STATUS_PEAK0)
        return fail(PRHF_EPEAK0, "density peak at index 0: no bottomside levels below the peak");
    if (bits & PRHF_STATUS_NANINPUT)
        return fail(PRHF_EINVAL, "NaN in a profile (alt anywhere in the column, bmag or bpsi below the density peak)");
    if (bits & PRHF_STATUS_NEGDEN) return fail(PRHF_ENEGDEN, "Density must be non-negative");
    if (bits & PRHF_STATUS_BADGROUP) return fail(PRHF_EINVAL, "ray_group outside [0, n_groups)");
    if (bits & PRHF_STATUS_BADINDEX) return fail(PRHF_EINVAL, "profile_index outside [0, n_prof)");
    return PRHF_OK;
}

int run(prhf_ctx* c, const double* freq, int64_t n_freq, const double* den, const double* bmag,
        const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt, int64_t prof_stride,
        int64_t alt_stride, const double* mult, int64_t mult_len, const prhf_segment* segs, int32_t n_segs,
        double* out, uint32_t flags, const Residual* post = nullptr) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (!freq || !den || !bmag || !bpsi || !alt || !mult || !segs || (!out && !post))
        return fail(PRHF_EINVAL, "null array pointer");
    if (post && (!post->vh_obs || (!post->residual && !post->cost)))
        return fail(PRHF_EINVAL, "null array pointer");
    if (n_freq < 1 || n_prof < 0 || n_alt < 1) return fail(PRHF_EINVAL, "bad shape");
    if (n_alt > kMaxAltTall) return fail(PRHF_EINVAL, "n_alt %lld exceeds the limit of %lld levels",
                                         (long long)n_alt, kMaxAltTall);
    // Profiles of more levels than LDS holds are staged in global memory and take the generic loop (vfo_tall_kernel):
    // no main loop, no candidate list, no short-grid kernels - the same values as any profile that leaves those paths.
    // Decided below, once the highest density peak of the launch is known: only the bottomside is staged, and a
    // column of 2 500 levels at 0.25 km has its peak near level 900.
    bool tall = n_alt > kMaxAlt;
    if (n_freq > (1 << 20)) return fail(PRHF_EINVAL, "n_freq too large");
    if (prof_stride < n_alt || (alt_stride != 0 && alt_stride < n_alt))
        return fail(PRHF_EINVAL, "row stride shorter than a row");
    if (n_segs < 1 || n_segs > PRHF_MAX_SEGMENTS)
        return fail(PRHF_EINVAL, "1..%d segments per launch", PRHF_MAX_SEGMENTS);
    if (flags & ~(PRHF_FLAG_DEVICE_PTRS | PRHF_FLAG_ASYNC | PRHF_FLAG_GRID_STABLE | PRHF_FLAG_SHARED_FIELD))
        return fail(PRHF_EINVAL, "unknown flag bits");
    const bool dev = (flags & PRHF_FLAG_DEVICE_PTRS) != 0;
    const bool shared_field = (flags & PRHF_FLAG_SHARED_FIELD) != 0;
    if ((flags & PRHF_FLAG_ASYNC) && !dev) return fail(PRHF_EINVAL, "PRHF_FLAG_ASYNC needs device pointers");
    // (A sounder frequency that is not a positive finite number gives a NaN column, host and device buffers alike:
    //  freq_table_kernel / pair_freq.  The reference returns NaN for 0 and NaN, and something meaningless for f < 0.)
    // The stretched grid must not decrease (smooth_nonuniform_grid never does): the top-segment search of the main
    // loop relies on it.  Checked here for host buffers; device-resident grids are the caller's.
    bool grid_known = false;                   // (a stable host grid that is cached already was checked when it was uploaded)
    if (!dev && (flags & PRHF_FLAG_GRID_STABLE))
        for (int g = 0; g < c->n_host_grids; ++g)
            grid_known = grid_known || (c->host_grid[g].host == mult && c->host_grid[g].len == mult_len);
    if (!dev && !grid_known) {
        const long long bad = first_decreasing_grid_entry(mult, mult_len, segs, n_segs);
        if (bad >= 0) return fail(PRHF_EINVAL, "multiplier[%lld] decreases: the stretched grid must be non-decreasing", bad);
    }

    ENTER_DEVICE(c->device);

    // More levels than LDS holds: find the highest peak index of the launch (np.argmax, the first NaN ranking highest -
    // stage_profile's rule).  When every bottomside fits, the LDS kernels run with their staged arrays sized for that
    // peak (KArgs::lds_levels).
    long long lds_levels = n_alt;
    if (tall && n_prof > 0 && c->knobs.trim_lds != 0) {
        long long max_peak = 0;
        if (dev) {
            HIP_TRY(hipMemsetAsync(c->d_status + 7, 0, sizeof(unsigned), c->stream));
            HIP_TRY(prhf::launch_peak_levels(den, n_prof, n_alt, prof_stride, c->d_status + 7, c->stream));
            unsigned peak = 0;
            HIP_TRY(hipMemcpyAsync(&peak, c->d_status + 7, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            max_peak = peak;
        } else {
            for (int64_t p = 0; p < n_prof; ++p) {
                const double* d = den + (size_t)p * prof_stride;
                double bv = -HUGE_VAL;
                long long bi = 0;
                for (int64_t i = 0; i < n_alt; ++i) {
                    const double key = (d[i] != d[i]) ? HUGE_VAL : d[i];
                    if (key > bv) { bv = key; bi = i; }
                }
                max_peak = std::max(max_peak, bi);
            }
        }
        if (max_peak + 1 <= kMaxAlt) {
            tall = false;
            lds_levels = max_peak + 1;
        }
    }

    const Knobs& kn = c->knobs;
    const double kWellConditioned = kn.well_conditioned, kThreadScanMinWork = kn.thread_scan_min;
    const int kLeanMinPoints = (int)kn.lean_min_points, kNoCandidates = kn.no_candidates != 0;
    const bool kPersistent = kn.persistent != 0, kShortKernel = kn.short_kernel != 0, kShortXKernel = kn.shortx_kernel != 0;
    const bool kShortConcurrent = kn.short_concurrent != 0;
    const int kShortQueueFixed = (int)kn.short_queue;
    const int n_user_segs = n_segs;
    prhf::KArgs a;
    std::memset(&a, 0, sizeof a);
    a.n_freq = n_freq;
    a.n_alt = n_alt;
    a.lds_levels = lds_levels;
    a.n_segs = n_segs;
    // Slices of short O-mode grids leave for a launch of their own (vfo_short_kernel): `a` keeps the others
    prhf::SegDev short_seg[PRHF_MAX_SEGMENTS], shortx_seg[PRHF_MAX_SEGMENTS];
    int n_short = 0, n_shortx = 0;
    // LDS budget of a short-grid workgroup: two per CU where its nodes allow that, else one (a few hundred bytes of
    // static LDS - tickets, counters - come on top)
    const size_t lds_half = 80 * 1024 - 512, lds_full = 160 * 1024 - 512;
    const size_t short_budget = prhf::short_queue_entries(lds_levels, n_freq, lds_half, PRHF_SHORT_THREADS) ? lds_half : lds_full;
    const int short_queue = prhf::short_queue_entries(lds_levels, n_freq, short_budget, PRHF_SHORT_THREADS);
    // The compact geometry of the short-grid O kernel: four 4-wave workgroups per CU instead of two
    // 8-wave ones - four independent profiles in flight per CU, so that one workgroup's staging and barrier waits are
    // covered by three others' items (config 3: -10 %).  A quarter of the LDS holds the lists and fewer levels than
    // the column has; a profile whose peak lies above them goes to a second launch with full-size arrays.  Taken when
    // those arrays hold at least half of the column (PyIRI columns peak at 25 - 50 % of their height).
#ifndef PRHF_COMPACT_RESERVE
#define PRHF_COMPACT_RESERVE 512    // bytes kept back per workgroup for its static LDS (tickets, counters)
#endif
    const size_t lds_quarter = (160 * 1024) / PRHF_COMPACT_WGS_PER_CU - PRHF_COMPACT_RESERVE;
    long long compact_levels = 0;
    int compact_queue = 0;
    if (kn.short_compact != 0 && short_queue > 0) {
        long long L = lds_levels;
        while (L > 1 && prhf::short_lds_fixed(L, n_freq, PRHF_COMPACT_THREADS) + 8 * PRHF_COMPACT_MIN_QUEUE > lds_quarter) --L;
        if (2 * L >= lds_levels && L >= 8 &&
            prhf::short_lds_fixed(L, n_freq, PRHF_COMPACT_THREADS) + 8 * PRHF_COMPACT_MIN_QUEUE <= lds_quarter) {
            compact_levels = L;
            compact_queue = prhf::short_queue_entries(L, n_freq, lds_quarter, PRHF_COMPACT_THREADS);
        }
    }
    long long blocks = 0, partial_elems = 0, altmin_elems = 0, out_rows = 0;
    int launch_tier = 0;
    bool want_pairs = false;
    // resident workgroups: LDS admits two per CU up to 80 KiB each, else one
    const long long wg_slots = (long long)c->cu_count * ((tall || prhf::lds_bytes_for(lds_levels) <= 80 * 1024) ? 2 : 1);
    {
        char why[160];
        if (validate_work_list(segs, n_segs, n_prof, n_freq, mult_len, why, sizeof why) != PRHF_OK) return fail(PRHF_EINVAL, "%s", why);
    }
    for (int i = 0; i < n_segs; ++i) {
        const prhf_segment& u = segs[i];
        prhf::SegDev& s = a.seg[i];
        s.prof_begin = u.prof_begin;
        s.prof_end = u.prof_end;
        s.mult_off = u.mult_offset;
        s.out_off = u.out_offset;
        s.mode = u.mode == PRHF_MODE_O ? PRHF_KMODE_O : PRHF_KMODE_X;
        s.n_points = u.n_points;
        s.tier = c->math == PRHF_MATH_AUTO ? (u.mode == PRHF_MODE_O ? 0 : 1) : (c->math == PRHF_MATH_FAST ? 1 : 0);
        // AUTO, O mode: the reference's operation order where it decides the answer (1 - X <= well_conditioned = 1e-5), the
        // reduced algebra elsewhere; PRHF_MATH_FAITHFUL keeps the reference's order everywhere
        s.well_conditioned = (c->math == PRHF_MATH_AUTO && s.tier == 0) ? kWellConditioned : HUGE_VAL;
        launch_tier = (i == 0 || launch_tier == s.tier) ? s.tier : 2;
        // the main loop needs the pair table: one more (small) kernel unless the caller's grid is cached - not
        // worth it for a handful of pairs on a short grid, where the launch itself is the cost
        const long long seg_pairs = (u.prof_end - u.prof_begin) * n_freq;
        // (decided from the slice's shape alone: host and device callers must get the same arithmetic)
        const bool table_is_cheap = seg_pairs >= 4096 || u.n_points >= 2048;
        // (a profile staged in global memory - `tall` - takes the main loop too: it reads the nodes from the workgroup's
        //  slab through a buffer resource, NodeSpace<true>; option tall_lean = 0: the generic loop, as up to round 4)
        s.lean = ((!tall || kn.tall_lean != 0) && (s.tier == 1 || s.well_conditioned < 1.0) && u.n_points >= kLeanMinPoints &&
                  seg_pairs > 0 && table_is_cheap) ? 1 : 0;
        want_pairs = want_pairs || s.lean != 0;
        s.thread_scan = (!tall && (double)n_freq * (double)u.n_points >= kThreadScanMinWork) ? 1 : 0;
        plan_slice(s, n_freq, wg_slots, kn);
        out_rows = std::max<long long>(out_rows, u.out_offset / n_freq + (u.prof_end - u.prof_begin));
    }
    // (a.seg[i] was filled for every slice; now the short-grid slices move out and the others close ranks)
    {
        int kept = 0;
        for (int i = 0; i < n_segs; ++i) {
            const prhf::SegDev& s = a.seg[i];
            // the short-grid kernel reads the per-frequency table (launches of >= 4096 pairs) and lists at most
            // PRHF_MAX_CAND frequencies per profile
            // (grids shorter than the general kernel's main loop takes - lean_min_points - are theirs too: the pair
            //  table is built for them)
            const long long slice_pairs = (s.prof_end - s.prof_begin) * n_freq;
            const bool table = !tall && (s.lean || (slice_pairs >= 4096 && s.n_points < kLeanMinPoints));
            const bool is_short = kShortKernel && s.tier == 0 && s.well_conditioned < 1.0 && table && s.chunks == 1 &&
                                  s.n_points >= PRHF_SHORT_MIN_POINTS && s.n_points <= PRHF_SHORT_MAX_POINTS &&
                                  n_freq <= PRHF_MAX_CAND && n_prof * n_freq >= 4096 && !kNoCandidates && short_queue > 0;
            // ... and its X-mode variant (fast tier, reflection heights per thread, no top-segment phase)
            const bool is_shortx = kShortXKernel && s.tier == 1 && s.mode == PRHF_KMODE_X && table && s.chunks == 1 &&
                                   s.n_points >= PRHF_SHORT_MIN_POINTS && s.n_points <= PRHF_SHORTX_MAX_POINTS &&
                                   n_freq <= PRHF_MAX_CAND && n_prof * n_freq >= 4096 && !kNoCandidates && s.thread_scan &&
                                   prhf::shortx_lds_bytes(lds_levels, n_freq) <= lds_full;
            if (is_short || is_shortx) {
                prhf::SegDev& t = is_short ? short_seg[n_short++] : shortx_seg[n_shortx++];
                t = s;
                t.lean = 1;
                want_pairs = true;
                t.blocks_per_prof = 1;
                t.tail_prof = t.prof_end - t.prof_begin;
                t.tail_bpp = 1;
                t.prio = 0;
            } else {
                if (kept != i) a.seg[kept] = a.seg[i];
                ++kept;
            }
        }
        n_segs = kept;
        a.n_segs = kept;
        launch_tier = 0;
        for (int i = 0; i < kept; ++i) launch_tier = (i == 0 || launch_tier == a.seg[i].tier) ? a.seg[i].tier : 2;
    }
    // Workgroups are dispatched roughly in index order: give the slices with the most work per workgroup
    // the lowest indices so that a mixed launch does not end on its longest workgroups.
    std::stable_sort(a.seg, a.seg + n_segs, [](const prhf::SegDev& x, const prhf::SegDev& y) {
        auto cost = [](const prhf::SegDev& s) {
            const double per_point = s.tier == 1 ? 1.0 : (s.well_conditioned < 1.0 ? 1.8 : 3.5);
            return (double)s.n_points / s.chunks / s.blocks_per_prof * per_point;   // head workgroups
        };
        return cost(x) > cost(y);
    });
    for (int i = 0; i < n_segs; ++i) {
        prhf::SegDev& s = a.seg[i];
        const long long P = s.prof_end - s.prof_begin;
        s.prio = std::max(0, 3 - i);               // (sorted: the slice with the longest workgroups first)
        s.block_begin = blocks;
        blocks += s.tail_prof * s.blocks_per_prof + (P - s.tail_prof) * s.tail_bpp;
        if (s.chunks > 1 && s.slots == 0) {
            s.partial_off = partial_elems;
            s.altmin_off = altmin_elems;
            partial_elems += P * n_freq * s.chunks;
            altmin_elems += P;
        }
    }
    if (blocks > 0x7fffffffLL) return fail(PRHF_EINVAL, "launch too large");

    int rc;
    if ((rc = ensure(c, c->partial, (size_t)partial_elems * 8)) != PRHF_OK) return rc;
    if ((rc = ensure(c, c->altmin, (size_t)altmin_elems * 8)) != PRHF_OK) return rc;
    a.partial = static_cast<double*>(c->partial.p);
    a.altmin = static_cast<double*>(c->altmin.p);
    a.status = c->h_status_dev;

    const size_t row_bytes = (size_t)n_alt * 8;
    double* d_out = nullptr;
    bool out_direct = false;
    prhf_ctx::HostGrid* grid = nullptr;        // host buffers with PRHF_FLAG_GRID_STABLE: the cached device copy of the grid
    const size_t out_elems = (size_t)out_rows * (size_t)n_freq;
    if (dev) {
        a.freq = freq; a.den = den; a.bmag = bmag; a.bpsi = bpsi; a.alt = alt; a.mult = mult;
        a.out = out;
        a.prof_stride = prof_stride;
        a.field_stride = shared_field ? 0 : prof_stride;
        a.alt_stride = alt_stride;
    } else {
        const size_t n_alt_rows = alt_stride ? (size_t)n_prof : 1;
        const size_t n_field_rows = shared_field ? 1 : (size_t)n_prof;
        const size_t post_elems = post ? (size_t)n_freq + out_elems + (size_t)n_prof : 0;
        // a stable host grid lives in a device buffer of its own, uploaded on first sight
        if (flags & PRHF_FLAG_GRID_STABLE) {
            for (int g = 0; g < c->n_host_grids; ++g)
                if (c->host_grid[g].host == mult && c->host_grid[g].len == mult_len) grid = &c->host_grid[g];
            if (!grid && c->n_host_grids < prhf_ctx::kHostGrids) {
                prhf_ctx::HostGrid& g = c->host_grid[c->n_host_grids];
                if ((rc = ensure(c, g.mult, (size_t)mult_len * 8)) != PRHF_OK) return rc;
                HIP_TRY(hipMemcpyAsync(g.mult.p, mult, (size_t)mult_len * 8, hipMemcpyHostToDevice, c->stream));
                g.host = mult;
                g.len = mult_len;
                g.pairs_ready = false;
                grid = &g;
                ++c->n_host_grids;
            }
        }
        const size_t mult_arena = grid ? 0 : (size_t)mult_len;
        const size_t elems = (size_t)n_freq + ((size_t)n_prof + 2 * n_field_rows) * n_alt + n_alt_rows * n_alt +
                             mult_arena + out_elems + post_elems;
        if ((rc = ensure(c, c->arena, elems * 8)) != PRHF_OK) return rc;
        double* base = static_cast<double*>(c->arena.p);
        double* d_freq = base;
        double* d_den = d_freq + n_freq;
        double* d_bmag = d_den + (size_t)n_prof * n_alt;
        double* d_bpsi = d_bmag + n_field_rows * n_alt;
        double* d_alt = d_bpsi + n_field_rows * n_alt;
        double* d_mult = d_alt + n_alt_rows * n_alt;
        d_out = d_mult + mult_arena;
        const size_t in_elems = (size_t)(d_out - base);
        if (in_elems * 8 <= kPackBytes && c->h_pack) {
            // A small call (the reference's usual one: a single profile): six separate uploads from pageable
            // memory cost more than the kernel.  Pack the inputs in the arena's own order into a pinned
            // buffer and send them in one piece.  The buffer is reused by the next call, so the copy must
            // have left it first: the previous call has synchronised unless it was asynchronous, which host
            // buffers never are.
            // On a large-BAR device (every Instinct board) the staging buffer is not needed either: device memory is
            // mapped into this process, so the CPU writes the 22 KB of a single-profile call straight into the arena -
            // posted write-combined stores over PCIe, 0.4 - 0.8 us (tools/probes/bar_probe.cpp) - and the copy kernel
            // the runtime would have launched for the upload, with its dependency in front of ours (~10 us between
            // them), is gone.  Ordering: the stores are fenced before the launch's doorbell write, which travels the same
            // way behind them; the kernel's start-of-kernel acquire drops what its L2 still holds of the arena.  The arena
            // must be idle: host-buffer calls synchronise before they return, but an asynchronous device-pointer call -
            // or a launch on a stream the context borrowed before - may still be using it; then the staged copy, which is
            // ordered by the stream, is taken.
            bool direct = c->large_bar && kn.direct_upload != 0 && in_elems * 8 <= kDirectBytes &&
                          hipStreamQuery(c->stream) == hipSuccess && (!c->timed || hipEventQuery(c->end_ev()) == hipSuccess);
            (void)hipGetLastError();               // (hipErrorNotReady from the two queries is not an error)
            double* h = direct ? base : c->h_pack;
            std::memcpy(h + (d_freq - base), freq, (size_t)n_freq * 8);
            for (int64_t p = 0; p < n_prof; ++p) {
                std::memcpy(h + (d_den - base) + (size_t)p * n_alt, den + (size_t)p * prof_stride, row_bytes);
                if (!shared_field || p == 0) {
                    std::memcpy(h + (d_bmag - base) + (size_t)p * n_alt, bmag + (size_t)p * prof_stride, row_bytes);
                    std::memcpy(h + (d_bpsi - base) + (size_t)p * n_alt, bpsi + (size_t)p * prof_stride, row_bytes);
                }
                if (alt_stride) std::memcpy(h + (d_alt - base) + (size_t)p * n_alt, alt + (size_t)p * alt_stride, row_bytes);
            }
            if (!alt_stride) std::memcpy(h + (d_alt - base), alt, row_bytes);
            if (!grid) std::memcpy(h + (d_mult - base), mult, (size_t)mult_len * 8);
            if (direct) _mm_sfence();
            else HIP_TRY(hipMemcpyAsync(base, h, in_elems * 8, hipMemcpyHostToDevice, c->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(d_freq, freq, (size_t)n_freq * 8, hipMemcpyHostToDevice, c->stream));
            if (!grid) HIP_TRY(hipMemcpyAsync(d_mult, mult, (size_t)mult_len * 8, hipMemcpyHostToDevice, c->stream));
            if (n_prof > 0) {
                HIP_TRY(hipMemcpy2DAsync(d_den, row_bytes, den, (size_t)prof_stride * 8, row_bytes, (size_t)n_prof,
                                         hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpy2DAsync(d_bmag, row_bytes, bmag, (size_t)prof_stride * 8, row_bytes, n_field_rows,
                                         hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpy2DAsync(d_bpsi, row_bytes, bpsi, (size_t)prof_stride * 8, row_bytes, n_field_rows,
                                         hipMemcpyHostToDevice, c->stream));
            }
            if (alt_stride) {
                if (n_prof > 0)
                    HIP_TRY(hipMemcpy2DAsync(d_alt, row_bytes, alt, (size_t)alt_stride * 8, row_bytes, (size_t)n_prof,
                                             hipMemcpyHostToDevice, c->stream));
            } else {
                HIP_TRY(hipMemcpyAsync(d_alt, alt, row_bytes, hipMemcpyHostToDevice, c->stream));
            }
        }
        a.freq = d_freq; a.den = d_den; a.bmag = d_bmag; a.bpsi = d_bpsi; a.alt = d_alt;
        a.mult = grid ? static_cast<const double*>(grid->mult.p) : d_mult;
        a.out = d_out;
        // rows that no segment covers must come back as NaN, not as whatever the arena held (all-ones bytes = NaN)
        long long covered = 0;
        for (int i = 0; i < n_user_segs; ++i) covered += segs[i].prof_end - segs[i].prof_begin;
        // A small result (the reference's usual call: one profile) goes straight from the kernel into pinned host
        // memory - the upper half of the pack buffer, which the device sees - instead of into the arena and through
        // a copy of its own: one runtime call and one DMA round trip less per call.
        out_direct = out && out_elems && !post && c->h_pack && out_elems * 8 <= kPackBytes / 4 &&
                     in_elems * 8 <= kPackBytes / 2;
        if (out_direct) {
            a.out = c->h_pack_dev + kPackBytes / 16;           // doubles: byte offset kPackBytes / 2
            if (covered < out_rows) std::memset(c->h_pack + kPackBytes / 16, 0xFF, out_elems * 8);
            covered = out_rows;                                // (no device-side fill)
        }
        if (covered < out_rows && out_elems) HIP_TRY(hipMemsetAsync(d_out, 0xFF, out_elems * 8, c->stream));
        a.prof_stride = n_alt;
        a.field_stride = shared_field ? 0 : n_alt;
        a.alt_stride = alt_stride ? n_alt : 0;
    }

    double* vh_dev = a.out;
    if (dev && post && !out) {                 // caller does not want the modeled trace: keep it in scratch
        if ((rc = ensure(c, c->arena, out_elems * 8)) != PRHF_OK) return rc;
        vh_dev = a.out = static_cast<double*>(c->arena.p);
    }
    const double* d_obs = post ? post->vh_obs : nullptr;
    double* d_res = post ? post->residual : nullptr;
    double* d_cost = post ? post->cost : nullptr;
    if (post && !dev) {
        double* p0 = d_out + out_elems;
        HIP_TRY(hipMemcpyAsync(p0, post->vh_obs, (size_t)n_freq * 8, hipMemcpyHostToDevice, c->stream));
        d_obs = p0;
        d_res = post->residual ? p0 + n_freq : nullptr;
        d_cost = post->cost ? p0 + n_freq + out_elems : nullptr;
    }

    // (a synchronous host-buffer call may go untimed - option `timing`: nothing is left on the stream when it returns,
    //  so no later launch or stream switch needs its end event either)
    const bool timed_launch = dev || kn.timing != 0;
    if (timed_launch) HIP_TRY(hipEventRecord(c->begin_ev(), c->stream));
    bool zeroed_by_table = false;              // the control words below were zeroed by the per-frequency table's kernel
    unsigned* order_made = nullptr;            // ... which also sorted the short-grid O launch's blocks by cost (short_order_kernel)
    // resident workgroups of the short-grid O launch (launch_short_kind sizes it the same way)
    auto short_o_slots = [&]() -> long long {
        const bool compact = compact_levels > 0;
        const int threads = compact ? PRHF_COMPACT_THREADS : PRHF_SHORT_THREADS;
        const size_t lds = prhf::short_lds_fixed(compact ? compact_levels : lds_levels, n_freq, threads) +
                           8 * (size_t)(compact ? compact_queue : short_queue);
        return (long long)c->cu_count * (compact ? PRHF_COMPACT_WGS_PER_CU : (lds <= lds_half ? 2 : 1));
    };
    if (want_pairs && grid) {
        if (!grid->pairs_ready) {
            if ((rc = ensure(c, grid->pairs, ((size_t)mult_len + PRHF_PAIR_PAD) * 16)) != PRHF_OK) return rc;
            HIP_TRY(prhf::launch_grid_pairs(a.mult, mult_len, static_cast<double*>(grid->pairs.p), c->stream));
            grid->pairs_ready = true;
        }
        a.pairs = static_cast<const double*>(grid->pairs.p);
    }
    if (want_pairs) {
        const bool stable = dev && (flags & PRHF_FLAG_GRID_STABLE) != 0;
        if (grid) {
            // (table cached with the grid, above)
        } else if (!(stable && c->pairs.p && c->pairs_src == a.mult && c->pairs_len == mult_len)) {
            c->pairs_src = nullptr;
            if ((rc = ensure(c, c->pairs, ((size_t)mult_len + PRHF_PAIR_PAD) * 16)) != PRHF_OK) return rc;
            HIP_TRY(prhf::launch_grid_pairs(a.mult, mult_len, static_cast<double*>(c->pairs.p), c->stream));
            if (stable) {
                c->pairs_src = a.mult;
                c->pairs_len = mult_len;
            }
        }
        if (!grid) a.pairs = static_cast<const double*>(c->pairs.p);
        // per-frequency scalars: long launches read them from a table instead of dividing once per pair (a short
        // launch - one profile - is latency bound: it does without the extra kernel)
        if (n_prof * n_freq >= 4096) {
            if ((rc = ensure(c, c->ftab, ((size_t)n_freq + 1) * 64)) != PRHF_OK) return rc;
            // The table's kernel zeroes, on the way, every control word the launches behind it count in: the block
            // queues and - their buffers sized here, as launch_short_kind sizes them again - the heads of the short-grid
            // kernels' lists and the classes of the block order.  One memset each, they were six operations on the stream
            // in front of a short-grid launch (config 3: ~25 us of 530).
            prhf::ZeroWords zero;
            std::memset(&zero, 0, sizeof zero);
            auto zero_head = [&](DevBuf& b, size_t bytes, int words) -> int {
                int rcz = ensure(c, b, bytes);
                if (rcz != PRHF_OK) return rcz;
                zero.p[zero.n] = static_cast<unsigned*>(b.p);
                zero.words[zero.n++] = words;
                return PRHF_OK;
            };
            zero.p[zero.n] = c->d_status;
            zero.words[zero.n++] = kStatusWords;
            long long o_blocks = 0;
            for (int xmode = 0; xmode < 2; ++xmode) {
                long long kind_blocks = 0;
                for (int i = 0; i < (xmode ? n_shortx : n_short); ++i)
                    kind_blocks += (xmode ? shortx_seg : short_seg)[i].prof_end - (xmode ? shortx_seg : short_seg)[i].prof_begin;
                if (kind_blocks == 0 || kind_blocks > 0x7fffffffLL) continue;
                if (!xmode) o_blocks = kind_blocks;
                const size_t list_bytes = (size_t)(kind_blocks + 1) * sizeof(unsigned);
                if ((rc = zero_head(xmode ? c->leftover_x : c->leftover, list_bytes, 1)) != PRHF_OK) return rc;
                if ((rc = zero_head(xmode ? c->leftover_tall_x : c->leftover_tall, list_bytes, 1)) != PRHF_OK) return rc;
            }
            // A short-grid O launch of four resident rounds and more draws its blocks in descending order of a cost
            // estimate: the kernel that sorts them makes the table as well (short_order_kernel)
            if (o_blocks > 0 && kn.short_order != 0 && o_blocks >= 4 * short_o_slots()) {
                const size_t words = PRHF_ORDER_CLASSES * (size_t)(o_blocks + 1);
                const void* before = c->order.p;
                if ((rc = ensure(c, c->order, words * sizeof(unsigned))) != PRHF_OK) return rc;
                unsigned* order = static_cast<unsigned*>(c->order.p);
                if (!c->order_clean || c->order.p != before)
                    HIP_TRY(hipMemsetAsync(order, 0, PRHF_ORDER_CLASSES * sizeof(unsigned), c->stream));
                prhf::KArgs ap = a;
                ap.n_segs = n_short;
                ap.n_blocks = 0;
                for (int i = 0; i < n_short; ++i) {
                    ap.seg[i] = short_seg[i];
                    ap.seg[i].block_begin = ap.n_blocks;
                    ap.n_blocks += short_seg[i].prof_end - short_seg[i].prof_begin;
                }
                c->order_clean = false;                // (until the follow-up kernel of this launch has run)
                HIP_TRY(prhf::launch_short_order(ap, order, a.freq, static_cast<double*>(c->ftab.p), zero, c->stream));
                order_made = order;
            } else {
                HIP_TRY(prhf::launch_freq_table(a.freq, n_freq, static_cast<double*>(c->ftab.p), zero, c->stream));
            }
            a.ftab = static_cast<const double*>(c->ftab.p);
            zeroed_by_table = true;
        }
    }
#ifdef PRHF_TRACE
    // diagnostics build (tools/wave_trace.py): per-wave wall-clock stamps of this launch, dumped to $PRHF_TRACE_FILE
    static DevBuf trace_buf;
    const size_t trace_words = (size_t)blocks * kWavesPerBlock * 6;   // start, end, staged, and three staging marks
    if (std::getenv("PRHF_TRACE_FILE") && blocks > 0) {
        if ((rc = ensure(c, trace_buf, trace_words * 8)) != PRHF_OK) return rc;
        HIP_TRY(hipMemsetAsync(trace_buf.p, 0, trace_words * 8, c->stream));
        a.trace = static_cast<unsigned long long*>(trace_buf.p);
    }
#endif
    a.n_blocks = blocks;
    a.no_candidates = kNoCandidates || tall;
    if (tall && blocks > 0) {
        a.tall_stride = prhf::tall_slab_bytes(n_alt);
        if ((rc = ensure(c, c->tall, (size_t)std::min(blocks, wg_slots) * a.tall_stride)) != PRHF_OK) return rc;
        a.tall = static_cast<unsigned char*>(c->tall.p);
    }
    // the launches' block queues: words 0 - 5, and word 8 for the second launch of the X-mode short grids (words 6 and 7
    // belong to the tracers and the peak pre-pass).  A launch without a queue - the single profile - enqueues neither.
    if (!zeroed_by_table && (n_short > 0 || n_shortx > 0 || ((kPersistent || tall) && blocks > wg_slots))) {
        HIP_TRY(hipMemsetAsync(c->d_status, 0, 6 * sizeof(unsigned), c->stream));
        if (n_shortx > 0) HIP_TRY(hipMemsetAsync(c->d_status + 8, 0, sizeof(unsigned), c->stream));
    }
    // A list with both kinds of slices: the general launch goes first, on the caller's stream, and takes every
    // workgroup slot; the short-grid launch runs on a second stream and its workgroups move in as the general
    // launch's persistent workgroups leave - its 30 - 100 us blocks fill the end of the launch, which otherwise drains
    // on a few long blocks.  (One after the other on one stream the config-5 shard took 8.04 ms, general kernel alone
    // 7.94 ms.)  Fork and join by events; a launch of one kind stays on the caller's stream.
    auto launch_general = [&]() -> int {
        long long grid_blocks = blocks;
        if ((kPersistent || tall) && blocks > wg_slots) {    // persistent workgroups pulling blocks from a queue (vfo_kernel)
            a.queue = c->d_status + 1;
            grid_blocks = wg_slots;
        }
        if (tall) HIP_TRY(prhf::launch_vfo_tall(a, grid_blocks, c->stream));
        else HIP_TRY(prhf::launch_vfo(a, grid_blocks, launch_tier, prhf::lds_bytes_for(lds_levels), c->stream));
        return PRHF_OK;
    };
    const bool any_short = n_short > 0 || n_shortx > 0;
    const bool forked = any_short && blocks > 0 && kShortConcurrent;
    hipStream_t short_stream = forked ? c->aux_stream : c->stream;
    if (forked) {
        HIP_TRY(hipEventRecord(c->fork_ev, c->stream));        // tables, queues and inputs are in place
        HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->fork_ev, 0));
        if ((rc = launch_general()) != PRHF_OK) return rc;
    }
    // Short grids: vfo_short_kernel (O mode) / vfo_shortx_kernel (X mode), each followed by the general kernel over the
    // profiles it left on its list (non-uniform altitude grid, fast-turning or vanishing field, negative density,
    // peak at level 0 or 1, a sum that is not finite)
    auto launch_short_kind = [&](bool xmode) -> int {
        const int n_kind = xmode ? n_shortx : n_short;
        if (n_kind == 0) return PRHF_OK;
        prhf::SegDev* kind_seg = xmode ? shortx_seg : short_seg;
        prhf::KArgs as = a;
        as.n_segs = n_kind;
        long long short_blocks = 0;
        for (int i = 0; i < n_kind; ++i) {
            kind_seg[i].block_begin = short_blocks;
            short_blocks += kind_seg[i].prof_end - kind_seg[i].prof_begin;
            as.seg[i] = kind_seg[i];
        }
        if (short_blocks > 0x7fffffffLL) return fail(PRHF_EINVAL, "launch too large");
        if (short_blocks == 0) return PRHF_OK;
        as.n_blocks = short_blocks;
        as.short_queue = 0;                        // (set with the geometry below)
        as.short_prio = (int)kn.short_prio;
        as.partial = nullptr;
        as.altmin = nullptr;
        as.trace = nullptr;
        as.queue = nullptr;
#ifdef PRHF_TRACE
        static DevBuf trace_short;
        const size_t short_trace_words = (size_t)short_blocks * kWavesPerBlock * 8;
        if (std::getenv("PRHF_TRACE_FILE") && !xmode) {
            int rct;
            if ((rct = ensure(c, trace_short, short_trace_words * 8)) != PRHF_OK) return rct;
            HIP_TRY(hipMemsetAsync(trace_short.p, 0, short_trace_words * 8, short_stream));
            as.trace = static_cast<unsigned long long*>(trace_short.p);
        }
#endif
        DevBuf& left = xmode ? c->leftover_x : c->leftover;
        int rcl;
        if ((rcl = ensure(c, left, (size_t)(short_blocks + 1) * sizeof(unsigned))) != PRHF_OK) return rcl;
        as.leftover = static_cast<unsigned*>(left.p);
        if (!zeroed_by_table) HIP_TRY(hipMemsetAsync(as.leftover, 0, sizeof(unsigned), short_stream));
        if (xmode) {
            // the compact geometry of the O kernel (four 4-wave workgroups per CU, staged arrays for as many levels as a
            // quarter of the LDS holds), taken on the same condition; a profile that peaks above them goes on a block
            // list of its own, which a second launch of this kernel with full-size arrays takes (a2 below, queue word 8)
            long long Lx = 0;
            if (kn.short_compact != 0) {
                long long L = lds_levels;
                while (L > 1 && prhf::shortx_lds_bytes(L, n_freq) > lds_quarter) --L;
                if (2 * L >= lds_levels && L >= 8 && prhf::shortx_lds_bytes(L, n_freq) <= lds_quarter) Lx = L;
            }
            const int threads = Lx > 0 ? PRHF_COMPACT_THREADS : PRHF_SHORT_THREADS;
            const bool second_x = Lx > 0 && Lx < lds_levels;      // some bottomsides may not fit the compact arrays
            if (Lx > 0) as.lds_levels = Lx;
            if (second_x) {
                if ((rcl = ensure(c, c->leftover_tall_x, (size_t)(short_blocks + 1) * sizeof(unsigned))) != PRHF_OK) return rcl;
                as.leftover_tall = static_cast<unsigned*>(c->leftover_tall_x.p);
                if (!zeroed_by_table) HIP_TRY(hipMemsetAsync(as.leftover_tall, 0, sizeof(unsigned), short_stream));
            }
            const size_t lds = prhf::shortx_lds_bytes(as.lds_levels, n_freq);
            const long long short_slots = (long long)c->cu_count * (Lx > 0 ? PRHF_COMPACT_WGS_PER_CU : (lds <= lds_half ? 2 : 1));
            long long grid_short = short_blocks;
            if (short_blocks > short_slots) {
                as.queue = c->d_status + 4;
                grid_short = short_slots;
            }
            HIP_TRY(prhf::launch_vfo_shortx(as, grid_short, lds, threads, short_stream));
            if (second_x) {
                // the profiles the compact launch left for full-size arrays: persistent workgroups read their number from
                // the device; what these leave - another input shape - joins the general list
                prhf::KArgs a2 = as;
                a2.lds_levels = lds_levels;
                a2.block_list = as.leftover_tall;
                a2.leftover_tall = nullptr;
                a2.queue = c->d_status + 8;
                const size_t lds2 = prhf::shortx_lds_bytes(lds_levels, n_freq);
                const long long slots2 = (long long)c->cu_count * (lds2 <= lds_half ? 2 : 1);
                HIP_TRY(prhf::launch_vfo_shortx(a2, std::min(short_blocks, slots2), lds2, PRHF_SHORT_THREADS, short_stream));
            }
        } else {
            const bool compact = compact_levels > 0;
            const bool second = compact && compact_levels < lds_levels;    // some bottomsides may not fit the compact arrays
            const int fixed_q = kShortQueueFixed > 0 ? -std::min(kShortQueueFixed, compact ? compact_queue : short_queue) : 0;
            if (second) {
                if ((rcl = ensure(c, c->leftover_tall, (size_t)(short_blocks + 1) * sizeof(unsigned))) != PRHF_OK) return rcl;
                as.leftover_tall = static_cast<unsigned*>(c->leftover_tall.p);
                if (!zeroed_by_table) HIP_TRY(hipMemsetAsync(as.leftover_tall, 0, sizeof(unsigned), short_stream));
            }
            const int threads = compact ? PRHF_COMPACT_THREADS : PRHF_SHORT_THREADS;
            // lanes per pair: eight on grids of up to 256 points (the launch's longest), sixteen beyond (Knobs::short_lanes)
            int longest = 0;
            for (int i = 0; i < n_kind; ++i) longest = std::max(longest, kind_seg[i].n_points);
            const int short_lanes = kn.short_lanes == 0 ? (longest <= 256 ? 8 : 16) : (kn.short_lanes < 12 ? 8 : 16);
            const int queue_entries = compact ? compact_queue : short_queue;
            as.lds_levels = compact ? compact_levels : lds_levels;
            as.short_queue = fixed_q ? fixed_q : queue_entries;
            const size_t lds = prhf::short_lds_fixed(as.lds_levels, n_freq, threads) + 8 * (size_t)queue_entries;
            const long long short_slots = (long long)c->cu_count * (compact ? PRHF_COMPACT_WGS_PER_CU : (lds <= lds_half ? 2 : 1));
            long long grid_short = short_blocks;
            if (short_blocks > short_slots) {
                as.queue = c->d_status + 2;
                grid_short = short_slots;
            }
            // from four resident rounds on, the blocks are drawn in descending order of a cost estimate
            as.order = order_made;             // (sorted beside the per-frequency table, or null: index order)
            HIP_TRY(prhf::launch_vfo_short(as, grid_short, lds, threads, short_lanes, short_stream));
            if (second) {
                // the profiles the compact launch left for full-size arrays: persistent workgroups read their number from
                // the device (3 us when there is none); what these leave - another input shape - joins the general list
                prhf::KArgs a2 = as;
                a2.trace = nullptr;
                a2.lds_levels = lds_levels;
                a2.short_queue = fixed_q ? -std::min(kShortQueueFixed, short_queue) : short_queue;
                a2.block_list = as.leftover_tall;
                a2.order = nullptr;
                a2.leftover_tall = nullptr;
                a2.queue = c->d_status + 0;
                const size_t lds2 = prhf::short_lds_fixed(lds_levels, n_freq, PRHF_SHORT_THREADS) + 8 * (size_t)short_queue;
                const long long slots2 = (long long)c->cu_count * (lds2 <= lds_half ? 2 : 1);
                HIP_TRY(prhf::launch_vfo_short(a2, std::min(short_blocks, slots2), lds2, PRHF_SHORT_THREADS, short_lanes, short_stream));
            }
        }
#ifdef PRHF_TRACE
        if (as.trace) {                            // eight wall-clock marks per wave and block (tools/wave_trace_short.py)
            std::vector<unsigned long long> host(short_trace_words);
            HIP_TRY(hipMemcpyAsync(host.data(), as.trace, short_trace_words * 8, hipMemcpyDeviceToHost, short_stream));
            HIP_TRY(hipStreamSynchronize(short_stream));
            if (FILE* fp = std::fopen(std::getenv("PRHF_TRACE_FILE"), "wb")) {
                std::fwrite(host.data(), 8, short_trace_words, fp);
                std::fclose(fp);
            }
        }
#endif
        prhf::KArgs af = as;
        af.trace = nullptr;
        af.lds_levels = lds_levels;
        af.leftover_tall = nullptr;
        af.order = nullptr;
        af.block_list = as.leftover;
        af.leftover = nullptr;
        af.queue = c->d_status + (xmode ? 5 : 3);
        af.zero_after = (!xmode && order_made) ? order_made : nullptr;
        HIP_TRY(prhf::launch_vfo(af, std::min(short_blocks, wg_slots), xmode ? 1 : 0, prhf::lds_bytes_for(lds_levels), short_stream));
        if (af.zero_after) c->order_clean = true;
        return PRHF_OK;
    };
    if ((rc = launch_short_kind(true)) != PRHF_OK) return rc;          // (the longer blocks first)
    if ((rc = launch_short_kind(false)) != PRHF_OK) return rc;
    if (forked) {
        HIP_TRY(hipEventRecord(c->join_ev, c->aux_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->join_ev, 0));
    } else {
        if ((rc = launch_general()) != PRHF_OK) return rc;
    }
#ifdef PRHF_TRACE
    if (a.trace) {
        std::vector<unsigned long long> host(trace_words);
        HIP_TRY(hipMemcpyAsync(host.data(), a.trace, trace_words * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (FILE* fp = std::fopen(std::getenv("PRHF_TRACE_FILE"), "wb")) {
            std::fwrite(host.data(), 8, trace_words, fp);
            std::fclose(fp);
        }
    }
#endif
    if (post) HIP_TRY(prhf::launch_residual(vh_dev, d_obs, n_prof, (int)n_freq, d_res, d_cost, c->stream));
    if (timed_launch) {
        HIP_TRY(hipEventRecord(c->pending_end_ev(), c->stream));
        c->mark_timed();
    }
    c->status_pending = true;

    // small results come back through the pinned buffer too (its upper half; the inputs of a call this small
    // fit the lower one) and are handed over after the synchronisation below
    const bool out_via_pack = !out_direct && !dev && out && out_elems && c->h_pack && out_elems * 8 <= kPackBytes / 4 &&
                              (size_t)(d_out - static_cast<double*>(c->arena.p)) * 8 <= kPackBytes / 2;
    double* h_out = c->h_pack ? c->h_pack + kPackBytes / 16 : nullptr;       // doubles: byte offset kPackBytes / 2
    if (out_direct) {
        // (written by the kernel itself)
    } else if (out_via_pack)
        HIP_TRY(hipMemcpyAsync(h_out, d_out, out_elems * 8, hipMemcpyDeviceToHost, c->stream));
    else if (!dev && out_elems && out)
        HIP_TRY(hipMemcpyAsync(out, d_out, out_elems * 8, hipMemcpyDeviceToHost, c->stream));
    if (!dev && post) {
        if (post->residual)
            HIP_TRY(hipMemcpyAsync(post->residual, d_res, out_elems * 8, hipMemcpyDeviceToHost, c->stream));
        if (post->cost)
            HIP_TRY(hipMemcpyAsync(post->cost, d_cost, (size_t)n_prof * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (flags & PRHF_FLAG_ASYNC) return PRHF_OK;
    rc = prhf_sync(c);
    if (out_via_pack || out_direct) std::memcpy(out, h_out, out_elems * 8);
    return rc;
}

// A large batch from HOST buffers (a NumPy caller with an ensemble): uploaded whole in front of one launch, the
// transfers - pageable memory, 10 - 30 GB/s - stood in front of and behind the kernel: 56.9 ms per call for a 41.2 ms
// kernel on config 4's shard (BENCH_r03 `host_buffers`).  Here the profiles go in three slabs of 10 %, 30 % and 60 %:
// slab k + 1 is uploaded (stream `up_stream`) while slab k is evaluated (the context's stream, through run() on the
// staged rows as device-resident input), and slab k's result rows travel back (`down_stream`) while slab k + 1 runs.
// A growing slab size keeps the first kernel's wait short and every later upload behind a kernel that is about as long.
// Values do not depend on the cut: a launch's rows do not depend on their neighbours (tests/test_gpu_full_size.py).
int run_host_slabs(prhf_ctx* c, const double* freq, int64_t n_freq, const double* den, const double* bmag,
                   const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt, int64_t prof_stride,
                   int64_t alt_stride, const double* mult, int32_t n_points, int32_t mode, double* out, uint32_t flags) {
    const bool shared_field = (flags & PRHF_FLAG_SHARED_FIELD) != 0;
    ENTER_DEVICE(c->device);
    const size_t n_alt_rows = alt_stride ? (size_t)n_prof : 1, n_field_rows = shared_field ? 1 : (size_t)n_prof;
    const size_t row_bytes = (size_t)n_alt * 8;
    const size_t elems = (size_t)n_freq + ((size_t)n_prof + 2 * n_field_rows + n_alt_rows) * (size_t)n_alt + (size_t)n_points +
                         (size_t)n_prof * (size_t)n_freq;
    int rc;
    if ((rc = ensure(c, c->arena, elems * 8)) != PRHF_OK) return rc;
    double* d_freq = static_cast<double*>(c->arena.p);
    double* d_den = d_freq + n_freq;
    double* d_bmag = d_den + (size_t)n_prof * n_alt;
    double* d_bpsi = d_bmag + n_field_rows * n_alt;
    double* d_alt = d_bpsi + n_field_rows * n_alt;
    double* d_mult = d_alt + n_alt_rows * n_alt;
    double* d_out = d_mult + n_points;
    // the arena may still be read by an earlier asynchronous launch of this context: the uploads wait for it
    HIP_TRY(hipEventRecord(c->slab_free, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->up_stream, c->slab_free, 0));
    HIP_TRY(hipMemcpyAsync(d_freq, freq, (size_t)n_freq * 8, hipMemcpyHostToDevice, c->up_stream));
    HIP_TRY(hipMemcpyAsync(d_mult, mult, (size_t)n_points * 8, hipMemcpyHostToDevice, c->up_stream));
    if (!alt_stride) HIP_TRY(hipMemcpyAsync(d_alt, alt, row_bytes, hipMemcpyHostToDevice, c->up_stream));
    if (shared_field) {
        HIP_TRY(hipMemcpyAsync(d_bmag, bmag, row_bytes, hipMemcpyHostToDevice, c->up_stream));
        HIP_TRY(hipMemcpyAsync(d_bpsi, bpsi, row_bytes, hipMemcpyHostToDevice, c->up_stream));
    }
    const int n_slabs = 3;
    const int64_t cut[4] = {0, std::max<int64_t>(1, n_prof / 10), std::max<int64_t>(2, (n_prof * 4) / 10), n_prof};
    const uint32_t dev_flags = PRHF_FLAG_DEVICE_PTRS | PRHF_FLAG_ASYNC | (shared_field ? PRHF_FLAG_SHARED_FIELD : 0);
    int first_error = PRHF_OK;
    for (int k = 0; k < n_slabs && first_error == PRHF_OK; ++k) {
        const int64_t p0 = cut[k], rows = cut[k + 1] - cut[k];
        if (rows <= 0) continue;
        auto up2d = [&](double* dst, const double* src, int64_t stride) {
            return hipMemcpy2DAsync(dst + (size_t)p0 * n_alt, row_bytes, src + (size_t)p0 * stride, (size_t)stride * 8, row_bytes,
                                    (size_t)rows, hipMemcpyHostToDevice, c->up_stream);
        };
        HIP_TRY(up2d(d_den, den, prof_stride));
        if (!shared_field) {
            HIP_TRY(up2d(d_bmag, bmag, prof_stride));
            HIP_TRY(up2d(d_bpsi, bpsi, prof_stride));
        }
        if (alt_stride) HIP_TRY(up2d(d_alt, alt, alt_stride));
        HIP_TRY(hipEventRecord(c->slab_up[k], c->up_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->slab_up[k], 0));
        prhf_segment seg;
        seg.prof_begin = 0;
        seg.prof_end = rows;
        seg.mode = mode;
        seg.n_points = n_points;
        seg.mult_offset = 0;
        seg.out_offset = 0;
        rc = run(c, d_freq, n_freq, d_den + (size_t)p0 * n_alt, shared_field ? d_bmag : d_bmag + (size_t)p0 * n_alt,
                 shared_field ? d_bpsi : d_bpsi + (size_t)p0 * n_alt, alt_stride ? d_alt + (size_t)p0 * n_alt : d_alt, rows, n_alt,
                 n_alt, alt_stride ? n_alt : 0, d_mult, n_points, &seg, 1, d_out + (size_t)p0 * n_freq, dev_flags);
        if (rc != PRHF_OK) { first_error = rc; break; }
        HIP_TRY(hipEventRecord(c->slab_done[k], c->stream));
        // the rows of the slab before this one go home while this one runs (issued here, behind this slab's upload:
        // a copy from or to pageable memory holds the calling thread, and the upload is what the next kernel waits for)
        if (k > 0) {
            HIP_TRY(hipStreamWaitEvent(c->down_stream, c->slab_done[k - 1], 0));
            HIP_TRY(hipMemcpyAsync(out + (size_t)cut[k - 1] * n_freq, d_out + (size_t)cut[k - 1] * n_freq,
                                   (size_t)(cut[k] - cut[k - 1]) * n_freq * 8, hipMemcpyDeviceToHost, c->down_stream));
        }
    }
    if (first_error == PRHF_OK) {
        HIP_TRY(hipStreamWaitEvent(c->down_stream, c->slab_done[n_slabs - 1], 0));
        HIP_TRY(hipMemcpyAsync(out + (size_t)cut[n_slabs - 1] * n_freq, d_out + (size_t)cut[n_slabs - 1] * n_freq,
                               (size_t)(cut[n_slabs] - cut[n_slabs - 1]) * n_freq * 8, hipMemcpyDeviceToHost, c->down_stream));
    }
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    const int rc_sync = prhf_sync(c);                  // the context's stream + the launches' status words
    HIP_TRY(hipStreamSynchronize(c->down_stream));
    return first_error != PRHF_OK ? first_error : rc_sync;
}

}  // namespace

extern "C" {

int prhf_abi_version(void) { return PRHF_ABI_VERSION; }

const char* prhf_last_error(void) { return g_err.c_str(); }

int prhf_device_count(int* n) {
    if (!n) return fail(PRHF_EINVAL, "null pointer");
    *n = 0;
    HIP_TRY(hipGetDeviceCount(n));
    return PRHF_OK;
}

int prhf_ctx_create(int device, prhf_ctx** out) {
    if (!out) return fail(PRHF_EINVAL, "null pointer");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(PRHF_EINVAL, "device %d not in [0, %d)", device, n);
    ENTER_DEVICE(device);
    prhf_ctx* c = new (std::nothrow) prhf_ctx;
    if (!c) return fail(PRHF_ENOMEM, "out of host memory");
    c->device = device;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->slab_free, hipEventDisableTiming)) != hipSuccess ||
        (e = create_events_untimed(c->slab_up, 3)) != hipSuccess ||
        (e = create_events_untimed(c->slab_done, 3)) != hipSuccess ||
        (e = create_events(c->ring0, prhf_ctx::kTimingRing)) != hipSuccess ||
        (e = create_events(c->ring1, prhf_ctx::kTimingRing)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&c->d_status), kStatusWords * sizeof(unsigned))) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&c->h_status), PRHF_STATUS_WORDS * sizeof(unsigned),
                           hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_status_dev), c->h_status, 0)) != hipSuccess ||
        (e = hipMemset(c->d_status, 0, kStatusWords * sizeof(unsigned))) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&c->h_pack), kPackBytes, hipHostMallocMapped | hipHostMallocCoherent)) !=
            hipSuccess ||
        (e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_pack_dev), c->h_pack, 0)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&c->d_words), 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&c->h_words), 2 * sizeof(unsigned long long),
                           hipHostMallocDefault)) != hipSuccess ||
        (e = prhf::configure_kernels(prhf::lds_bytes_for(kMaxAlt))) != hipSuccess) {
        prhf_ctx_destroy(c);
        return fail(PRHF_EHIP, "context setup failed: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    std::memset(c->h_status, 0, PRHF_STATUS_WORDS * sizeof(unsigned));
#ifdef PRHF_DIAG
    // diagnostics build only: PRHF_<OPTION NAME IN CAPITALS>=value presets the options of every new context
    for (const KnobName& k : kKnobNames) {
        std::string env = "PRHF_";
        for (const char* p = k.name; *p; ++p) env += (char)std::toupper((unsigned char)*p);
        if (const char* v = std::getenv(env.c_str())) c->knobs.*(k.field) = std::min(k.hi, std::max(k.lo, std::atof(v)));
    }
#endif
    (void)hipDeviceGetAttribute(&c->cu_count, hipDeviceAttributeMultiprocessorCount, device);
    if (hipDeviceGetAttribute(&c->large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) c->large_bar = 0;
    if (c->cu_count < 1) c->cu_count = 256;
    *out = c;
    return PRHF_OK;
}

int prhf_ctx_destroy(prhf_ctx* c) {
    if (!c) return PRHF_OK;
    DeviceScope device_scope_(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->arena.p) (void)hipFree(c->arena.p);
    if (c->partial.p) (void)hipFree(c->partial.p);
    if (c->altmin.p) (void)hipFree(c->altmin.p);
    if (c->pairs.p) (void)hipFree(c->pairs.p);
    if (c->ftab.p) (void)hipFree(c->ftab.p);
    if (c->levels.p) (void)hipFree(c->levels.p);
    if (c->ptab.p) (void)hipFree(c->ptab.p);
    if (c->leftover.p) (void)hipFree(c->leftover.p);
    if (c->leftover_x.p) (void)hipFree(c->leftover_x.p);
    if (c->leftover_tall.p) (void)hipFree(c->leftover_tall.p);
    if (c->leftover_tall_x.p) (void)hipFree(c->leftover_tall_x.p);
    if (c->order.p) (void)hipFree(c->order.p);
    if (c->tall.p) (void)hipFree(c->tall.p);
    for (int g = 0; g < c->n_host_grids; ++g) {
        if (c->host_grid[g].mult.p) (void)hipFree(c->host_grid[g].mult.p);
        if (c->host_grid[g].pairs.p) (void)hipFree(c->host_grid[g].pairs.p);
    }
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    if (c->d_words) (void)hipFree(c->d_words);
    if (c->h_words) (void)hipHostFree(c->h_words);
    for (int i = 0; i < prhf_ctx::kTimingRing; ++i) {
        if (c->ring0[i]) (void)hipEventDestroy(c->ring0[i]);
        if (c->ring1[i]) (void)hipEventDestroy(c->ring1[i]);
    }
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->up_stream) { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamDestroy(c->up_stream); }
    if (c->down_stream) { (void)hipStreamSynchronize(c->down_stream); (void)hipStreamDestroy(c->down_stream); }
    for (int i = 0; i < 3; ++i) {
        if (c->slab_up[i]) (void)hipEventDestroy(c->slab_up[i]);
        if (c->slab_done[i]) (void)hipEventDestroy(c->slab_done[i]);
    }
    if (c->slab_free) (void)hipEventDestroy(c->slab_free);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PRHF_OK;
}

int prhf_ctx_set_stream(prhf_ctx* c, void* hip_stream, int32_t borrow) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    // a borrowed NULL is the legacy default stream (what torch reports for its default stream)
    hipStream_t next = borrow ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    if (next == c->stream) return PRHF_OK;
    ENTER_DEVICE(c->device);
    // Scratch buffers are reused across launches: work on the new stream must come after the last launch on
    // the old one.  Ordered through the event recorded behind that launch, not by synchronising the old
    // stream - a borrowed stream may have been destroyed by its owner since.
    if (c->timed) HIP_TRY(hipStreamWaitEvent(next, c->end_ev(), 0));
    c->stream = next;
    return PRHF_OK;
}

int prhf_ctx_set_option(prhf_ctx* c, const char* name, double value) {
    if (!c || !name) return fail(PRHF_EINVAL, "null pointer");
    for (const KnobName& k : kKnobNames)
        if (std::strcmp(k.name, name) == 0) {
            if (!(value >= k.lo && value <= k.hi))
                return fail(PRHF_EINVAL, "option %s: %g outside [%g, %g]", name, value, k.lo, k.hi);
            c->knobs.*(k.field) = value;
            return PRHF_OK;
        }
    return fail(PRHF_EINVAL, "unknown option '%s'", name);
}

int prhf_ctx_set_math(prhf_ctx* c, int level) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (level != PRHF_MATH_FAITHFUL && level != PRHF_MATH_FAST && level != PRHF_MATH_AUTO)
        return fail(PRHF_EINVAL, "unknown math tier");
    c->math = level;
    return PRHF_OK;
}

int prhf_vfo_batch_f64(prhf_ctx* ctx, const double* freq_mhz, int64_t n_freq, const double* den,
                       const double* bmag, const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                       int64_t prof_stride_elems, int64_t alt_stride_elems, const double* multiplier,
                       int32_t n_points, int32_t mode, double* vh_out, uint32_t flags) {
    prhf_segment seg;
    seg.prof_begin = 0;
    seg.prof_end = n_prof;
    seg.mode = mode;
    seg.n_points = n_points;
    seg.mult_offset = 0;
    seg.out_offset = 0;
    // a large batch from host buffers: in slabs, transfers beside the kernels (run_host_slabs).  The checks of run()
    // that concern the whole call are made by its first slab; shapes it would refuse are left to it here too.
    if (ctx && !(flags & PRHF_FLAG_DEVICE_PTRS) && ctx->knobs.host_slabs >= 3 && freq_mhz && den && bmag && bpsi && alt &&
        multiplier && vh_out && n_freq >= 1 && n_alt >= 1 && n_points >= 1 && n_prof >= 64 &&
        prof_stride_elems >= n_alt && (alt_stride_elems == 0 || alt_stride_elems >= n_alt) &&
        (mode == PRHF_MODE_O || mode == PRHF_MODE_X) && !(flags & ~(PRHF_FLAG_GRID_STABLE | PRHF_FLAG_SHARED_FIELD)) &&
        (size_t)n_prof * (size_t)n_alt * 24 >= kSlabMinBytes && n_alt <= kMaxAltTall && n_freq <= (1 << 20)) {
        for (int64_t i = 1; i < n_points; ++i)
            if (multiplier[i] < multiplier[i - 1])
                return fail(PRHF_EINVAL, "multiplier[%lld] decreases: the stretched grid must be non-decreasing", (long long)i);
        return run_host_slabs(ctx, freq_mhz, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride_elems, alt_stride_elems,
                              multiplier, n_points, mode, vh_out, flags);
    }
    return run(ctx, freq_mhz, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride_elems, alt_stride_elems,
               multiplier, n_points, &seg, 1, vh_out, flags);
}

int prhf_vfo_worklist_f64(prhf_ctx* ctx, const double* freq_mhz, int64_t n_freq, const double* den,
                          const double* bmag, const double* bpsi, const double* alt, int64_t n_prof,
                          int64_t n_alt, int64_t prof_stride_elems, int64_t alt_stride_elems,
                          const double* multiplier, int64_t multiplier_len, const prhf_segment* segs,
                          int32_t n_segs, double* vh_out, uint32_t flags) {
    return run(ctx, freq_mhz, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride_elems, alt_stride_elems,
               multiplier, multiplier_len, segs, n_segs, vh_out, flags);
}

int prhf_vfo_residual_f64(prhf_ctx* ctx, const double* freq_mhz, int64_t n_freq, const double* den,
                          const double* bmag, const double* bpsi, const double* alt, int64_t n_prof,
                          int64_t n_alt, int64_t prof_stride_elems, int64_t alt_stride_elems,
                          const double* multiplier, int32_t n_points, int32_t mode, const double* vh_obs,
                          double* vh_out, double* residual_out, double* cost_out, uint32_t flags) {
    prhf_segment seg;
    seg.prof_begin = 0;
    seg.prof_end = n_prof;
    seg.mode = mode;
    seg.n_points = n_points;
    seg.mult_offset = 0;
    seg.out_offset = 0;
    Residual post{vh_obs, residual_out, cost_out};
    return run(ctx, freq_mhz, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride_elems, alt_stride_elems,
               multiplier, n_points, &seg, 1, vh_out, flags, &post);
}

int prhf_mu_mup_f64(prhf_ctx* c, const double* X, const double* Y, const double* psi_deg, int64_t n,
                    int32_t mode, double* mu_out, double* mup_out, uint32_t flags) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (!X || !Y || !psi_deg || !mu_out || !mup_out) return fail(PRHF_EINVAL, "null array pointer");
    if (n < 0) return fail(PRHF_EINVAL, "bad shape");
    if (mode != PRHF_MODE_O && mode != PRHF_MODE_X) return fail(PRHF_EINVAL, "Mode must be O or X");
    if (flags & ~PRHF_FLAG_DEVICE_PTRS) return fail(PRHF_EINVAL, "unknown flag bits");
    if (n == 0) return PRHF_OK;
    ENTER_DEVICE(c->device);
    const bool dev = (flags & PRHF_FLAG_DEVICE_PTRS) != 0;
    const double *dX = X, *dY = Y, *dP = psi_deg;
    double *dMu = mu_out, *dMup = mup_out;
    const size_t bytes = (size_t)n * 8;
    if (!dev) {
        int rc = ensure(c, c->arena, 5 * bytes);
        if (rc != PRHF_OK) return rc;
        double* base = static_cast<double*>(c->arena.p);
        HIP_TRY(hipMemcpyAsync(base, X, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(base + n, Y, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(base + 2 * n, psi_deg, bytes, hipMemcpyHostToDevice, c->stream));
        dX = base; dY = base + n; dP = base + 2 * n; dMu = base + 3 * n; dMup = base + 4 * n;
    }
    HIP_TRY(hipEventRecord(c->begin_ev(), c->stream));
    HIP_TRY(prhf::launch_mu_mup(dX, dY, dP, n, mode == PRHF_MODE_O ? PRHF_KMODE_O : PRHF_KMODE_X,
                                c->math == PRHF_MATH_FAST ? 1 : 0, c->d_words, c->h_words, dMu, dMup, c->stream));
    HIP_TRY(hipEventRecord(c->pending_end_ev(), c->stream));
    c->mark_timed();
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(mu_out, dMu, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mup_out, dMup, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PRHF_OK;
}

int prhf_find_vh_f64(prhf_ctx* c, const double* X, const double* Y, const double* psi_deg, const double* dh,
                     int64_t n_rows, int64_t n_cols, double alt_min, int32_t mode, double* vh_out,
                     uint32_t flags) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (!X || !Y || !psi_deg || !dh || !vh_out) return fail(PRHF_EINVAL, "null array pointer");
    if (n_rows < 0 || n_cols < 0) return fail(PRHF_EINVAL, "bad shape");
    if (mode != PRHF_MODE_O && mode != PRHF_MODE_X) return fail(PRHF_EINVAL, "Mode must be O or X");
    if (flags & ~PRHF_FLAG_DEVICE_PTRS) return fail(PRHF_EINVAL, "unknown flag bits");
    if (n_rows == 0) return PRHF_OK;
    ENTER_DEVICE(c->device);
    const bool dev = (flags & PRHF_FLAG_DEVICE_PTRS) != 0;
    const int64_t n = n_rows * n_cols;
    const double *dX = X, *dY = Y, *dP = psi_deg, *dD = dh;
    double* dV = vh_out;
    if (!dev) {
        int rc = ensure(c, c->arena, (size_t)(4 * n + n_rows) * 8);
        if (rc != PRHF_OK) return rc;
        double* base = static_cast<double*>(c->arena.p);
        const double* src[4] = {X, Y, psi_deg, dh};
        for (int k = 0; k < 4 && n > 0; ++k)
            HIP_TRY(hipMemcpyAsync(base + k * n, src[k], (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        dX = base; dY = base + n; dP = base + 2 * n; dD = base + 3 * n; dV = base + 4 * n;
    }
    HIP_TRY(hipEventRecord(c->begin_ev(), c->stream));
    HIP_TRY(prhf::launch_find_vh(dX, dY, dP, dD, n_rows, n_cols, alt_min,
                                 mode == PRHF_MODE_O ? PRHF_KMODE_O : PRHF_KMODE_X,
                                 c->math == PRHF_MATH_FAST ? 1 : 0, c->d_words, c->h_words, dV, c->stream));
    HIP_TRY(hipEventRecord(c->pending_end_ev(), c->stream));
    c->mark_timed();
    if (!dev) HIP_TRY(hipMemcpyAsync(vh_out, dV, (size_t)n_rows * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PRHF_OK;
}

int prhf_regrid_f64(prhf_ctx* c, const double* freq_hz, int64_t n_freq, const double* den, const double* bmag,
                    const double* bpsi, const double* alt, int64_t n_alt, const double* multiplier,
                    int32_t n_points, int32_t mode, double* out_freq, double* out_den, double* out_bmag,
                    double* out_bpsi, double* out_dist, double* out_alt, double* out_crit, int64_t* out_ind,
                    uint32_t flags) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (!freq_hz || !den || !bmag || !bpsi || !alt || !multiplier || !out_freq || !out_den || !out_bmag ||
        !out_bpsi || !out_dist || !out_alt || !out_crit || !out_ind)
        return fail(PRHF_EINVAL, "null array pointer");
    if (n_freq < 1 || n_alt < 1 || n_alt > kMaxAlt || n_points < 1) return fail(PRHF_EINVAL, "bad shape");
    if (mode != PRHF_MODE_O && mode != PRHF_MODE_X) return fail(PRHF_EINVAL, "mode must be 'O' or 'X'");
    if (flags & ~PRHF_FLAG_DEVICE_PTRS) return fail(PRHF_EINVAL, "unknown flag bits");
    ENTER_DEVICE(c->device);
    const bool dev = (flags & PRHF_FLAG_DEVICE_PTRS) != 0;
    const size_t fn = (size_t)n_freq * (size_t)n_points;
    prhf::RegridArgs a;
    std::memset(&a, 0, sizeof a);
    a.n_freq = n_freq; a.n_alt = n_alt; a.n_points = n_points;
    a.mode = mode == PRHF_MODE_O ? PRHF_KMODE_O : PRHF_KMODE_X;
    a.status = c->h_status_dev;
    double* base = nullptr;
    if (dev) {
        a.freq_hz = freq_hz; a.den = den; a.bmag = bmag; a.bpsi = bpsi; a.alt = alt; a.mult = multiplier;
        a.out_freq = out_freq; a.out_den = out_den; a.out_bmag = out_bmag; a.out_bpsi = out_bpsi;
        a.out_dist = out_dist; a.out_alt = out_alt; a.out_crit = out_crit;
        a.out_ind = reinterpret_cast<long long*>(out_ind);
    } else {
        const size_t in_elems = (size_t)n_freq + 4 * (size_t)n_alt + (size_t)n_points;
        int rc = ensure(c, c->arena, (in_elems + 8 * fn) * 8);
        if (rc != PRHF_OK) return rc;
        base = static_cast<double*>(c->arena.p);
        double* d_freq = base;
        double* d_den = d_freq + n_freq;
        double* d_bmag = d_den + n_alt;
        double* d_bpsi = d_bmag + n_alt;
        double* d_alt = d_bpsi + n_alt;
        double* d_mult = d_alt + n_alt;
        double* d_out = d_mult + n_points;
        HIP_TRY(hipMemcpyAsync(d_freq, freq_hz, (size_t)n_freq * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_den, den, (size_t)n_alt * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_bmag, bmag, (size_t)n_alt * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_bpsi, bpsi, (size_t)n_alt * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_alt, alt, (size_t)n_alt * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_mult, multiplier, (size_t)n_points * 8, hipMemcpyHostToDevice, c->stream));
        a.freq_hz = d_freq; a.den = d_den; a.bmag = d_bmag; a.bpsi = d_bpsi; a.alt = d_alt; a.mult = d_mult;
        a.out_freq = d_out; a.out_den = d_out + fn; a.out_bmag = d_out + 2 * fn; a.out_bpsi = d_out + 3 * fn;
        a.out_dist = d_out + 4 * fn; a.out_alt = d_out + 5 * fn; a.out_crit = d_out + 6 * fn;
        a.out_ind = reinterpret_cast<long long*>(d_out + 7 * fn);
    }
    HIP_TRY(hipEventRecord(c->begin_ev(), c->stream));
    HIP_TRY(prhf::launch_regrid(a, prhf::lds_bytes_for(n_alt), c->stream));
    HIP_TRY(hipEventRecord(c->pending_end_ev(), c->stream));
    c->mark_timed();
    c->status_pending = true;
    if (!dev) {
        double* host[7] = {out_freq, out_den, out_bmag, out_bpsi, out_dist, out_alt, out_crit};
        double* devp[7] = {a.out_freq, a.out_den, a.out_bmag, a.out_bpsi, a.out_dist, a.out_alt, a.out_crit};
        for (int k = 0; k < 7; ++k)
            HIP_TRY(hipMemcpyAsync(host[k], devp[k], fn * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_ind, a.out_ind, fn * 8, hipMemcpyDeviceToHost, c->stream));
    }
    return prhf_sync(c);
}

int prhf_residual_f64(prhf_ctx* c, const double* vh_model, const double* vh_obs, int64_t n_prof, int64_t n_freq,
                      double* residual_out, double* cost_out, uint32_t flags) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (!vh_model || !vh_obs || (!residual_out && !cost_out)) return fail(PRHF_EINVAL, "null array pointer");
    if (n_prof < 0 || n_freq < 1 || n_freq > (1 << 20)) return fail(PRHF_EINVAL, "bad shape");
    if (flags & ~(PRHF_FLAG_DEVICE_PTRS | PRHF_FLAG_ASYNC | PRHF_FLAG_GRID_STABLE))
        return fail(PRHF_EINVAL, "unknown flag bits");
    const bool dev = (flags & PRHF_FLAG_DEVICE_PTRS) != 0;
    if ((flags & (PRHF_FLAG_ASYNC | PRHF_FLAG_GRID_STABLE)) && !dev)
        return fail(PRHF_EINVAL, "PRHF_FLAG_ASYNC and PRHF_FLAG_GRID_STABLE need device pointers");
    if (n_prof == 0) return PRHF_OK;
    ENTER_DEVICE(c->device);
    const size_t pf = (size_t)n_prof * (size_t)n_freq;
    const double *dM = vh_model, *dO = vh_obs;
    double *dR = residual_out, *dC = cost_out;
    if (!dev) {
        int rc = ensure(c, c->arena, (2 * pf + (size_t)n_freq + (size_t)n_prof) * 8);
        if (rc != PRHF_OK) return rc;
        double* base = static_cast<double*>(c->arena.p);
        HIP_TRY(hipMemcpyAsync(base, vh_model, pf * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(base + pf, vh_obs, (size_t)n_freq * 8, hipMemcpyHostToDevice, c->stream));
        dM = base; dO = base + pf;
        dR = residual_out ? base + pf + n_freq : nullptr;
        dC = cost_out ? base + 2 * pf + n_freq : nullptr;
    }
    HIP_TRY(hipEventRecord(c->begin_ev(), c->stream));
    HIP_TRY(prhf::launch_residual(dM, dO, n_prof, (int)n_freq, dR, dC, c->stream));
    HIP_TRY(hipEventRecord(c->pending_end_ev(), c->stream));
    c->mark_timed();
    if (!dev) {
        if (residual_out) HIP_TRY(hipMemcpyAsync(residual_out, dR, pf * 8, hipMemcpyDeviceToHost, c->stream));
        if (cost_out) HIP_TRY(hipMemcpyAsync(cost_out, dC, (size_t)n_prof * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (flags & PRHF_FLAG_ASYNC) return PRHF_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PRHF_OK;
}

namespace {
struct SnellGeometry {
    int geometry;
    double earth_radius_km, dz_target_km, apex_boost;
    int max_substeps;
};
// Grouped launch: freq_hz / profile_index describe n_groups (profile, frequency) groups and ray_group[r] names the
// group of ray r; ungrouped: ray_group == nullptr, n_groups == 0 and freq_hz / profile_index are per ray.
int snell_run(prhf_ctx* c, const SnellGeometry& geo, const double* freq_hz, const double* elevation_deg,
              const int64_t* profile_index, int64_t n_rays, const double* den, const double* bmag,
              const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt, int64_t alt_stride_elems,
              int32_t mode, double* out, double* path_x, double* path_z, int64_t path_stride, uint32_t flags,
              const int64_t* ray_group = nullptr, int64_t n_groups = 0) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    if (!freq_hz || !elevation_deg || !den || !bmag || !bpsi || !alt || !out)
        return fail(PRHF_EINVAL, "null array pointer");
    const bool grouped = ray_group != nullptr;
    if (grouped && n_groups < 1) return fail(PRHF_EINVAL, "a grouped launch needs at least one group");
    const int64_t n_keys = grouped ? n_groups : n_rays;          // entries of freq_hz / profile_index
    // per group and level: mu' (8 B), the compacted entry (32 B), its grid level (4 B); per group: four scalars
    const size_t level_cells = grouped ? (size_t)n_groups * (size_t)(n_alt + 1) : 0;
    const size_t mup_cells = (level_cells + 1) & ~(size_t)1;      // (the entries behind them are read 16 bytes at a time)
    const size_t level_bytes = mup_cells * 8 + level_cells * 36 + (grouped ? (size_t)n_groups * 16 : 0);
    if (level_bytes > ((size_t)64 << 30) || n_groups > 0x7fffffffLL || (grouped && n_rays > 0x7fffffffLL))
        return fail(PRHF_EINVAL, "level tables of %lld groups exceed 64 GiB (or 2^31 - 1 groups / rays): trace in batches",
                    (long long)n_groups);
    if (n_rays < 0 || n_prof < 1 || n_alt < 2 || n_alt > 3000) return fail(PRHF_EINVAL, "bad shape");
    if (grouped && n_prof > 0x7fffffffLL) return fail(PRHF_EINVAL, "a grouped launch takes at most 2^31 - 1 profiles");
    if ((path_x == nullptr) != (path_z == nullptr)) return fail(PRHF_EINVAL, "path_x and path_z go together");
    if (path_x && path_stride < 2 * (n_alt + 1) - 1)
        return fail(PRHF_EINVAL, "path_stride must hold 2 (n_alt + 1) - 1 nodes");
    if (mode != PRHF_MODE_O && mode != PRHF_MODE_X) return fail(PRHF_EINVAL, "Mode must be O or X");
    if (alt_stride_elems != 0 && alt_stride_elems != n_alt) return fail(PRHF_EINVAL, "alt stride is 0 or n_alt");
    if (flags & ~PRHF_FLAG_DEVICE_PTRS) return fail(PRHF_EINVAL, "unknown flag bits");
    if (n_rays == 0) return PRHF_OK;
    const bool dev = (flags & PRHF_FLAG_DEVICE_PTRS) != 0;
    if (!dev && profile_index)
        for (int64_t r = 0; r < n_keys; ++r)
            if (profile_index[r] < 0 || profile_index[r] >= n_prof)
                return fail(PRHF_EINVAL, "profile_index[%lld] outside [0, n_prof)", (long long)r);
    if (!dev && grouped)
        for (int64_t r = 0; r < n_rays; ++r)
            if (ray_group[r] < 0 || ray_group[r] >= n_groups)
                return fail(PRHF_EINVAL, "ray_group[%lld] outside [0, n_groups)", (long long)r);
    ENTER_DEVICE(c->device);
    prhf::SnellArgs a;
    std::memset(&a, 0, sizeof a);
    a.n_rays = n_rays; a.n_alt = n_alt; a.prof_stride = n_alt; a.alt_stride = alt_stride_elems;
    a.path_stride = path_x ? path_stride : 0;
    a.mode = mode == PRHF_MODE_O ? PRHF_KMODE_O : PRHF_KMODE_X;
    a.geometry = geo.geometry;
    a.reduced = (c->math == PRHF_MATH_FAITHFUL) ? 0 : 1;        // (grouped launches read faithful level tables either way)
    a.earth_radius_km = geo.earth_radius_km;
    a.dz_target_km = geo.dz_target_km;
    a.apex_boost = geo.apex_boost;
    a.max_substeps = geo.max_substeps;
    a.status = c->h_status_dev;
    const size_t prof_elems = (size_t)n_prof * (size_t)n_alt;
    const size_t alt_elems = alt_stride_elems ? prof_elems : (size_t)n_alt;
    const size_t path_elems = path_x ? (size_t)n_rays * (size_t)path_stride : 0;
    const double* d_keyf = nullptr;
    const long long* d_keyp = nullptr;
    const long long* d_group = nullptr;
    bool small_out = false;                    // a small host-buffer call: the kernel writes its results into pinned host memory
    if (dev) {
        a.den = den; a.bmag = bmag; a.bpsi = bpsi; a.alt = alt; a.elev_deg = elevation_deg;
        d_keyf = freq_hz;
        d_keyp = reinterpret_cast<const long long*>(profile_index);
        d_group = reinterpret_cast<const long long*>(ray_group);
        a.out = out; a.path_x = path_x; a.path_z = path_z;
    } else {
        const size_t elems = 3 * prof_elems + alt_elems + 2 * (size_t)n_keys + 2 * (size_t)n_rays +
                             PRHF_SNELL_OUTPUTS * (size_t)n_rays + 2 * path_elems;
        int rc = ensure(c, c->arena, elems * 8);
        if (rc != PRHF_OK) return rc;
        double* p = static_cast<double*>(c->arena.p);
        double* d_den = p; p += prof_elems;
        double* d_bmag = p; p += prof_elems;
        double* d_bpsi = p; p += prof_elems;
        double* d_alt = p; p += alt_elems;
        double* d_f = p; p += n_keys;
        long long* d_i = reinterpret_cast<long long*>(p); p += n_keys;
        double* d_e = p; p += n_rays;
        long long* d_g = reinterpret_cast<long long*>(p); p += n_rays;
        double* d_out = p; p += PRHF_SNELL_OUTPUTS * (size_t)n_rays;
        double* d_px = path_x ? p : nullptr; p += path_elems;
        double* d_pz = path_x ? p : nullptr;
        // A small call (the reference's own: ONE ray, its path arrays back): six to eight uploads from pageable memory
        // and three copies back cost several times the kernels.  As in run(): the inputs are packed in the arena's own
        // order and sent in one piece - or, on a large-BAR device with the arena idle, written straight into it by the
        // CPU - and the kernel writes the results into pinned host memory that the device sees (the upper half of the
        // pack buffer).
        double* base = static_cast<double*>(c->arena.p);
        const size_t in_elems = (size_t)(d_out - base);
        const size_t out_elems = PRHF_SNELL_OUTPUTS * (size_t)n_rays + 2 * path_elems;
        small_out = c->h_pack && in_elems * 8 <= kPackBytes / 2 && out_elems * 8 <= kPackBytes / 4;
        if (small_out) {
            const bool direct = c->large_bar && c->knobs.direct_upload != 0 && in_elems * 8 <= kDirectBytes &&
                                hipStreamQuery(c->stream) == hipSuccess &&
                                (!c->timed || hipEventQuery(c->end_ev()) == hipSuccess);
            (void)hipGetLastError();               // (hipErrorNotReady from the two queries is not an error)
            double* h = direct ? base : c->h_pack;
            std::memcpy(h + (d_den - base), den, prof_elems * 8);
            std::memcpy(h + (d_bmag - base), bmag, prof_elems * 8);
            std::memcpy(h + (d_bpsi - base), bpsi, prof_elems * 8);
            std::memcpy(h + (d_alt - base), alt, alt_elems * 8);
            std::memcpy(h + (d_f - base), freq_hz, (size_t)n_keys * 8);
            std::memcpy(h + (d_e - base), elevation_deg, (size_t)n_rays * 8);
            if (profile_index) std::memcpy(h + (reinterpret_cast<double*>(d_i) - base), profile_index, (size_t)n_keys * 8);
            if (grouped) std::memcpy(h + (reinterpret_cast<double*>(d_g) - base), ray_group, (size_t)n_rays * 8);
            if (direct) _mm_sfence();
            else HIP_TRY(hipMemcpyAsync(base, h, in_elems * 8, hipMemcpyHostToDevice, c->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(d_den, den, prof_elems * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_bmag, bmag, prof_elems * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_bpsi, bpsi, prof_elems * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_alt, alt, alt_elems * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_f, freq_hz, (size_t)n_keys * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_e, elevation_deg, (size_t)n_rays * 8, hipMemcpyHostToDevice, c->stream));
            if (profile_index)
                HIP_TRY(hipMemcpyAsync(d_i, profile_index, (size_t)n_keys * 8, hipMemcpyHostToDevice, c->stream));
            if (grouped)
                HIP_TRY(hipMemcpyAsync(d_g, ray_group, (size_t)n_rays * 8, hipMemcpyHostToDevice, c->stream));
        }
        a.den = d_den; a.bmag = d_bmag; a.bpsi = d_bpsi; a.alt = d_alt; a.elev_deg = d_e;
        d_keyf = d_f;
        d_keyp = profile_index ? d_i : nullptr;
        d_group = grouped ? d_g : nullptr;
        a.out = d_out; a.path_x = d_px; a.path_z = d_pz;
        if (small_out) {
            a.out = c->h_pack_dev + kPackBytes / 16;               // doubles: byte offset kPackBytes / 2
            a.path_x = path_x ? a.out + PRHF_SNELL_OUTPUTS * (size_t)n_rays : nullptr;
            a.path_z = path_x ? a.path_x + path_elems : nullptr;
        }
    }
    if (grouped) {
        a.ray_group = d_group; a.group_freq = d_keyf; a.group_prof = d_keyp; a.n_groups = n_groups;
        int rc3 = ensure(c, c->levels, level_bytes);
        if (rc3 != PRHF_OK) return rc3;
        a.levels = static_cast<double*>(c->levels.p);
        a.group_entries = a.levels + mup_cells;
        a.group_info = reinterpret_cast<int*>(a.group_entries + 4 * level_cells);
        a.group_kidx = a.group_info + 4 * (size_t)n_groups;
    } else {
        a.freq_hz = d_keyf; a.prof_idx = d_keyp;
        // per-ray launch: persistent wavefronts drawing rays from a queue
        a.ray_queue = nullptr;                                     // (set below: behind the per-profile scalars)
    }
    a.resident_cus = c->cu_count;
    {
        // per-profile scalars (the operator's chunk scratch is free here); behind them the per-ray launch's queue counters
        const size_t info_bytes = (((size_t)n_prof * 32 + 127) / 128) * 128;
        int rc2 = ensure(c, c->partial, info_bytes + prhf::snell_queue_bytes());
        if (rc2 != PRHF_OK) return rc2;
        a.prof_info = static_cast<double*>(c->partial.p);
        a.n_prof = n_prof;
        if (!grouped) a.ray_queue = reinterpret_cast<unsigned*>(static_cast<char*>(c->partial.p) + info_bytes);
    }
    a.ptab = nullptr;
    if (c->knobs.snell_table > 0 && (double)n_keys >= c->knobs.snell_table * (double)n_prof && prof_elems * 32 <= ((size_t)1 << 30)) {
        int rc4 = ensure(c, c->ptab, prof_elems * 32);
        if (rc4 != PRHF_OK) return rc4;
        a.ptab = static_cast<double*>(c->ptab.p);
    }
    HIP_TRY(hipEventRecord(c->begin_ev(), c->stream));
    HIP_TRY(prhf::launch_snell(a, c->stream));
    HIP_TRY(hipEventRecord(c->pending_end_ev(), c->stream));
    c->mark_timed();
    c->status_pending = true;
    if (small_out) {                                               // (written by the kernel itself into pinned memory)
        const int rc = prhf_sync(c);
        const double* h_out = c->h_pack + kPackBytes / 16;
        std::memcpy(out, h_out, PRHF_SNELL_OUTPUTS * (size_t)n_rays * 8);
        if (path_x) {
            std::memcpy(path_x, h_out + PRHF_SNELL_OUTPUTS * (size_t)n_rays, path_elems * 8);
            std::memcpy(path_z, h_out + PRHF_SNELL_OUTPUTS * (size_t)n_rays + path_elems, path_elems * 8);
        }
        return rc;
    }
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(out, a.out, PRHF_SNELL_OUTPUTS * (size_t)n_rays * 8, hipMemcpyDeviceToHost, c->stream));
        if (path_x) {
            HIP_TRY(hipMemcpyAsync(path_x, a.path_x, path_elems * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(path_z, a.path_z, path_elems * 8, hipMemcpyDeviceToHost, c->stream));
        }
    }
    return prhf_sync(c);
}
}  // namespace

int prhf_snell_cartesian_f64(prhf_ctx* c, const double* freq_hz, const double* elevation_deg,
                             const int64_t* profile_index, int64_t n_rays, const double* den, const double* bmag,
                             const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                             int64_t alt_stride_elems, int32_t mode, double* out, double* path_x, double* path_z,
                             int64_t path_stride, uint32_t flags) {
    const SnellGeometry geo{0, 6371.0, 1.0, 200.0, 400};
    return snell_run(c, geo, freq_hz, elevation_deg, profile_index, n_rays, den, bmag, bpsi, alt, n_prof, n_alt,
                     alt_stride_elems, mode, out, path_x, path_z, path_stride, flags);
}

int prhf_snell_spherical_f64(prhf_ctx* c, const double* freq_hz, const double* elevation_deg,
                             const int64_t* profile_index, int64_t n_rays, const double* den, const double* bmag,
                             const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                             int64_t alt_stride_elems, int32_t mode, double earth_radius_km, double dz_target_km,
                             double apex_boost, int32_t max_substeps, double* out, double* path_x, double* path_z,
                             int64_t path_stride, uint32_t flags) {
    if (!(earth_radius_km > 0.0) || !(earth_radius_km < 1e300) || !(dz_target_km > 0.0) || !(apex_boost >= 0.0) || max_substeps < 1)
        return fail(PRHF_EINVAL, "bad spherical tracer controls");
    const SnellGeometry geo{1, earth_radius_km, dz_target_km, apex_boost, max_substeps};
    return snell_run(c, geo, freq_hz, elevation_deg, profile_index, n_rays, den, bmag, bpsi, alt, n_prof, n_alt,
                     alt_stride_elems, mode, out, path_x, path_z, path_stride, flags);
}

int prhf_snell_fan_f64(prhf_ctx* c, int32_t geometry, const double* group_freq_hz, const int64_t* group_profile_index,
                       int64_t n_groups, const int64_t* ray_group, const double* elevation_deg, int64_t n_rays,
                       const double* den, const double* bmag, const double* bpsi, const double* alt, int64_t n_prof,
                       int64_t n_alt, int64_t alt_stride_elems, int32_t mode, double earth_radius_km,
                       double dz_target_km, double apex_boost, int32_t max_substeps, double* out, double* path_x,
                       double* path_z, int64_t path_stride, uint32_t flags) {
    if (geometry != 0 && geometry != 1) return fail(PRHF_EINVAL, "geometry is 0 (flat Earth) or 1 (spherical Earth)");
    if (!ray_group) return fail(PRHF_EINVAL, "null array pointer");
    if (geometry == 1 && (!(earth_radius_km > 0.0) || !(earth_radius_km < 1e300) || !(dz_target_km > 0.0) || !(apex_boost >= 0.0) ||
                          max_substeps < 1))
        return fail(PRHF_EINVAL, "bad spherical tracer controls");
    const SnellGeometry geo = geometry == 0 ? SnellGeometry{0, 6371.0, 1.0, 200.0, 400}
                                            : SnellGeometry{1, earth_radius_km, dz_target_km, apex_boost, max_substeps};
    return snell_run(c, geo, group_freq_hz, elevation_deg, group_profile_index, n_rays, den, bmag, bpsi, alt, n_prof, n_alt,
                     alt_stride_elems, mode, out, path_x, path_z, path_stride, flags, ray_group, n_groups);
}

int prhf_occupancy(prhf_ctx* c, int64_t n_alt, int32_t math, int32_t* workgroups_per_cu) {
    if (!c || !workgroups_per_cu) return fail(PRHF_EINVAL, "null pointer");
    if (n_alt < 1 || n_alt > kMaxAlt) return fail(PRHF_EINVAL, "n_alt out of range");
    ENTER_DEVICE(c->device);
    int n = 0;
    HIP_TRY(prhf::query_occupancy(math, prhf::lds_bytes_for(n_alt), &n));
    *workgroups_per_cu = n;
    return PRHF_OK;
}

int prhf_sync(prhf_ctx* c) {
    if (!c) return fail(PRHF_EINVAL, "null context");
    ENTER_DEVICE(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->status_pending) {
        c->status_pending = false;
        unsigned bits = 0;
        for (int b = 0; b < PRHF_STATUS_WORDS; ++b) {
            if (reinterpret_cast<volatile unsigned*>(c->h_status)[b]) bits |= 1u << b;
            c->h_status[b] = 0;
        }
        return status_to_code(bits);
    }
    return PRHF_OK;
}

int prhf_last_kernel_ms(prhf_ctx* c, double* ms) {
    if (!c || !ms) return fail(PRHF_EINVAL, "null pointer");
    if (!c->timed) return fail(PRHF_EINVAL, "no launch has been timed on this context");
    ENTER_DEVICE(c->device);
    HIP_TRY(hipEventSynchronize(c->end_ev()));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, c->ring0[c->slot], c->end_ev()));
    *ms = t;
    return PRHF_OK;
}

int prhf_recent_kernel_ms(prhf_ctx* c, double* ms, int32_t capacity, int32_t* n_out) {
    if (!c || !ms || !n_out || capacity < 0) return fail(PRHF_EINVAL, "null pointer or negative capacity");
    ENTER_DEVICE(c->device);
    long long n = (long long)std::min<unsigned long long>(c->n_timed, (unsigned long long)prhf_ctx::kTimingRing);
    if (n > capacity) n = capacity;
    *n_out = (int32_t)n;
    if (n == 0) return PRHF_OK;
    HIP_TRY(hipEventSynchronize(c->end_ev()));             // the newest; the older ones completed before it
    for (long long i = 0; i < n; ++i) {                      // oldest first
        const int s = (int)((c->n_timed - (unsigned long long)n + (unsigned long long)i) % prhf_ctx::kTimingRing);
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, c->ring0[s], c->ring1[s]));
        ms[i] = t;
    }
    return PRHF_OK;
}

}  // extern "C"

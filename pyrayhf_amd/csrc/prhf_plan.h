// prhf_plan.h - the host-side launch planner of libprhf.so: the context options (Knobs), the decomposition of a slice
// into wave-sized items and blocks (plan_slice) and the validation of a work list (validate_work_list).  No HIP call
// and no device state in here - plain arithmetic on the caller's descriptors - so that the same code is compiled
// into prhf_api.cpp and, under -fsanitize=address,undefined, into the host test tests/devtools/sanitize_host.cpp
// (tests/test_sanitizers_host.py).  Everything lives in an anonymous namespace: internal to the including file.

#ifndef PRHF_PLAN_H
#define PRHF_PLAN_H

#include <algorithm>
#include <cstdint>
#include <cstdio>

#include "prhf.h"
#include "prhf_kernels.h"

namespace {

// Launch-shaping and arithmetic settings of one context (prhf_ctx_set_option).  The defaults are
// the measured best; tests and A/B runs change them per context.  Only a -DPRHF_DIAG build reads them from the
// environment as well (PRHF_<NAME>, at context creation).
struct Knobs {
    double target_waves = 4096;        // waves resident at two 8-wave workgroups per CU: few-pair launches are chunked up to this
    double lean_min_points = 65;       // shorter grids skip the pair table and the main loop
    double well_conditioned = 1e-5;    // default O-mode arithmetic: the reference's operation order where 1 - X <= this
    double thread_scan_min = 0.0;      // n_freq x n_points from which X mode settles reflection heights per thread
    double no_candidates = 0;          // 1: no per-profile candidate list, every frequency is a work item
    double persistent = 1;             // 0: one workgroup per block, hardware dispatch order
    double tail_rounds = 1.0;          // resident rounds of workgroups at the end of a long slice that are cut finer
    double tail_bpp = 4;               // ... into this many workgroups per profile (1: no tail refinement)
    double split_min_points = 1024;    // few-profile slices are cut into several workgroups per profile from this grid size
    double split_few_profiles = 1;     // 0: one workgroup per profile whatever their number
    double short_kernel = 1;           // 0: short O-mode grids stay in the general kernel
    double shortx_kernel = 1;          // 0: X-mode grids of up to 4096 points stay in the general kernel
    double short_concurrent = 1;       // 0: short-grid and general launch of a mixed list one after the other
    double short_queue = 0;            // > 0: the short-grid kernel's queue holds exactly this many entries (tests)
    double direct_upload = 1;          // small host-buffer calls on a large-BAR device: the CPU writes the inputs straight into
                                       // device memory (0: pinned staging buffer + hipMemcpyAsync)
    double timing = 0;                 // 1: synchronous host-buffer operator calls record timing events too (device-pointer
                                       // launches always do).  Off by default: the two event records cost 3.5 us of a
                                       // 41 us single-profile call, and nothing is left on the stream when such a call
                                       // returns; prhf_last_kernel_ms keeps reporting the last launch that was timed
    double trim_lds = 1;               // columns of more than 1400 levels: stage only up to the highest peak of the launch and
                                       // stay on the LDS kernels when that fits (0: always the global-memory slabs)
    double short_compact = 1;          // short O-mode grids: four 4-wave workgroups per CU whose staged arrays hold as many
                                       // levels as a quarter of the LDS allows; a profile whose peak lies higher goes to a
                                       // second launch with full-size arrays (0: two 8-wave workgroups per CU only)
    double short_prio = 4;             // short-grid O kernel, bits: wave priority of a block's items by age (1: the blocks of the last
                                       // three resident rounds rank below everything pulled before them - config 3 -2.7 %),
                                       // by cost (2: a profile with many reflecting frequencies outranks its neighbours), both (3);
                                       // by phase (4, the default since round 5: staging, lists, the queue and the final sums -
                                       // latency chains of few instructions - run above every main loop of the CU, whose other
                                       // workgroups' loops fill the issue slots: config 3 -1.5 %, the O/200 slice of config 5
                                       // -1.8 % against 1; with the age rule on top, 5: -1.9 % / +1.4 %)
    double short_order = 1;            // long short-grid O launches draw their blocks in descending order of a cost estimate
                                       // (a pre-pass over sixteen samples of every density column; 0: index order)
    double short_lanes = 0;            // short-grid O kernel: lanes per pair, 16 (four pairs per work item) or 8 (eight: half the
                                       // items and their set-up per profile, no half-empty last wave-iteration on 200 points;
                                       // but eight pairs' nodes per LDS read: more bank conflicts).  0: eight on grids of up
                                       // to 256 points, sixteen beyond (measured: -12 % at 50 points, -4 % at 200, -1 % at 256,
                                       // +1 % at 400, +22 % at 1000)
    double host_slabs = 3;             // large host-buffer batches are uploaded, evaluated and returned in this many slabs of
                                       // profiles (10 % / 30 % / 60 %) so that the transfers of one overlap the kernel of
                                       // another (1: one upload, one launch, one download)
    double local_chunks = 1;           // few-pair launches: a pair's chunks are waves of ONE workgroup, which adds them up
                                       // itself (0: chunks anywhere in the launch, sums through scratch + vfo_finalize_kernel)
    double tall_lean = 1;              // profiles staged in global memory (more than 1400 levels below the highest peak): the
                                       // main loop on the slab's nodes (0: the generic loop)
    double snell_table = 4;            // tracers: the frequency-independent parts of every level's mu, mu' (f_N^2, g_p |B|,
                                       // sin psi, cos psi) once per profile when the rays (groups) number at least this
                                       // many times the profiles - a ray stops at its turning point, after a third to a
                                       // half of the column, the table covers all of it - and the table stays under
                                       // 1 GiB (0: never; values do not depend on it)
};
struct KnobName {
    const char* name;
    double Knobs::*field;
    double lo, hi;
};
const KnobName kKnobNames[] = {
    {"target_waves", &Knobs::target_waves, 64, 1e9},
    {"lean_min_points", &Knobs::lean_min_points, 2, 1e9},
    {"well_conditioned", &Knobs::well_conditioned, 0, 1},
    {"thread_scan_min", &Knobs::thread_scan_min, 0, 1e300},
    {"no_candidates", &Knobs::no_candidates, 0, 1},
    {"persistent", &Knobs::persistent, 0, 1},
    {"tail_rounds", &Knobs::tail_rounds, 0, 1e6},
    {"tail_bpp", &Knobs::tail_bpp, 1, 64},
    {"split_min_points", &Knobs::split_min_points, 1, 1e9},
    {"split_few_profiles", &Knobs::split_few_profiles, 0, 1},
    {"short_kernel", &Knobs::short_kernel, 0, 1},
    {"shortx_kernel", &Knobs::shortx_kernel, 0, 1},
    {"short_concurrent", &Knobs::short_concurrent, 0, 1},
    {"short_queue", &Knobs::short_queue, 0, PRHF_SHORT_MAX_QUEUE},
    {"local_chunks", &Knobs::local_chunks, 0, 1},
    {"direct_upload", &Knobs::direct_upload, 0, 1},
    {"timing", &Knobs::timing, 0, 1},
    {"trim_lds", &Knobs::trim_lds, 0, 1},
    {"short_compact", &Knobs::short_compact, 0, 1},
    {"short_prio", &Knobs::short_prio, 0, 7},
    {"host_slabs", &Knobs::host_slabs, 1, 3},
    {"short_order", &Knobs::short_order, 0, 1},
    {"short_lanes", &Knobs::short_lanes, 0, 16},
    {"snell_table", &Knobs::snell_table, 0, 1e9},
    {"tall_lean", &Knobs::tall_lean, 0, 1},
};
constexpr int kWavesPerBlock = PRHF_BLOCK_THREADS / 64;

// Decompose one slice into wave-sized items and blocks ("Launch geometry").
inline void plan_slice(prhf::SegDev& s, long long n_freq, long long wg_slots, const Knobs& kn) {
    const long long kTargetWaves = (long long)kn.target_waves;
    const bool kSplitFewProfiles = kn.split_few_profiles != 0;
    const long long kSplitMinPoints = (long long)kn.split_min_points;
    const double kTailRounds = kn.tail_rounds;
    const int kTailBpp = (int)kn.tail_bpp;
    const long long P = s.prof_end - s.prof_begin;
    const long long pairs = P * n_freq;
    const long long N = s.n_points;
    long long chunks = 1, chunk_len = ((N + 63) / 64) * 64;
    if (pairs > 0 && pairs < kTargetWaves && N > 256) {
        long long want = std::min((kTargetWaves + pairs - 1) / pairs, (N + 255) / 256);
        chunk_len = (((N + want - 1) / want + 63) / 64) * 64;
        chunks = (N + chunk_len - 1) / chunk_len;
    }
    s.slots = 0;
    if (chunks > 1 && kn.local_chunks != 0) {
        // Block-local chunks: S = 2, 4 or 8 slots per pair (the power of two at or below what the waves target asks
        // for), the pair's <= S chunks on consecutive waves of one workgroup, 8 / S pairs per workgroup.  One profile x
        // 174 frequencies x 20000 points: 174 workgroups of 8 chunks instead of 501 workgroups + a second kernel.
        long long want = std::min<long long>(std::min((kTargetWaves + pairs - 1) / pairs, (N + 255) / 256), kWavesPerBlock);
        long long S = 1;
        while (S * 2 <= want) S *= 2;
        if (S > 1) {
            chunk_len = (((N + S - 1) / S + 63) / 64) * 64;
            chunks = (N + chunk_len - 1) / chunk_len;          // <= S
            s.slots = (int)S;
        }
    }
    s.chunks = (int)chunks;
    s.chunk_len = (int)chunk_len;
    if (s.slots > 0) {
        s.blocks_per_prof = (int)((n_freq * s.slots + kWavesPerBlock - 1) / kWavesPerBlock);
        s.tail_prof = P;
        s.tail_bpp = s.blocks_per_prof;
        return;
    }
    const long long items = n_freq * chunks;
    long long waves = std::max<long long>(1, std::min(items, (kTargetWaves + std::max<long long>(P, 1) - 1) /
                                                                 std::max<long long>(P, 1)));
    s.blocks_per_prof = (int)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    // A long slice with few profiles - fewer than four resident rounds of one-workgroup profiles - is a launch of one
    // or two rounds whose last one is mostly empty slots (625 profiles of 20000 points on 512 slots: 7.8 ms for
    // 4.4 ms of work, tools/slice_cost5.py).  Cut every profile of such a slice into several workgroups (up to 32), each
    // with its share of the frequencies (it stages the profile again: ~13 us against milliseconds of items).
    if (kSplitFewProfiles && chunks == 1 && N >= kSplitMinPoints && P > 0 && P * s.blocks_per_prof < 4 * wg_slots) {
        long long bpp = std::min<long long>(32, (4 * wg_slots + P - 1) / P);
        while (bpp > 1 && items < bpp * kWavesPerBlock * 2) --bpp;     // at least two items per wave
        if (bpp > s.blocks_per_prof) s.blocks_per_prof = (int)bpp;
    }
    // A long slice of one-workgroup profiles ends on whole workgroups (milliseconds each at n_points = 20000)
    // while most of the chip has already drained.  Cut the profiles of the last kTailRounds rounds of
    // workgroup slots into kTailBpp workgroups each: the launch then drains in a fraction of a workgroup time.
    s.tail_prof = P;
    s.tail_bpp = s.blocks_per_prof;
    const long long tail = (long long)(kTailRounds * (double)wg_slots);
    if (kTailBpp > 1 && s.blocks_per_prof == 1 && chunks == 1 && N >= 1024 && P >= 4 * tail && tail > 0 &&
        n_freq >= (long long)kTailBpp * kWavesPerBlock) {
        s.tail_prof = P - tail;
        s.tail_bpp = kTailBpp;
    }
}

// The stretched grid must not decrease (smooth_nonuniform_grid never does): the top-segment search of the main loop
// relies on it.  Returns the first index i with mult[i] < mult[i - 1] inside a slice's range, -1 when there is none
// (ranges that do not lie inside the array are validate_work_list's to report).
inline long long first_decreasing_grid_entry(const double* mult, int64_t mult_len, const prhf_segment* segs, int32_t n_segs) {
    for (int32_t g = 0; g < n_segs; ++g)
        if (segs[g].mult_offset >= 0 && segs[g].n_points >= 1 && segs[g].mult_offset + segs[g].n_points <= mult_len)
            for (int64_t i = segs[g].mult_offset + 1; i < segs[g].mult_offset + segs[g].n_points; ++i)
                if (mult[i] < mult[i - 1]) return (long long)i;
    return -1;
}

// The caller's work list against the shapes of its arrays (include/prhf.h, prhf_vfo_worklist_f64): profile ranges inside
// [0, n_prof], a mode, a grid of at least one point that lies inside the multiplier array, output rows that start on
// a row boundary and that no two slices share.  PRHF_OK, or PRHF_EINVAL with the reason in `msg`.
inline int validate_work_list(const prhf_segment* segs, int32_t n_segs, int64_t n_prof, int64_t n_freq, int64_t mult_len,
                              char* msg, size_t msg_len) {
    for (int i = 0; i < n_segs; ++i) {
        const prhf_segment& u = segs[i];
        if (u.prof_begin < 0 || u.prof_end < u.prof_begin || u.prof_end > n_prof) {
            std::snprintf(msg, msg_len, "segment %d: profile range outside [0, n_prof]", i);
            return PRHF_EINVAL;
        }
        if (u.mode != PRHF_MODE_O && u.mode != PRHF_MODE_X) {
            std::snprintf(msg, msg_len, "mode must be 'O' or 'X'");
            return PRHF_EINVAL;
        }
        if (u.n_points < 1) {
            std::snprintf(msg, msg_len, "n_points must be >= 1");
            return PRHF_EINVAL;
        }
        if (u.mult_offset < 0 || u.mult_offset + u.n_points > mult_len) {
            std::snprintf(msg, msg_len, "segment %d: multiplier range outside the array", i);
            return PRHF_EINVAL;
        }
        if (u.out_offset < 0 || u.out_offset % n_freq != 0) {
            std::snprintf(msg, msg_len, "segment %d: output offset must be a non-negative multiple of n_freq", i);
            return PRHF_EINVAL;
        }
        for (int k = 0; k < i; ++k) {          // rows [out_offset / n_freq, + profiles) of two segments must not overlap
            const long long a0 = segs[k].out_offset / n_freq, a1 = a0 + (segs[k].prof_end - segs[k].prof_begin);
            const long long b0 = u.out_offset / n_freq, b1 = b0 + (u.prof_end - u.prof_begin);
            if (a0 < b1 && b0 < a1 && a0 < a1 && b0 < b1) {
                std::snprintf(msg, msg_len, "segments %d and %d write the same output rows", k, i);
                return PRHF_EINVAL;
            }
        }
    }
    return PRHF_OK;
}

}  // namespace

#endif  // PRHF_PLAN_H

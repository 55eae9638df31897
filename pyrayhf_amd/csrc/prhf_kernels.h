// prhf_kernels.h - launch interface between the C ABI (prhf_api.cpp) and the HIP kernels.
#ifndef PRHF_KERNELS_H
#define PRHF_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#define PRHF_KMODE_O 0
#define PRHF_KMODE_X 1

#define PRHF_STATUS_NEGDEN 0x1
#define PRHF_STATUS_PEAK0  0x2
#define PRHF_STATUS_BADINDEX 0x4    // a ray's profile_index outside [0, n_prof) (tracers)
#define PRHF_STATUS_WORDS 8         // one word of host-visible memory per status bit (post_status)
#define PRHF_STATUS_NANINPUT 0x10   // NaN in a profile's altitude column, or in |B| / psi below its peak
#define PRHF_STATUS_BADGROUP 0x8    // a ray's ray_group outside [0, n_groups) (grouped tracer launch)

#ifndef PRHF_BLOCK_THREADS
#define PRHF_BLOCK_THREADS 512      // 8 wavefronts share one staged profile
#endif
#ifndef PRHF_SHORT_THREADS
#define PRHF_SHORT_THREADS 512      // workgroup of the short-grid kernels (vfo_short_kernel, vfo_shortx_kernel)
#endif
#ifndef PRHF_SHORT_WAVES_PER_SIMD
#define PRHF_SHORT_WAVES_PER_SIMD 4 // ... and the occupancy their register budget is set for
#endif
#define PRHF_HINT_BUCKETS 1024      // uint16 segment hints, 2 KiB of LDS (non-uniform altitude grids)
#define PRHF_MAX_CAND 1024          // uint16 list of the frequencies that may reflect, 2 KiB of LDS
#define PRHF_MAX_SEGMENTS 8
#define PRHF_RED_DOUBLES 160        // block-reduction scratch (9 rows x up to 16 waves) + per-profile scalars
#define PRHF_NODE_BYTES 96          // one staged bottomside level
#define PRHF_SNODE_BYTES 80         // ... of the short-grid kernel (20 dwords: conflict-free under ds_read_b128)
#define PRHF_SHORT_MIN_POINTS 2     // grids of 2 .. 1024 points in the default O-mode arithmetic: vfo_short_kernel
#define PRHF_SHORT_MAX_POINTS 1024  // (16 wave-iterations: one violation mask each, per wave)
#define PRHF_SHORTX_MAX_POINTS 1024  // X mode (fast tier) up to this many points: vfo_shortx_kernel (no top-segment phase;
                                    // measured against the general kernel: -38 % at 200 points, -21 % at 500, -7 % at 1000, +6 % at 2000)
#define PRHF_SHORT_MASKS 32         // violation masks a wave keeps per work item: wave-iterations of the item that hold
                                    // ill-conditioned points (more: every point of the item's pairs in the reference's order)
#define PRHF_SHORT_MAX_QUEUE 4096   // entries of the LDS queue of ill-conditioned points, at most
#define PRHF_ORDER_CLASSES 16       // cost classes of the short-grid launch's block order (short_order_kernel)
#ifndef PRHF_COMPACT_THREADS
#define PRHF_COMPACT_THREADS 256    // the compact geometry of the short-grid O kernel: four 4-wave workgroups per CU, staged
#endif
#ifndef PRHF_COMPACT_WGS_PER_CU
#define PRHF_COMPACT_WGS_PER_CU 4   // arrays for as many levels as a quarter of the LDS holds
#endif
#ifndef PRHF_COMPACT_MIN_QUEUE
#define PRHF_COMPACT_MIN_QUEUE 256  // queue entries such a workgroup has at least (a profile's unused nodes come on top)
#endif
#define PRHF_PAIR_PAD 256           // entries behind the pair table that the main loop's prefetch may touch
#ifndef PRHF_TOP_MIN_POINTS
#define PRHF_TOP_MIN_POINTS 1024    // grids from this many points on give their top segment a loop of its own
#endif
#ifndef PRHF_TOP3_MIN_POINTS
#define PRHF_TOP3_MIN_POINTS 8192   // ... and from this many on, the two segments below it as well
#endif
#ifndef PRHF_MIN_WAVES_PER_SIMD
#define PRHF_MIN_WAVES_PER_SIMD 4   // two 8-wave workgroups per CU: caps VGPRs at 128
#endif

namespace prhf {

// waves per SIMD the short-grid kernels' register budget is set for: the compact geometry keeps PRHF_COMPACT_WGS_PER_CU
// workgroups of `threads` per CU resident (256 threads: 4 x 4 waves = 4 per SIMD; 320: 5), the full-size one two of 512
constexpr int short_waves_per_simd(int threads) {
    return threads < PRHF_SHORT_THREADS && threads * PRHF_COMPACT_WGS_PER_CU / 256 > PRHF_SHORT_WAVES_PER_SIMD
               ? threads * PRHF_COMPACT_WGS_PER_CU / 256 : PRHF_SHORT_WAVES_PER_SIMD;
}

// One homogeneous slice of a launch, with its decomposition into blocks.
struct SegDev {
    long long prof_begin, prof_end;  // profile rows [begin, end)
    long long mult_off;              // first element of this slice's multiplier grid
    long long out_off;               // first element of this slice's output rows
    long long block_begin;           // first block index of this slice
    long long partial_off;           // chunk-sum scratch offset (chunks > 1)
    long long altmin_off;            // per-profile min(alt) scratch offset (chunks > 1)
    int mode, n_points;
    int chunks;                      // wave-sized work items per pair
    int chunk_len;                   // grid points per chunk (multiple of 64)
    int slots;                       // > 0: block-local chunks - a pair's chunks (<= slots, a power of two <= 8) are the
                                     // consecutive waves of ONE workgroup, which adds them up itself (run_items)
    int blocks_per_prof;
    int tier;                        // 0 faithful, 1 fast (read by the mixed-tier kernel)
    // The last profiles of a long slice may be cut into more, shorter workgroups so that the launch
    // drains quickly: profiles [tail_prof, P) of the slice use tail_bpp blocks each (tail_prof = P: none).
    long long tail_prof;
    int tail_bpp;
    // Faithful tier only: where 1 - X exceeds this at all 64 points of a wave-iteration the reduced
    // algebra is used instead of the reference's operation order (+inf: never).
    double well_conditioned;
    int lean;                        // this slice uses the main loop (and KArgs::pairs); decided per slice so that a
                                     // slice gets the same arithmetic alone and inside a mixed launch
    int prio;                        // wave priority (0..3) of this slice's workgroups in a mixed launch: see vfo_kernel
    int thread_scan;                 // X mode: reflection heights settled one frequency per thread while the candidate
                                     // list is made (on by default; PRHF_THREAD_SCAN_MIN turns it off for A/B runs.
                                     // O mode always does, by binary search)
};

struct KArgs {
    const double* freq;
    const double* den;
    const double* bmag;
    const double* bpsi;
    const double* alt;
    const double* mult;
    const double* pairs;             // (m_i, m_i+1 - m_i) interleaved, same indexing as mult; may be null
    const double* ftab;              // (n_freq, 8) per-frequency scalars (freq_table_kernel); null: computed per pair
    double* out;
    double* partial;
    double* altmin;
    unsigned* status;
    unsigned* queue;                 // persistent launch: next block index - gridDim.x, zero at launch; null = one block per workgroup
    long long n_blocks;              // blocks of work in this launch
    unsigned long long* trace;       // -DPRHF_TRACE builds: (start, end) wall clock of every wave, else unused
    long long n_freq, n_alt, prof_stride, alt_stride;
    long long field_stride;          // row stride of bmag and bpsi: prof_stride, or 0 when one row serves every profile
    int n_segs;
    int no_candidates;               // PRHF_NO_CANDIDATES=1 (A/B runs): every frequency is a work item, none is pre-filtered
    // Short-grid launches (vfo_short_kernel) and their follow-up:
    unsigned* leftover;              // [0] count, [1..] block indices of the profiles the short-grid kernel does not take
    unsigned* leftover_tall;         // compact short-grid launch: ... of the profiles whose bottomside its staged arrays do not
                                     // hold (they go to the short-grid launch with full-size arrays); null: `leftover`
    const unsigned* block_list;      // follow-up launches: evaluate blocks block_list[1 .. block_list[0]] instead of 0 .. n_blocks
    int short_queue;                 // entries of the short-grid kernel's LDS queue
    int short_prio;                  // wave priorities of the short-grid O kernel's blocks (vfo_short_kernel)
    const unsigned* order;           // short-grid O launch: its blocks by cost class (short_order_kernel), or null: index order
    unsigned* zero_after;            // general kernel, follow-up of such a launch: the class counters to leave at zero (or null)
    // Profiles taller than LDS holds (vfo_tall_kernel): one slab of tall_stride bytes per workgroup of the launch
    unsigned char* tall;
    unsigned long long tall_stride;
    // Levels the staged arrays (nodes, f_N^2, g_p |B|) have room for: n_alt, or - a column of more than 1400 levels
    // whose bottomsides all fit LDS (launch_peak_levels) - the highest peak index of the launch + 1
    long long lds_levels;
    SegDev seg[PRHF_MAX_SEGMENTS];
};

// n_alt + 1 nodes (the last one may be the +inf sentinel), f_N^2 and g_p*B per level,
// segment hints, candidate frequencies, block-reduction scratch
inline size_t lds_bytes_for(long long n_alt) {
    return (size_t)(n_alt + 1) * PRHF_NODE_BYTES + (size_t)n_alt * 16 + PRHF_HINT_BUCKETS * 2 + PRHF_MAX_CAND * 2 +
           PRHF_RED_DOUBLES * 8;
}

// ... of a tall launch (vfo_tall_kernel): hints, candidate list (unused), reduction scratch; the staged profile
// itself - the first two terms of lds_bytes_for - is a slab of global memory per workgroup
inline size_t lds_bytes_tall() { return PRHF_HINT_BUCKETS * 2 + PRHF_MAX_CAND * 2 + PRHF_RED_DOUBLES * 8; }
inline size_t tall_slab_bytes(long long n_alt) {
    return (((size_t)(n_alt + 1) * PRHF_NODE_BYTES + (size_t)n_alt * 16) + 255) & ~(size_t)255;
}

// LDS of one short-grid workgroup (vfo_short_kernel): the per-frequency lists and scratch in front, then n_alt + 1
// nodes, then `queue` entries of 8 bytes (a profile with K < n_alt levels adds its unused nodes to the queue).
inline __host__ __device__ size_t short_lds_lists(long long n_alt, long long n_freq, int threads) {
    const size_t b = (size_t)(n_alt > n_freq ? n_alt : n_freq) * 8 + (size_t)n_freq * 24 + (size_t)(threads / 64) * PRHF_SHORT_MASKS * 12 +
                     PRHF_RED_DOUBLES * 8 + (size_t)n_freq * 4 + (size_t)n_freq * 2;
    return (b + 15) & ~(size_t)15;
}
inline size_t short_lds_fixed(long long n_alt, long long n_freq, int threads) {
    return short_lds_lists(n_alt, n_freq, threads) + (size_t)(n_alt + 1) * PRHF_SNODE_BYTES;
}
// ... of the X-mode variant (vfo_shortx_kernel): no queue; f_N^2 and g_p |B| per level, three doubles and an index
// per frequency, scratch, nodes
inline __host__ __device__ size_t shortx_lds_lists(long long n_alt, long long n_freq) {
    const size_t b = (size_t)n_alt * 16 + (size_t)n_freq * 24 + PRHF_RED_DOUBLES * 8 + (size_t)n_freq * 2;
    return (b + 15) & ~(size_t)15;
}
inline size_t shortx_lds_bytes(long long n_alt, long long n_freq) {
    return shortx_lds_lists(n_alt, n_freq) + (size_t)(n_alt + 1) * PRHF_SNODE_BYTES;
}
// queue entries that fit `budget` bytes beside a full node table (0: the kernel cannot run)
inline int short_queue_entries(long long n_alt, long long n_freq, size_t budget, int threads) {
    const size_t fixed = short_lds_fixed(n_alt, n_freq, threads);
    if (fixed + 8 * 64 > budget) return 0;
    const size_t q = (budget - fixed) / 8;
    return (int)(q > PRHF_SHORT_MAX_QUEUE ? PRHF_SHORT_MAX_QUEUE : q);
}

hipError_t configure_kernels(size_t max_lds_bytes);
hipError_t query_occupancy(int tier, size_t lds_bytes, int* blocks_per_cu);
// pairs[2 i], pairs[2 i + 1] = mult[i], mult[i + 1] - mult[i]; `pairs` holds n + PRHF_PAIR_PAD entries
hipError_t launch_grid_pairs(const double* mult, long long n, double* pairs, hipStream_t stream);
// tab[8 f ..] = f_hz, f2, cp^2/f2, (g_p/f)^2, 1/f2, 1/f_hz, 0, 0; row n_freq: min |freq_mhz| (tab holds n_freq + 1 rows)
// Control words that the per-frequency table's kernel zeroes on the way (its workgroup 0): the block queues and list
// heads of the launches that follow it on the stream.
struct ZeroWords {
    unsigned* p[8];
    int words[8];
    int n;
};
hipError_t launch_freq_table(const double* freq_mhz, long long n_freq, double* tab, const ZeroWords& zero, hipStream_t stream);
// tier: 0 faithful, 1 fast, 2 per slice (SegDev::tier)
// a.n_blocks blocks of work; with a.queue set the grid is `grid_blocks` persistent workgroups that pull
// block indices from the queue, else grid_blocks must equal a.n_blocks
hipError_t launch_vfo(const KArgs& a, long long grid_blocks, int tier, size_t lds_bytes, hipStream_t stream);
// profiles of more than 1400 levels: a.tall holds grid_blocks slabs of a.tall_stride >= tall_slab_bytes(n_alt) bytes
hipError_t launch_vfo_tall(const KArgs& a, long long grid_blocks, hipStream_t stream);
// *max_peak (one device word, zeroed by the caller) = the highest density-peak index - np.argmax, the first NaN
// ranking highest, exactly stage_profile's rule - over n_prof profiles: a column of more levels than LDS holds can
// still take the LDS kernels when every bottomside fits
hipError_t launch_peak_levels(const double* den, long long n_prof, long long n_alt, long long prof_stride,
                              unsigned* max_peak, hipStream_t stream);
// the short-grid kernel over a.n_blocks one-profile blocks (a.queue set: `grid_blocks` persistent workgroups);
// lds_bytes = short_lds_fixed + 8 a.short_queue; threads: PRHF_SHORT_THREADS or PRHF_COMPACT_THREADS
hipError_t launch_vfo_short(const KArgs& a, long long grid_blocks, size_t lds_bytes, int threads, int lanes, hipStream_t stream);
// the blocks of the short-grid launch `a` sorted into cost classes: order[0 .. CLASSES) counts (zero when the launch
// starts), then CLASSES lists of a.n_blocks entries; the same launch makes the per-frequency table `tab` of
// launch_freq_table and zeroes `zero`'s control words
hipError_t launch_short_order(const KArgs& a, unsigned* order, const double* freq_mhz, double* tab, const ZeroWords& zero,
                              hipStream_t stream);
// the X-mode variant; lds_bytes = shortx_lds_bytes(a.lds_levels, n_freq); threads: PRHF_SHORT_THREADS or PRHF_COMPACT_THREADS
hipError_t launch_vfo_shortx(const KArgs& a, long long grid_blocks, size_t lds_bytes, int threads, hipStream_t stream);
// absmax_scratch: 2 x u64 device words, absmax_host: 2 x u64 pinned host words
hipError_t launch_mu_mup(const double* X, const double* Y, const double* psi, long long n, int mode, int tier,
                         unsigned long long* absmax_scratch, unsigned long long* absmax_host,
                         double* mu, double* mup, hipStream_t stream);
hipError_t launch_find_vh(const double* X, const double* Y, const double* psi, const double* dh, long long n_rows,
                          long long n_cols, double alt_min, int mode, int tier,
                          unsigned long long* absmax_scratch, unsigned long long* absmax_host, double* vh,
                          hipStream_t stream);

// Standalone regrid of one profile (library.py:324-438); all pointers are device memory.
struct RegridArgs {
    const double* freq_hz;
    const double* den;
    const double* bmag;
    const double* bpsi;
    const double* alt;
    const double* mult;
    double* out_freq;
    double* out_den;
    double* out_bmag;
    double* out_bpsi;
    double* out_dist;
    double* out_alt;
    double* out_crit;
    long long* out_ind;
    unsigned* status;
    long long n_freq, n_alt;
    int n_points, mode;
};
hipError_t launch_regrid(const RegridArgs& a, size_t lds_bytes, hipStream_t stream);
// Stratified Snell's-law tracer (library.py:1096-1268), one wavefront per ray; device pointers.
#define PRHF_SNELL_OUTPUTS 8   // path km, delay s, x_mid, z_mid, ground range, x_turn, z_turn, n_path
struct SnellArgs {
    const double* den;
    const double* bmag;
    const double* bpsi;
    const double* alt;
    const double* freq_hz;       // (n_rays)
    const double* elev_deg;      // (n_rays)
    const long long* prof_idx;   // (n_rays) or null: every ray uses profile 0
    // Grouped launch (prhf_snell_fan_f64): rays that share (profile, frequency) read their levels from a table
    // computed once per group instead of evaluating the Appleton-Hartree index level by level themselves.
    const long long* ray_group;  // (n_rays) group of each ray, or null: ungrouped (freq_hz / prof_idx per ray)
    const double* group_freq;    // (n_groups) [Hz]
    const long long* group_prof; // (n_groups) or null: profile 0
    long long n_groups;
    double* levels;              // (n_groups, n_alt + 1) mu' of every level (ground level first when inserted)
    // ... and, made beside it once per group (snell_tables_kernel), what every ray of the group would otherwise make itself:
    double* group_entries;       // (n_groups, n_alt + 1, 4) the levels with a finite mu, compacted: altitude, mu, mu', running
                                 // minimum of the bracket criterion (mu, or mu r on a spherical Earth) over the entries 1 .. i
    int* group_kidx;             // (n_groups, n_alt + 1) grid level of each entry
    int* group_info;             // (n_groups, 4) entries, "ground level inserted", error bits (1 profile index, 2 negative
                                 // density), profile
    // Per-ray launch: persistent wavefronts drawing rays from queues (snell_queue_bytes() of counters, zeroed by the launch's first kernel)
    unsigned* ray_queue;
    int resident_cus;            // multiprocessors of the device (sizes the persistent grid)
    double* prof_info;           // (n_prof, 4) scratch: max|B|, "has a negative density", the level of the largest density (a
                                 // hint: where a ray that escapes would have come closest to turning), filled by launch_snell
    // Per-profile level table (or null), filled by launch_snell: what a level's mu and mu' need that does not depend on
    // the frequency - f_N^2 = (sqrt(den) c_p)^2, g_p |B|, sin(psi), cos(psi) - so that a ray pays two quotients and
    // the Appleton-Hartree algebra per level instead of a square root and a correctly rounded sine / cosine on top
    // (the same operations on the same values: bit-identical).  Worth it when the rays outnumber the profiles.
    double* ptab;                // (n_prof, n_alt, 4)
    long long n_prof;
    double* out;                 // (n_rays, PRHF_SNELL_OUTPUTS)
    double* path_x;              // (n_rays, path_stride) or null
    double* path_z;
    unsigned* status;
    long long n_rays, n_alt, prof_stride, alt_stride, path_stride;
    int mode;
    int reduced;                 // per-ray launch: levels far from reflection and from the ray's turning point in the reduced
                                 // algebra (the default; 0: every level in the reference's operation order) - prhf_snell.inc
    int geometry;                // 0 flat Earth, 1 spherical Earth
    double earth_radius_km;      // spherical only (library.py:1473, :1552-1553)
    double dz_target_km;         // spherical sub-step controls (library.py:1470-1472)
    double apex_boost;
    int max_substeps;
};
size_t snell_queue_bytes();            // the per-ray launch's queue counters (SnellArgs::ray_queue, 128-byte aligned)
hipError_t launch_snell(const SnellArgs& a, hipStream_t stream);   // with a.ray_group: level table kernel first
// wavefronts of the per-ray kernel that one device keeps resident; cu_count: multiprocessors of the device
hipError_t snell_resident_waves(long long n_alt, int cu_count, long long* waves, bool ptab = false, bool reduced = false,
                                int geometry = 0);

// residual / cost may be null
hipError_t launch_residual(const double* vh_model, const double* vh_obs, long long n_prof, int n_freq,
                           double* residual, double* cost, hipStream_t stream);

}  // namespace prhf

#endif

/*
 * prhf.h - C ABI of libprhf.so, the MI355X (gfx950) vertical-ionogram forward operator.
 *
 * Drop-in boundary.  The reference has no FFI layer: its boundary is the Python function
 *     PyRayHF.library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode='O', n_points=200)
 * (reference PyRayHF/library.py:459-509).  The entry points below are what a ctypes binding
 * for that function calls; pyrayhf_amd/library.py is that binding.
 *
 * Conventions
 *   - plain pointers and sizes only; all arrays are float64, C-contiguous rows;
 *   - every function returns 0 (PRHF_OK) or a negative PRHF_E* code; prhf_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - the library borrows caller buffers for the duration of a call (or, with
 *     PRHF_FLAG_ASYNC, until the next prhf_sync on that context) and owns nothing but its
 *     context (stream, scratch, events);
 *   - a context is bound to one device and is not thread-safe; different contexts are
 *     independent and may be used from different threads;
 *   - every entry point runs on its context's device and restores the calling thread's current
 *     HIP device before it returns (a process that drives several GPUs keeps its own current device).
 *
 * Units are the reference's (library.py:465-474): freq MHz, den m^-3, bmag Tesla,
 * bpsi degrees, alt km; the result is virtual height in km, NaN where the sounder
 * frequency is not reflected below the density peak.
 */
#ifndef PRHF_H
#define PRHF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRHF_ABI_VERSION 3   /* 2: + prhf_snell_fan_f64, prhf_recent_kernel_ms, PRHF_FLAG_SHARED_FIELD (round 2)
                              * 3: + prhf_ctx_set_option (round 3) */

/* return codes */
#define PRHF_OK        0
#define PRHF_EINVAL   -1   /* null pointer, bad shape, n_points < 1, bad mode, bad flag combination, bad segment or index,
                            * a host-buffer multiplier grid that decreases, an unknown option */
#define PRHF_ENEGDEN  -2   /* a density below the peak is negative (reference library.py:93-94 raises ValueError) */
#define PRHF_EPEAK0   -3   /* density peak at index 0: empty bottomside (the reference raises IndexError) */
#define PRHF_EHIP     -4   /* HIP runtime failure; message carries hipGetErrorString */
#define PRHF_ENOMEM   -5   /* scratch allocation failed */

/* wave modes: reference 'O' and 'X' (library.py:391-396, :221-226) */
#define PRHF_MODE_O 0
#define PRHF_MODE_X 1

/* flags of prhf_vfo_batch_f64 / prhf_vfo_worklist_f64 */
#define PRHF_FLAG_DEVICE_PTRS 0x1u  /* every array pointer (inputs, multiplier, output) is device memory */
#define PRHF_FLAG_ASYNC       0x2u  /* device pointers only: enqueue and return; data errors surface at prhf_sync */
#define PRHF_FLAG_GRID_STABLE 0x4u  /* the multiplier array at this address (device or host) keeps its contents for as
                                     * long as the context lives, so the table the library derives from it (grid steps,
                                     * one small kernel) - and, for a host array, its device copy - is made once per
                                     * (address, length) and reused (at most 16 host grids are remembered) */
#define PRHF_FLAG_SHARED_FIELD 0x8u /* bmag and bpsi are ONE row of n_alt values each, shared by every profile (a fit's
                                     * candidates differ in their density only, library.py:589-591): a third of the
                                     * bytes to upload and to read */

/* arithmetic tiers, prhf_ctx_set_math (see DESIGN.md "Arithmetic tiers") */
#define PRHF_MATH_FAITHFUL 0  /* reference operation order, IEEE divide/sqrt, no contraction */
#define PRHF_MATH_FAST     1  /* reduced algebra, rsqrt + Newton, contracted; X-mode error <= 1e-9 relative */
#define PRHF_MATH_AUTO     2  /* the default, per slice.  X mode: fast.  O mode (ill conditioned near X = 1): the
                               * reference's operation order at every grid point with 1 - X <= 1e-5, the reduced
                               * algebra where the order cannot matter; reproduces the reference to 1e-10 */

typedef struct prhf_ctx prhf_ctx;

/* One homogeneous slice of a mixed launch (BASELINE config 5): profiles [prof_begin, prof_end)
 * are evaluated with one mode and one grid size.  mult_offset indexes the concatenated
 * multiplier array; the slice's output rows start at vh_out + out_offset (row-major
 * (prof_end - prof_begin, n_freq)).  out_offset must be a non-negative multiple of n_freq and the
 * output rows of two segments must not overlap (PRHF_EINVAL otherwise).  With host buffers, output rows
 * below the highest written row that no segment covers come back as NaN; with device pointers the
 * library writes only the rows its segments cover. */
typedef struct prhf_segment {
    int64_t prof_begin;
    int64_t prof_end;
    int32_t mode;
    int32_t n_points;
    int64_t mult_offset;
    int64_t out_offset;
} prhf_segment;

int prhf_abi_version(void);
const char* prhf_last_error(void);
int prhf_device_count(int* n);

int prhf_ctx_create(int device, prhf_ctx** out);
int prhf_ctx_destroy(prhf_ctx* ctx);

/* borrow != 0: launch on the caller's hipStream_t (e.g. torch's current stream; a NULL handle is the legacy
 * default stream, which is what torch reports for its default stream).  borrow == 0: return to the context's
 * own non-blocking stream (hip_stream is ignored). */
int prhf_ctx_set_stream(prhf_ctx* ctx, void* hip_stream, int32_t borrow);

/* Select the arithmetic tier (PRHF_MATH_*), default PRHF_MATH_AUTO. */
int prhf_ctx_set_math(prhf_ctx* ctx, int level);

/* Launch-shaping and arithmetic settings of this context, by name (tests and A/B measurements; the defaults are
 * the measured best and no caller needs to touch them).  The library reads no environment variable: only a
 * -DPRHF_DIAG build presets these from PRHF_<NAME> at context creation.  PRHF_EINVAL for an unknown name or a
 * value outside the option's range.
 *   "well_conditioned"   default O-mode arithmetic: the reference's operation order where 1 - X <= this (1e-5)
 *   "short_kernel"       0: short O-mode grids (2 .. 1024 points) stay in the general kernel (1)
 *   "shortx_kernel"      0: X-mode grids of 2 .. 1024 points (fast tier) stay in the general kernel (1)
 *   "short_concurrent"   0: a mixed list runs its short-grid and general launches one after the other (1)
 *   "short_queue"        > 0: the short-grid kernel's queue of ill-conditioned points holds exactly this many entries (0)
 *   "persistent", "tail_bpp", "tail_rounds", "split_few_profiles", "split_min_points", "target_waves"
 *                        how a launch is cut into workgroups (never the value of a pair)
 *   "no_candidates", "thread_scan_min", "lean_min_points"
 *                        which of the equivalent search / loop paths a pair takes
 *   "local_chunks"       0: the chunks of a few-pair launch on a long grid are spread over the launch and added by a
 *                        second kernel (1: a pair's chunks are waves of one workgroup, which adds them up itself)
 *   "direct_upload"      0: a small host-buffer call stages its inputs in pinned memory and copies them (1: on a
 *                        large-BAR device the CPU writes them straight into device memory)
 *   "timing"             1: synchronous host-buffer operator calls record timing events too (0; see prhf_last_kernel_ms)
 *   "trim_lds"           0: a column of more than 1400 levels is always staged in global memory (1: when every
 *                        bottomside of the launch fits LDS, only the levels up to the highest peak are staged)
 *   "tall_lean"          0: a profile staged in global memory takes the generic loop (1: the main loop reads its nodes
 *                        from the slab)
 *   "short_compact", "short_prio", "short_order", "short_lanes"
 *                        geometry of the short-grid kernels: four 4-wave workgroups per CU (1), wave priorities by age (1),
 *                        blocks in descending cost order (1), lanes per pair in the O kernel (8; 16: four pairs per work
 *                        item instead of eight - another order of additions in a pair's sum, 1e-16 apart)
 *   "host_slabs"         large host-buffer batches go in this many slabs so that transfers overlap the kernel (3; 1: none)
 *   "snell_table"        tracers: the frequency-independent parts of a level's mu, mu' (f_N^2, g_p |B|, sin psi, cos psi)
 *                        are tabulated once per profile when the rays (groups) number at least this many times the
 *                        profiles and the table stays under 1 GiB (4; 0: never) */
int prhf_ctx_set_option(prhf_ctx* ctx, const char* name, double value);

/*
 * Virtual heights of n_prof profiles x n_freq sounder frequencies.
 *
 * Replaces: vertical_forward_operator (reference library.py:459-509) and everything below it
 * (regrid_to_nonuniform_grid :324-438, find_X :120-137, find_Y :140-158, find_mu_mup :161-256,
 * find_vh :259-293), evaluated once per profile row; semantics are a loop of single-profile
 * reference calls.
 *
 *   freq_mhz    (n_freq)                       sounder frequencies, MHz
 *   den,bmag,bpsi  n_prof rows of n_alt values, row p at ptr + p*prof_stride_elems
 *   alt         (n_alt) shared by all profiles when alt_stride_elems == 0, else row p at
 *               alt + p*alt_stride_elems; ascending
 *   multiplier  (n_points) stretched unit grid, smooth_nonuniform_grid(0,1,n_points,10.)
 *               (library.py:296-321, :361-364) computed by the host in float64; values in [0, 1] and
 *               NON-DECREASING (the main loop's top-segment search relies on it: PRHF_EINVAL for a host
 *               buffer that decreases; a device-resident grid is the caller's responsibility)
 *   A frequency that is not a positive finite number gives a NaN column (the reference: NaN for 0 and NaN).
 *   vh_out      (n_prof, n_freq) row-major
 *   NaN inputs are not errors; they behave as in the reference: a NaN in den ranks as the column's maximum, the
 *   first one wins (np.argmax, library.py:371: a density column padded with NaN is cut at the padding); a NaN in alt
 *   makes the profile's row NaN (:507), and so does a NaN in bmag below the peak in X mode (:389); in O mode, and for
 *   a NaN in bpsi, the grid points of the two segments next to that level drop out of the sum (:288).
 *   (prhf_regrid_f64 refuses a NaN in alt, or in bmag / bpsi below the peak: PRHF_EINVAL.)
 * Limits: n_alt <= 65535, n_freq <= 2^20, n_points >= 1.  A profile's bottomside - the levels below its density
 * peak - is held in LDS when it has at most 1400 levels (for n_alt > 1400 a pre-pass finds the highest peak of the
 * launch: one synchronisation); taller bottomsides are staged in global memory (one slab per resident workgroup,
 * allocated by the context) and run the same main loop on the slab's nodes - 1.1 to 1.4 times the time per grid point
 * of a profile held in LDS (option "tall_lean" 0: the generic loop, about three times).
 */
int prhf_vfo_batch_f64(prhf_ctx* ctx,
                       const double* freq_mhz, int64_t n_freq,
                       const double* den, const double* bmag, const double* bpsi,
                       const double* alt,
                       int64_t n_prof, int64_t n_alt,
                       int64_t prof_stride_elems, int64_t alt_stride_elems,
                       const double* multiplier, int32_t n_points, int32_t mode,
                       double* vh_out, uint32_t flags);

/* Mixed O/X and mixed n_points in one launch.  multiplier is the concatenation of the
 * segments' grids; segs is host memory in every flag combination. */
int prhf_vfo_worklist_f64(prhf_ctx* ctx,
                          const double* freq_mhz, int64_t n_freq,
                          const double* den, const double* bmag, const double* bpsi,
                          const double* alt,
                          int64_t n_prof, int64_t n_alt,
                          int64_t prof_stride_elems, int64_t alt_stride_elems,
                          const double* multiplier, int64_t multiplier_len,
                          const prhf_segment* segs, int32_t n_segs,
                          double* vh_out, uint32_t flags);

/*
 * Appleton-Hartree phase index mu and group index mu' on flat arrays of n elements.
 * Replaces: find_mu_mup (reference library.py:161-256); psi_deg in degrees.  As in the reference,
 * the isotropic formulas are used when max|Y| over the whole array is below 1e-12 (:201-207).
 * Synchronous (one device->host word decides the branch).
 */
int prhf_mu_mup_f64(prhf_ctx* ctx, const double* X, const double* Y, const double* psi_deg, int64_t n,
                    int32_t mode, double* mu_out, double* mup_out, uint32_t flags);

/*
 * Virtual heights from already-regridded arrays.  Replaces: find_vh (reference library.py:259-293):
 * vh[r] = nansum_c(mu'(X,Y,psi)[r,c] * dh[r,c]), exact 0 -> NaN, + alt_min.  X, Y, psi_deg, dh are
 * (n_rows, n_cols) row-major; the isotropic test spans the whole array as in find_mu_mup.  Synchronous.
 */
int prhf_find_vh_f64(prhf_ctx* ctx, const double* X, const double* Y, const double* psi_deg, const double* dh,
                     int64_t n_rows, int64_t n_cols, double alt_min, int32_t mode, double* vh_out,
                     uint32_t flags);

/*
 * The un-fused first half of the operator for ONE profile.  Replaces: regrid_to_nonuniform_grid
 * (reference library.py:324-438): peak truncation, reflection heights, stretched altitudes and the
 * profile sampled on them.  freq_hz in Hz (as the reference's function takes it); every output is
 * (n_freq, n_points) row-major: the reference's dict entries 'freq', 'den', 'bmag', 'bpsi', 'dist',
 * 'alt', 'crit_height' (float64) and 'ind' (int64).  IEEE arithmetic in the reference's order: the
 * outputs are bit-identical to NumPy's.  Synchronous; returns PRHF_ENEGDEN / PRHF_EPEAK0 on bad input.
 */
int prhf_regrid_f64(prhf_ctx* ctx, const double* freq_hz, int64_t n_freq, const double* den, const double* bmag,
                    const double* bpsi, const double* alt, int64_t n_alt, const double* multiplier,
                    int32_t n_points, int32_t mode, double* out_freq, double* out_den, double* out_bmag,
                    double* out_bpsi, double* out_dist, double* out_alt, double* out_crit, int64_t* out_ind,
                    uint32_t flags);

/*
 * Residual rows of the fitting driver for n_prof candidate profiles against one observed trace.
 * Replaces: the arithmetic of residual_VH (reference library.py:660-669) applied to a batch: modeled
 * NaNs are replaced by max(nanmean|vh_model row|, 100), residual = vh_obs - vh_model, and
 * cost[p] = sum_f residual[p,f]^2 (the objective of the brute-force search, library.py:794-798).
 * vh_model (n_prof, n_freq) is normally the output of prhf_vfo_batch_f64 left on the device;
 * residual_out or cost_out may be NULL.
 */
int prhf_residual_f64(prhf_ctx* ctx, const double* vh_model, const double* vh_obs, int64_t n_prof, int64_t n_freq,
                      double* residual_out, double* cost_out, uint32_t flags);

/*
 * prhf_vfo_batch_f64 followed by prhf_residual_f64 in one call: the candidate profiles are staged once,
 * the modeled traces never leave HBM between the two kernels.  vh_out may be NULL (only residuals / costs
 * wanted); residual_out or cost_out may be NULL.  This is the whole inner loop of the reference's
 * brute-force fit (minimize_parameters, library.py:794-798: one residual_VH call per grid node) as one launch.
 */
int prhf_vfo_residual_f64(prhf_ctx* ctx, const double* freq_mhz, int64_t n_freq, const double* den,
                          const double* bmag, const double* bpsi, const double* alt, int64_t n_prof,
                          int64_t n_alt, int64_t prof_stride_elems, int64_t alt_stride_elems,
                          const double* multiplier, int32_t n_points, int32_t mode, const double* vh_obs,
                          double* vh_out, double* residual_out, double* cost_out, uint32_t flags);

/*
 * Stratified Snell's-law ray tracing over a flat Earth for n_rays rays (one wavefront each).
 * Replaces: trace_ray_cartesian_snells (reference library.py:1096-1268) with tan_from_mu_scalar
 * (:1034-1062) and find_turning_point (:1065-1093).  Ray r has frequency freq_hz[r] [Hz], launch elevation
 * elevation_deg[r] above the horizon and uses profile profile_index[r] (NULL: profile 0) of the
 * (n_prof, n_alt) columns; a ground level at z = 0 is inserted when alt[0] > 0, as in the reference.
 * out is (n_rays, 8): group_path_km, group_delay_sec, x_midpoint, z_midpoint, ground_range_km (the
 * reference's dict entries; its x_apex_km / z_apex_km equal the midpoint), then x and z of the turning
 * point and the number of path nodes.  The midpoint is the path node before the apex: the exact-arithmetic answer of
 * the reference's search (library.py:1248-1252), whose own rounding lands there for two rays in three and on the apex
 * for the third.  Arithmetic (prhf_ctx_set_math): PRHF_MATH_FAITHFUL evaluates mu and mu' of every level in the
 * reference's operation order (reference-run rays to 1e-12); the default evaluates levels far from reflection and from
 * the ray's turning point in the reduced algebra (within 1e-10 of the former; spherical rays at the edge of a skip zone, one
 * in a few million of a random set: up to 3e-10).  Rays that never turn give NaN (node count 0).  path_x / path_z
 * (optional, (n_rays, path_stride), path_stride >= 2 n_alt + 1) receive the reference's 'x' and 'z'
 * arrays padded with NaN.  Synchronous; PRHF_ENEGDEN on a negative density; PRHF_EINVAL when a
 * profile_index lies outside [0, n_prof) - checked on the host for host buffers and by the kernel for
 * device-resident ones (that ray's outputs are NaN, no memory outside the columns is read).
 */
int prhf_snell_cartesian_f64(prhf_ctx* ctx, const double* freq_hz, const double* elevation_deg,
                             const int64_t* profile_index, int64_t n_rays, const double* den, const double* bmag,
                             const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                             int64_t alt_stride_elems, int32_t mode, double* out, double* path_x, double* path_z,
                             int64_t path_stride, uint32_t flags);

/*
 * The same over a spherical Earth (Bouguer's law mu r sin(theta) = const, adaptive midpoint sub-steps towards
 * the apex).  Replaces: trace_ray_spherical_snells (reference library.py:1460-1713); the four controls are
 * its R_E (6371 km), dz_target_km (1.0), apex_boost (200.0) and max_substeps (400).  Outputs as for the
 * flat-Earth tracer, with x = R_E * phi.
 */
int prhf_snell_spherical_f64(prhf_ctx* ctx, const double* freq_hz, const double* elevation_deg,
                             const int64_t* profile_index, int64_t n_rays, const double* den, const double* bmag,
                             const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                             int64_t alt_stride_elems, int32_t mode, double earth_radius_km, double dz_target_km,
                             double apex_boost, int32_t max_substeps, double* out, double* path_x, double* path_z,
                             int64_t path_stride, uint32_t flags);

/*
 * Fans of rays: many elevations per (profile, frequency).  The tracers above evaluate the Appleton-Hartree index
 * level by level for every ray (library.py:1184-1189 / :1566-1571) although it depends on the profile and the
 * frequency only; here the n_groups (profile, frequency) groups get their tables once - mu and mu' of every level
 * (one thread per group and level, the reference's operation order) and the list of the levels with a finite mu,
 * compacted, with the running minimum of the turning-point criterion, so that a ray finds its bracket
 * (library.py:1085-1093 / :1598-1603: the first pair of consecutive finite levels the invariant falls between) with
 * two loads - and every ray reads the tables of its group: group_freq_hz[g] [Hz], group_profile_index[g] (NULL:
 * profile 0), ray_group[r] in [0, n_groups), elevation_deg[r].  geometry 0: flat Earth (the four controls are
 * ignored), 1: spherical Earth.  Outputs, paths, flags and errors as for the per-ray calls; the results are those of
 * the per-ray calls on the same rays in PRHF_MATH_FAITHFUL to 1e-15 (other orders of summation).  PRHF_EINVAL when the
 * tables (n_groups x (n_alt + 1) x 44 bytes in the context's scratch) would exceed 64 GiB, and - checked on the host
 * for host buffers, by the kernel for device-resident arrays (reported at the synchronisation) - when a ray_group or
 * a profile index is out of range.
 */
int prhf_snell_fan_f64(prhf_ctx* ctx, int32_t geometry, const double* group_freq_hz,
                       const int64_t* group_profile_index, int64_t n_groups, const int64_t* ray_group,
                       const double* elevation_deg, int64_t n_rays, const double* den, const double* bmag,
                       const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                       int64_t alt_stride_elems, int32_t mode, double earth_radius_km, double dz_target_km,
                       double apex_boost, int32_t max_substeps, double* out, double* path_x, double* path_z,
                       int64_t path_stride, uint32_t flags);

/* Diagnostics: workgroups of the fused kernel the runtime expects to keep resident per CU for
 * profiles of n_alt levels (LDS-limited) in arithmetic tier `math`. */
int prhf_occupancy(prhf_ctx* ctx, int64_t n_alt, int32_t math, int32_t* workgroups_per_cu);

/* Wait for everything enqueued on the context; returns PRHF_ENEGDEN / PRHF_EPEAK0 if a
 * kernel flagged bad input since the last sync. */
int prhf_sync(prhf_ctx* ctx);

/* Device time of the most recent TIMED launch (all kernels of that call), from HIP events on the
 * context's stream.  Synchronises on the stop event.  Every launch on device pointers is timed; a synchronous
 * host-buffer call of the operator (prhf_vfo_batch_f64 / _worklist / _residual without PRHF_FLAG_DEVICE_PTRS) is
 * timed only after prhf_ctx_set_option(ctx, "timing", 1): its two event records are 3.5 us of a 41 us call. */
int prhf_last_kernel_ms(prhf_ctx* ctx, double* ms);

/* The same for the most recent launches, oldest first: at most `capacity` of them (the context remembers 64);
 * *n = how many were written.  One synchronisation, on the newest launch - what a caller needs that enqueues a
 * series of launches and wants every one's device time without a host round trip between them (bench.py). */
int prhf_recent_kernel_ms(prhf_ctx* ctx, double* ms, int32_t capacity, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* PRHF_H */

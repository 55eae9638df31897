// prhf_kernels.hip - fused vertical-ionogram forward operator for gfx950 (MI355X, CDNA4).
//
// One wavefront (64 lanes) evaluates one (profile, frequency) pair - or one chunk of its stretched grid when
// few pairs are submitted.  (Short O-mode grids have a kernel of their own: prhf_short.inc.)  A workgroup shares one profile: its bottomside
// columns are staged once into LDS as 96-byte nodes (stage_profile), so that every grid point costs at most one
// LDS round trip and no HBM traffic.  Per pair:
//   S3-S6  reflection height (reflection_height): X mode - lanes stride over the levels, first level with
//          X + Y > 1 by ballot, running maximum below it by a wave max-reduce; O mode - a 64-ary search in the
//          profile's running maximum of f_N^2 (prefix_max_in_place); np.interp semantics either way;
//   S7-S10 the n_points stretched altitudes: segment (closed form on uniform grids, LDS hint table otherwise;
//          none at all in the top segment, whose node is held in registers), linear interpolation of
//          den/|B|/psi, Appleton-Hartree mu and mu' (lean_loop: the main loop; integrate_chunk: the generic loop);
//   S11    left-rectangle sum of mu'*dh with NaNs skipped, wave sum-reduce, 0 -> NaN, + min(alt).
// Frequencies that escape for certain never become work items (list_candidates).
// Stage names S0-S11 are SURVEY.md section 2.1; reference line numbers are PyRayHF/library.py.
//
// Two arithmetic tiers (DESIGN.md "Arithmetic tiers"):
//   TIER 0: the reference's operation order, IEEE divide and sqrt, no contraction, sin/cos/pow rounded like a
//           correctly rounding libm (prhf_crmath.h) - everywhere, or (default for O mode) where 1 - X <= 1e-5;
//   TIER 1: algebraically reduced mu' (group_index_lean: 30 operations from (den, Y^2, sin^2 psi), 2 v_rsq_f64),
//           sin^2(psi) by a per-segment polynomial, FMA contraction - the default for X mode.
//
// No MFMA: the work is elementwise float64 transcendental + reduction ("Roofline").

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prhf_crmath.h"
#include "prhf_kernels.h"

namespace prhf {

namespace {

constexpr double kPlasma = 8.97866275;              // library.py:61
constexpr double kGyro = 2.799249247e10;            // library.py:64
constexpr double kBackoff = 1e-6;                   // library.py:378
constexpr double kDegToRad = 0.017453292519943295;  // numpy deg2rad multiplies by pi/180
constexpr double kUnmagTol = 1e-12;                 // library.py:163
constexpr double kLightKmS = 299792.458;            // library.py:70
constexpr double kPolyAngle = 3e-4;                 // rad per segment below which the sin^2 cubic errs < 3e-15
constexpr double kQuadAngle = 4e-5;                 // ... and below which its economised quadratic errs < 1.4e-15
constexpr double kTrigAngle = 0.05;                 // rad per segment up to which the angle is rotated from the node's
                                                    // cos 2psi / sin 2psi by a short Taylor pair (|2 theta| <= 0.1: 3e-17)
constexpr double kLinTol = 1e-10;                   // a segment's quadratic term may be economised away when that costs
                                                    // < 1e-10 in sin^2 psi: an offset of 4e-11 everywhere moves a virtual
                                                    // height by <= 1.6e-11 (measured), an equioscillating one by less

// One bottomside level = the left end of one np.interp segment [alt_j, alt_j+1): den, b are the
// level values, sden, sb the np.interp slopes, and the abscissa is dz = z - alt_j.
//   reference-order arithmetic: psi [deg], spsi = d(psi)/dz [deg/km]; dz is computed as the reference does.
//   reduced arithmetic:         dz = m*span + off in one FMA, off = alt_0 - alt_j (z = m*span + alt_0, :413).
//                  Segment turning psi by < kPolyAngle: 2 cos^2(psi) = u0 + dz*(u1 + dz*(u2 + dz*u3));
//                  other segments: u0 = psi_j [rad], u1 = d(psi)/dz [rad/km], u3 = NaN (the flag).
// Both sets are staged in both tiers: the faithful tier's default mode uses the reduced arithmetic wherever
// 1 - X is not small.
// Everything is anchored at the LEFT level on purpose: below a steep layer den_j can be 0 (or
// 1e-15 of den_j+1) and den_j + sden*dz keeps its relative accuracy there, which an expansion
// about the segment centre would not (a density of -1e-6 m^-3 is enough to flip mu > 1, :238).
// off sits next to alt so that the main loop reads off..u3 as 8 + 4 x 16 bytes.
struct __attribute__((aligned(16))) Node {
    double alt, off, den, sden, b, sb, u0, u1, u2, u3, psi, spsi;
};
static_assert(sizeof(Node) == PRHF_NODE_BYTES, "node size");

__device__ __forceinline__ double qnan() { return __builtin_nan(""); }

// Data errors are reported through a few words of pinned HOST memory that the device sees (prhf_api.cpp): word b is
// set to 1 for status bit b.  Plain stores of the same value - no atomics over the bus, nothing to copy back or to
// reset on the device: the host reads the words after the synchronisation and clears them itself.
__device__ __forceinline__ void post_status(unsigned* status, unsigned bits) {
    for (int b = 0; b < PRHF_STATUS_WORDS; ++b)
        if (bits & (1u << b)) __hip_atomic_store(status + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cross-lane exchanges without address registers.  __shfl_xor / __shfl_up compile to ds_bpermute_b32 with the
// source lane in a VGPR; inside the persistent loop the compiler hoists those twelve lane patterns out of the
// loop, runs out of registers and reloads them from scratch one by one where they are used - the running maximum
// of a profile waited 2 us for six such reloads.  ds_swizzle (lane ^ 1..16), v_permlane32_swap (the two halves
// of the wave, gfx950) and DPP row shifts carry the pattern in the instruction.
template <int OFF>
__device__ __forceinline__ double lane_xor(double v) {          // the value of lane ^ OFF, OFF = 1, 2, 4, 8, 16
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (OFF << 10) | 0x1f);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (OFF << 10) | 0x1f);
    return __hiloint2double(hi, lo);
}
template <int OFF>
__device__ __forceinline__ int lane_xor(int v) {
    return __builtin_amdgcn_ds_swizzle(v, (OFF << 10) | 0x1f);
}
// own[l] and other[l] are v[l] and v[l ^ 32] in SOME order (the lower half gets them as named, the upper half
// swapped): enough for any symmetric combination
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void halves(double v, double* own, double* other) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const uint2v a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const uint2v b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    *own = __hiloint2double((int)b.x, (int)a.x);
    *other = __hiloint2double((int)b.y, (int)a.y);
}
__device__ __forceinline__ void halves(int v, int* own, int* other) {
    const uint2v a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    *own = (int)a.x;
    *other = (int)a.y;
}
// (the order of the steps - 32, 16, ..., 1 - is the order the shuffle versions had: same sums bit for bit)
__device__ __forceinline__ double wave_max(double v) {
    double p, q;
    halves(v, &p, &q);
    v = fmax(p, q);
    v = fmax(v, lane_xor<16>(v)); v = fmax(v, lane_xor<8>(v)); v = fmax(v, lane_xor<4>(v));
    v = fmax(v, lane_xor<2>(v)); v = fmax(v, lane_xor<1>(v));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    double p, q;
    halves(v, &p, &q);
    v = fmin(p, q);
    v = fmin(v, lane_xor<16>(v)); v = fmin(v, lane_xor<8>(v)); v = fmin(v, lane_xor<4>(v));
    v = fmin(v, lane_xor<2>(v)); v = fmin(v, lane_xor<1>(v));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    double p, q;
    halves(v, &p, &q);
    v = p + q;
    v = v + lane_xor<16>(v); v = v + lane_xor<8>(v); v = v + lane_xor<4>(v);
    v = v + lane_xor<2>(v); v = v + lane_xor<1>(v);
    return v;
}
// Inclusive running maximum over the lanes, and the same shifted up by one lane (-inf into lane 0): DPP row
// shifts inside the rows of 16, row_bcast:15 / :31 across them.
#define PRHF_DPP_MAX(v, ctrl, rows) do {                                                                              \
        const double ninf_ = -__builtin_inf();                                                                       \
        const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(ninf_), __double2loint(v), ctrl, rows, 0xf, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(ninf_), __double2hiint(v), ctrl, rows, 0xf, false); \
        v = fmax(v, __hiloint2double(hi_, lo_));                                                                     \
    } while (0)
__device__ __forceinline__ double wave_scan_max(double v) {
    PRHF_DPP_MAX(v, 0x111, 0xf);    // row_shr:1
    PRHF_DPP_MAX(v, 0x112, 0xf);    // row_shr:2
    PRHF_DPP_MAX(v, 0x114, 0xf);    // row_shr:4
    PRHF_DPP_MAX(v, 0x118, 0xf);    // row_shr:8
    PRHF_DPP_MAX(v, 0x142, 0xa);    // row_bcast:15 into rows 1 and 3
    PRHF_DPP_MAX(v, 0x143, 0xc);    // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double wave_shift_up_1(double v, double into_lane_0) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(into_lane_0), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(into_lane_0), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);              // wave_shr:1
}

// Wave-uniform values computed by the vector ALU live in VGPRs unless told otherwise; moving
// them to SGPRs keeps the hot loop under the 128-VGPR budget of two workgroups per CU.
__device__ __forceinline__ double uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 1/sqrt(x) to ~0.6 ulp: v_rsq_f64 is good to 2^-24 on gfx950 (tools/probe_math.hip), one
// third-order step cubes that.  NaN for x < 0, +inf for x == 0; no range fix-ups (arguments
// here are O(1e-30 .. 1e3), far from the subnormal range).
__device__ __forceinline__ double rsqrt_cubic(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    const double e = __builtin_fma(-t, y, 1.0);            // 1 - x y^2
    const double p = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, p, y);                     // y (1 + e/2 + 3 e^2/8)
}

// One Newton step instead: 2^-48 (measured).  Enough for X mode, whose answer is conditioned like its inputs (the
// reference's own +-1 ulp response is 3e-11) - and, since round 5, for the reduced algebra in O mode as well, which
// is only taken where 1 - X > 1e-5: there the 3.5e-15 it leaves in beta and w move a virtual height by ~2e-12 (median
// distance from the reference on 35 000 pairs of config 3: 2.4e-12 -> 4.3e-12; share within 1e-6, worst pair and the
// counts under the reference's own noise rule unchanged: profiles/omode_onewton_r05.txt) for one instruction less per
// rsqrt - config 3 -2.5 %, O/500 -3.5 %, O/2000 and O/20000 -2.5 ... -4 % (profiles/ab_r05_onewton.txt).
__device__ __forceinline__ double rsqrt_newton(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    const double e = __builtin_fma(-t, y, 1.0);
    return __builtin_fma(y * e, 0.5, y);                   // y (1 + e/2)
}

// THIRD: the third-order step.  The reduced algebra takes it in O mode where nothing guards its use (the opt-in "fast"
// tier everywhere: next to reflection, where 1 - X is tiny, the Newton step's 3.5e-15 would show - share of config-like
// pairs within 1e-6 of the reference 98.0 % -> 96.0 %) and the Newton step where 1 - X > 1e-5 is checked (the default
// O-mode arithmetic's main loops), in X mode and in the tracers' guarded levels.
template <bool THIRD>
__device__ __forceinline__ double rsqrt_tier(double x) {
    return THIRD ? rsqrt_cubic(x) : rsqrt_newton(x);
}

// ---------------------------------------------------------------------------------------
// Appleton-Hartree group index, reference operation order (library.py:194-256).
// ---------------------------------------------------------------------------------------
// sin and cos of a field angle given in degrees, rounded the way the reference's libm rounds them (prhf_crmath.h)
__device__ __forceinline__ void faithful_sincos(double psi_deg, double* s_out, double* c_out) {
#pragma clang fp contract(off)
    const double r = psi_deg * kDegToRad;
    double s, c;
    if (__builtin_fabs(r) < 1.0e6) prhf_cr::sincos_table(r, &s, &c);
    else sincos(r, &s, &c);
    *s_out = s;
    *c_out = c;
}

// sin^2 of an angle given in degrees for the REDUCED algebra (index_fast), to 2e-16 absolute - the reference's order
// takes faithful_sincos.  r = n pi/2 + y with |y| <= pi/4 (two-piece Cody-Waite reduction, exact enough up to |r| = 1e5);
// sin^2 r is sin^2 y for even n and 1 - sin^2 y for odd n, so one odd polynomial (fdlibm's kernel_sin coefficients,
// 2^-58 on that interval) serves every quadrant: ~22 instructions for the device library's ~90.
__device__ __forceinline__ double sin_sq_deg(double psi_deg) {
#pragma clang fp contract(fast)
    const double r = psi_deg * kDegToRad;
    if (!(__builtin_fabs(r) < 1.0e5)) {                // (never a field angle; NaN comes through here as NaN)
        const double sl = sin(r);
        return sl * sl;
    }
    const double n = __builtin_rint(r * 0.6366197723675814);
    double y = __builtin_fma(-n, 1.5707963267948966, r);
    y = __builtin_fma(-n, 6.123233995736766e-17, y);
    const double z = y * y;
    double p = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    p = __builtin_fma(z, p, 2.75573137070700676789e-06);
    p = __builtin_fma(z, p, -1.98412698298579493134e-04);
    p = __builtin_fma(z, p, 8.33333333332248946124e-03);
    p = __builtin_fma(z, p, -1.66666666666666324348e-01);
    const double sy = __builtin_fma(y * z, p, y);
    const double s2 = sy * sy;
    return ((int)n & 1) ? 1.0 - s2 : s2;
}

// ... from sin(psi) and cos(psi) (faithful_sincos): callers that evaluate many frequencies on the same levels compute
// those once per level (the tracers' per-profile table, prhf_snell.inc)
template <int MODE>
__device__ __forceinline__ void index_faithful_sc(double X, double Y, double s, double c, double* mu_out,
                                                  double* mup_out) {
#pragma clang fp contract(off)
    constexpr double sgn = (MODE == PRHF_KMODE_O) ? 1.0 : -1.0;   // library.py:221-224
    // sin, cos, YT**4 and YT**3 rounded the way the reference's libm / NumPy pow round them (prhf_crmath.h):
    // near reflection D cancels to 1e-9 of its terms and these roundings decide the last grid points
    const double YT = Y * s;                                   // :210
    const double YL = Y * c;                                   // :211
    const double Xm1 = 1.0 - X;                                // :214
    const double YT2 = YT * YT;
    const double YL2 = YL * YL;
    const double Xm12 = Xm1 * Xm1;
    const double alpha = 0.25 * prhf_cr::pow4(YT) + YL2 * Xm12;  // :217 (YT**4 is a pow in NumPy)
    const double beta = sqrt(alpha);                           // :218
    const double D = (Xm1 - 0.5 * YT2) + sgn * beta;           // :229
    const double XXm1 = X * Xm1;
    const double q = XXm1 / D;
    double rad = 1.0 - q;                                      // :232
    if (rad < 0.0) rad = qnan();                               // :233
    double mu = sqrt(rad);
    if (mu > 1.0) mu = qnan();                                 // :238
    const double dbdX = ((-YL2) * Xm1) / beta;                 // :241
    const double dDdX = -1.0 + sgn * dbdX;                     // :242
    const double dadY = prhf_cr::pow3(YT) * s + ((2.0 * YL) * Xm12) * c;   // :244-245
    const double dbdY = (0.5 * dadY) / beta;                   // :246
    const double dDdY = (-YT) * s + sgn * dbdY;                // :247
    const double two_mu = 2.0 * mu;
    const double dmudY = (XXm1 * dDdY) / (two_mu * (D * D));   // :250
    const double dmudX = (1.0 / (two_mu * D)) * (((2.0 * X) - 1.0) + q * dDdX);   // :251
    *mu_out = mu;
    *mup_out = mu - ((2.0 * X) * dmudX + Y * dmudY);           // :254
}

template <int MODE>
__device__ __forceinline__ void index_faithful(double X, double Y, double psi_deg, double* mu_out,
                                               double* mup_out) {
    double s, c;
    faithful_sincos(psi_deg, &s, &c);
    index_faithful_sc<MODE>(X, Y, s, c, mu_out, mup_out);
}

// The same mu and mu' in reduced form.  With S2 = sin^2 psi, Y2 = Y^2, a = 1 - X, s = +1 (O) / -1 (X):
//   h = YT^2/2,  t = YL^2 a,  alpha = h^2 + t a,  beta = sqrt(alpha),  D = a - h + s beta          (:217-229)
//   mu^2 = N/D with N = D - X a;  w = 1/sqrt(N D) gives mu = |N w| and 1/(2 mu D) = sign(D) w / 2   (:232)
//   dD/dX = -1 - s t/beta,  Y dD/dY = -2h + s (beta + h^2/beta)                              (:241-247)
//   mu' = mu - [2X (2X - 1 + q dD/dX) + q Y dD/dY] / (2 mu D),  q = X a / D = 1 - mu^2       (:250-254)
// Collecting the bracket and eliminating beta + (h^2 - 2Xt)/beta through D itself
// (s h^2/beta = D - a + h - s t a/beta), and then D mu^2 = N and N - X^2 = D - X, leaves
//   mu' = sign(D) w [ D - X + q (1 + s t (1 + X) / (2 beta)) ]
// - 6 operations after w (checked against the long form to 9e-16 on 1e5 random points per mode).
// YL^2 = Y2 - Y2 S2 (absolute error 1e-16 Y2: harmless, YL^2 only enters through alpha and t).
// index_fast_core leaves the validity test (:233, :238) to the caller and hands out q = X(1-X)/D.
// The reference keeps a point when 0 <= fl(1 - q) and sqrt(fl(1 - q)) <= 1.  In vacuum (X -> 0) mu
// is 1 to rounding, so the upper cliff must sit where the reference's does:
//   sqrt(fl(1 - q)) > 1  <=>  fl(1 - q) >= 1 + 2^-51  <=>  q <= -1.5 * 2^-52  (ties go to even);
// below, mu^2 < 0 makes w (and with it q) NaN.  Hence: keep the point iff q > kQCliff.
constexpr double kQCliff = -3.3306690738754696e-16;   // -1.5 * 2^-52

template <int MODE, bool THIRD = (MODE == PRHF_KMODE_O)>
__device__ __forceinline__ void index_fast_core(double X, double Y2, double S2, double* mu_out,
                                                double* mup_out, double* q_out) {
#pragma clang fp contract(fast)
    constexpr double sgn = (MODE == PRHF_KMODE_O) ? 1.0 : -1.0;
    const double Xm1 = 1.0 - X;
    const double YT2 = Y2 * S2;
    const double YL2 = Y2 - YT2;                               // Y^2 cos^2 psi
    const double h = 0.5 * YT2;
    const double h2 = h * h;
    const double t = YL2 * Xm1;
    const double alpha = h2 + t * Xm1;
    const double rbeta = rsqrt_tier<THIRD>(alpha);
    const double D = (Xm1 - h) + sgn * (alpha * rbeta);
    const double XXm1 = X * Xm1;
    const double N = D - XXm1;
    const double w = rsqrt_tier<THIRD>(N * D);                 // NaN when mu^2 < 0 (:233)
    const double Nw = N * w;
    const double mu = __builtin_fabs(Nw);
    // q = X(1-X)/D = 1 - mu^2.  The callers test q against the mu > 1 cliff at the 1e-16 level and need its
    // sign exact: q = X(1-X) * (N w^2).  (The main loop, group_index_lean, has no such test.)
    const double q = XXm1 * (Nw * w);
    const double Sp1 = (t * rbeta) * ((0.5 * sgn) * X + (0.5 * sgn)) + 1.0;     // 1 + s t (1 + X) / (2 beta)
    const double U = q * Sp1 + (D - X);
    *mu_out = mu;
    *mup_out = __builtin_copysign(w, D) * U;
    *q_out = q;
}

template <int MODE>
__device__ __forceinline__ void index_fast(double X, double Y2, double S2, double* mu_out, double* mup_out) {
    double mu, mup, q;
    index_fast_core<MODE>(X, Y2, S2, &mu, &mup, &q);
    if (!(q > kQCliff)) { mu = qnan(); mup = qnan(); }         // :233, :238
    *mu_out = mu;
    *mup_out = mup;
}

// Isotropic plasma (library.py:201-207): mu = sqrt(1-X) for X < 1, mu' = 1/mu.
__device__ __forceinline__ void index_unmagnetised(double X, double* mu_out, double* mup_out) {
#pragma clang fp contract(off)
    const double m2 = 1.0 - X;
    if (!(m2 > 0.0)) { *mu_out = qnan(); *mup_out = qnan(); return; }
    const double mu = sqrt(m2);
    *mu_out = mu;
    *mup_out = 1.0 / mu;
}

// ---------------------------------------------------------------------------------------
// S3-S6: reflection height of one pair.  Returns false when the frequency escapes.
// np.interp(1.0, running_max, alt) semantics (library.py:388-407): j = last level whose
// running maximum is <= 1; exact hit returns alt[j]; otherwise linear between j and j+1.
// Always IEEE arithmetic: the comparisons with 1.0 decide NaN masks.
// pf2: f_N^2 per level for X mode; its RUNNING MAXIMUM over the levels for O mode (prefix_max_in_place).
// ---------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ bool reflection_height(const Node* __restrict__ nodes,
                                                  const double* __restrict__ pf2,
                                                  const double* __restrict__ gb, int K, double f_hz,
                                                  double f2, int lane, double* h_out) {
#pragma clang fp contract(off)
    double lmax = -__builtin_inf();
    int kstar = K;
    double col_star = 0.0;
    if (MODE == PRHF_KMODE_O) {
        // O mode compares one rounded quotient with 1, which needs no division: fl(a / b) > 1 <=> a / b >
        // 1 + 2^-53 (round to nearest even) <=> a - b > b 2^-53, where a - b is exact for b <= a <= 2b
        // (Sterbenz) and beyond 2b the test is true either way.  The running maximum of the quotients is the
        // quotient of the running maximum (division is monotone), and the running maximum of f_N^2 does not
        // depend on the frequency: pf2 holds it (prefix_max_in_place, once per profile).  It is non-decreasing,
        // so the first level above 1 comes from a 64-ary search - two LDS round trips for any K <= 4096 - and
        // the maximum below it is the entry before: two divisions per pair, no scan, no wave reduction, the
        // same values bit for bit.
        const double ulp_half = f2 * 0x1p-53;
        const int stride = (K + 63) >> 6;
        const int probe = min((lane + 1) * stride - 1, K - 1);
        const unsigned long long coarse = __ballot((pf2[probe] - f2) > ulp_half);
        double amax = -__builtin_inf(), a_star = 0.0;
        if (coarse) {
            const int lo = (__ffsll((long long)coarse) - 1) * stride;
            const int k = min(lo + lane, K - 1);
            const unsigned long long fine = __ballot(lane < stride && (pf2[k] - f2) > ulp_half);
            kstar = lo + __ffsll((long long)fine) - 1;
            a_star = pf2[kstar];
            if (kstar > 0) amax = pf2[kstar - 1];
        } else {
            amax = pf2[K - 1];
        }
        lmax = amax / f2;                       // :136; -inf stays -inf (no level below the first hit)
        col_star = a_star / f2;
    } else {
        for (int base = 0; base < K; base += 64) {
            const int k = base + lane;
            double col = -__builtin_inf();
            if (k < K) col = pf2[k] / f2 + gb[k] / f_hz;            // :136, :157, :389
            const unsigned long long hit = __ballot(col > 1.0);
            if (hit) {
                const int first = __ffsll((long long)hit) - 1;
                kstar = base + first;
                col_star = __shfl(col, first);
                if (lane < first) lmax = fmax(lmax, col);
                break;
            }
            lmax = fmax(lmax, col);
        }
    }
    const double below = (MODE == PRHF_KMODE_O) ? lmax : wave_max(lmax);        // running maximum at level kstar-1
    double h;
    if (kstar == K) {
        if (!(below >= 1.0)) return false;      // never reaches the cutoff (:399)
        h = nodes[K - 1].alt;                   // running max == 1 exactly at the top level
    } else if (kstar == 0) {
        h = nodes[0].alt;                       // already above cutoff at the bottom: left clamp
    } else {
        const int j = kstar - 1;
        const double aj = nodes[j].alt;
        if (below == 1.0) {
            h = aj;
        } else {
            const double slope = (nodes[j + 1].alt - aj) / (col_star - below);
            h = slope * (1.0 - below) + aj;
        }
    }
    *h_out = h - kBackoff;                      // :407
    return true;
}

struct BlockInfo {
    int K;            // bottomside levels (index of the density peak)
    int bad;          // PRHF_STATUS_* bits for this profile
    int unmag;        // isotropic branch
    int uniform;      // altitude grid is uniform below the peak
    int poly_angle;   // every segment has a 2 cos^2(psi) polynomial: 1 cubic, 2 quadratic (all u3 = 0), 3 linear (u2 = 0 too);
                      // 4: every segment in rotation form (some turn the field by 3e-4 .. 0.05 rad); 0: some need sin() per point
    int n_cand;       // entries of the candidate list (frequencies that may reflect), -1: no list, every frequency
    int nan_b;        // a NaN in |B| below the peak: in X mode the reference's whole trace is NaN (run_block)
    int nan_p;        // a NaN in psi below the peak
    const double* heights;   // O mode with a candidate list: reflection height of every entry (they all reflect); else null
    double a0;        // alt[0]
    double inv_w;     // hint buckets per km
    double inv_step;  // 1 / level spacing (uniform grids)
};

// Per-profile scalars that are read once per pair only live in LDS (in the reduction scratch, behind its
// 10 rows) rather than in SGPRs: the main loop needs every scalar register it can get.
enum { kKeepAltMin = 0,     // min over the whole altitude column (:507)
       kKeepPf2Max = 1,     // max f_N^2 over the bottomside levels
       kKeepGbMax = 2 };    // g_p max|B| over the bottomside levels
template <int THREADS>
__device__ __forceinline__ const double* kept_scalars(const double* red) { return red + 10 * (THREADS / 64); }

constexpr int kHintBuckets = PRHF_HINT_BUCKETS;
// BlockInfo::bad beyond the PRHF_STATUS_* bits: every frequency of this profile is NaN and that is NOT an error (the
// reference returns such a row: np.min(alt) of a column with a NaN, :507; X + Y with a NaN |B| below the peak, :389,
// :399).  post_status only looks at the PRHF_STATUS_WORDS low bits.
constexpr int kNanRow = 0x100;
static_assert(kNanRow >= (1 << PRHF_STATUS_WORDS), "kNanRow must not be a status bit");

// -DPRHF_TRACE builds: wall-clock marks of the staging phases of the current block (thread 0 writes them)
#ifdef PRHF_TRACE
__device__ __forceinline__ unsigned long long* trace_marks() {
    __shared__ unsigned long long marks[4];
    return marks;
}
#define PRHF_MARK(i) do { if (threadIdx.x == 0) trace_marks()[i] = wall_clock64(); } while (0)
#else
#define PRHF_MARK(i) do { } while (0)
#endif

// Stage one profile into LDS.  Every thread of the block calls this.
template <int TIER, int THREADS>
__device__ __forceinline__ BlockInfo stage_profile(const double* __restrict__ den,
                                                   const double* __restrict__ bmag,
                                                   const double* __restrict__ bpsi,
                                                   const double* __restrict__ alt,
                                                   const double* __restrict__ freq, int n_freq,
                                                   int n_alt, Node* nodes, double* pf2, double* gb,
                                                   unsigned short* hint, double* red, int capacity) {
#pragma clang fp contract(off)
    constexpr int W = THREADS / 64;
    static_assert(10 * W + 3 <= PRHF_RED_DOUBLES, "reduction scratch too small");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // red rows of W doubles: 0 peak value, 1 peak index, 2 alt min, 3 freq min,
    //                        4 |B| max, 5 negative density, 6 max angle step, 7 non-uniform grid,
    //                        8 f_N^2 max below the peak, 9 NaN in the density or altitude column
    // ---- phase 1: first-occurrence argmax of density, min altitude, min frequency ---------
    double bv = -__builtin_inf();
    int bi = 0x7fffffff;
    double amin = __builtin_inf();
    // NaN inputs behave as they do in the reference (fixture G13).  np.argmax ranks a NaN above every number and
    // returns the first one (library.py:371): a density column padded with NaN is cut at the padding.  A NaN altitude
    // makes np.min(alt) - and with it the whole trace - NaN (:507).  A NaN in |B| below the peak makes the X-mode
    // trace NaN (np.maximum.accumulate of X + Y, :389, :399); in O mode, like a NaN in psi in either mode, it blanks
    // the grid points of the two segments next to that level (np.interp), which the sum then skips (:288) - the
    // generic loop does exactly that (phase 2 keeps such a profile out of the main loop).
    int nan_in = 0;
    for (int i = tid; i < n_alt; i += THREADS) {
        const double v = den[i], al = alt[i];
        const double key = (v != v) ? __builtin_inf() : v;
        if (key > bv) { bv = key; bi = i; }
        amin = fmin(amin, al);
        nan_in |= (al != al) ? 1 : 0;
    }
    double fm = __builtin_inf();
    for (int i = tid; i < n_freq; i += THREADS) fm = fmin(fm, fabs(freq[i]));
    {
        // first occurrence of the maximum: a symmetric choice between the two halves, then lane ^ 16 ... ^ 1
        double v0, v1;
        int i0, i1;
        halves(bv, &v0, &v1);
        halves(bi, &i0, &i1);
        const bool second = v1 > v0 || (v1 == v0 && i1 < i0);
        bv = second ? v1 : v0;
        bi = second ? i1 : i0;
#define PRHF_ARGMAX_STEP(OFF) do {                                                  \
            const double ov = lane_xor<OFF>(bv);                                    \
            const int oi = lane_xor<OFF>(bi);                                       \
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }             \
        } while (0)
        PRHF_ARGMAX_STEP(16); PRHF_ARGMAX_STEP(8); PRHF_ARGMAX_STEP(4); PRHF_ARGMAX_STEP(2); PRHF_ARGMAX_STEP(1);
#undef PRHF_ARGMAX_STEP
    }
    amin = wave_min(amin);
    fm = wave_min(fm);
    nan_in = __any(nan_in) ? 1 : 0;
    if (lane == 0) {
        red[wave] = bv;
        red[W + wave] = (double)bi;
        red[2 * W + wave] = amin;
        red[3 * W + wave] = fm;
        red[9 * W + wave] = (double)nan_in;
    }
    __syncthreads();
    bv = red[0];
    bi = (int)red[W];
    amin = red[2 * W];
    fm = red[3 * W];
    nan_in = (int)red[9 * W];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const double ov = red[w];
        const int oi = (int)red[W + w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        amin = fmin(amin, red[2 * W + w]);
        fm = fmin(fm, red[3 * W + w]);
        nan_in |= (int)red[9 * W + w];
    }
    PRHF_MARK(0);
    BlockInfo info;
    info.K = uniform((bi == 0x7fffffff) ? 0 : bi);      // library.py:371-375: levels [0, argmax)
    // every thread holds the same reduced value and writes it: a wave reads back what it wrote itself
    red[10 * W + kKeepAltMin] = amin;
    fm = uniform(fm);
    info.bad = 0;
    info.unmag = 0;
    info.uniform = 0;
    info.poly_angle = 0;
    info.n_cand = -1;
    info.a0 = 0.0;
    info.inv_w = 0.0;
    info.inv_step = 0.0;
    const int K = info.K;
    info.nan_b = 0;
    info.nan_p = 0;
    if (uniform(nan_in)) {
        info.bad = kNanRow;
        info.K = 0;
        return info;
    }
    if (K == 0) {
        info.bad = PRHF_STATUS_PEAK0;
        return info;
    }
    if (K >= capacity) {                           // cannot happen: the host sized the staged arrays for the highest
        info.bad = PRHF_STATUS_BADINDEX;           // peak of the launch (launch_peak_levels: the same argmax rule) -
        info.K = 0;                                // but an overrun of LDS must not be what tells us otherwise
        return info;
    }
    // ---- phase 2: nodes (values, np.interp slopes), f_N^2, g_p B, and the per-profile flags ------
    const double step0 = (K > 1) ? alt[1] - alt[0] : 1.0;
    double bmax = 0.0, pmax = 0.0;
    int neg = 0, ragged = 0, trig = 0, cubic = 0, quadratic = 0, steep = 0;
    for (int k = tid; k <= K; k += THREADS) {
        Node nd;
        if (k == K) {                              // sentinel: no abscissa is >= +inf
            nd.alt = __builtin_inf();
            nd.den = nd.sden = nd.b = nd.sb = nd.u0 = nd.u1 = nd.u2 = nd.u3 = nd.off = nd.psi = nd.spsi = 0.0;
            nodes[k] = nd;
            continue;
        }
        const double a = alt[k], d = den[k], b = bmag[k], p = bpsi[k];
        nd.alt = a; nd.den = d; nd.b = b; nd.off = 0.0;
        double spsi = 0.0, turn = 0.0;
        if (k + 1 < K) {
            const double da = alt[k + 1] - a;
            const double dp = bpsi[k + 1] - p;
            nd.sden = (den[k + 1] - d) / da;       // numpy arr_interp: (dy[i+1]-dy[i])/(dx[i+1]-dx[i])
            nd.sb = (bmag[k + 1] - b) / da;
            spsi = dp / da;
            turn = fabs(dp) * kDegToRad;
            // "uniform" must hold to 1e-11 per step: the main loop's closed-form segment index then misses a
            // level by at most 1e-8 of a step, and evaluating a linear piece that far outside its segment is
            // harmless (see lean_step); looser grids go through the hint table
            ragged |= (fabs(da - step0) > 1e-11 * fabs(step0)) ? 1 : 0;
        } else {
            nd.sden = 0.0; nd.sb = 0.0;
        }
        nd.psi = p;
        nd.spsi = spsi;
        nd.off = alt[0] - a;
        if (turn < kPolyAngle) {
            // sin^2(psi_j + r dz) = S + sin(2 psi_j) r dz + cos(2 psi_j) (r dz)^2 - (2/3) sin(2 psi_j) (r dz)^3 + O(4)
            const double r = spsi * kDegToRad;
            double sp, cp;
            prhf_cr::sincos_table(p * kDegToRad, &sp, &cp);      // (100 vector instructions; the device library's: 155)
            const double s2p = 2.0 * (sp * cp), c2p = (cp - sp) * (cp + sp);
            nd.u0 = sp * sp;
            nd.u1 = s2p * r;
            nd.u2 = c2p * (r * r);
            nd.u3 = (-2.0 / 3.0) * s2p * (r * r * r);
            if (turn < kQuadAngle) {
                // Chebyshev economisation on [0, L]: x^3 ~ (3L/2) x^2 - (9L^2/16) x + L^3/32, error
                // |u3| L^3 / 32 <= turn^3 / 48: the main loop then evaluates one FMA less per point
                const double L = (k + 1 < K) ? alt[k + 1] - a : 0.0;
                nd.u0 = nd.u0 + nd.u3 * (L * L * L) * (1.0 / 32.0);
                nd.u1 = nd.u1 - nd.u3 * (L * L) * (9.0 / 16.0);
                nd.u2 = nd.u2 + nd.u3 * L * 1.5;
                nd.u3 = 0.0;
                // ... and x^2 ~ L x - L^2/8 on [0, L], error |u2| L^2 / 8: for fields that turn by ~1e-5 rad per
                // level (PyIRI) the quadratic term is below kLinTol and the main loop saves another FMA
                if (fabs(nd.u2) * (L * L) * 0.125 <= kLinTol) {
                    nd.u0 = nd.u0 - nd.u2 * (L * L) * 0.125;
                    nd.u1 = nd.u1 + nd.u2 * L;
                    nd.u2 = 0.0;
                } else {
                    quadratic = 1;
                }
            } else {
                cubic = 1;
            }
            // stored as 2 cos^2(psi) = 2 - 2 sin^2(psi): the main loop wants Y_L^2 first (group_index_lean)
            nd.u0 = 2.0 - 2.0 * nd.u0; nd.u1 = -2.0 * nd.u1; nd.u2 = -2.0 * nd.u2; nd.u3 = -2.0 * nd.u3;
        } else {                                   // this segment turns the field too far for the cubic:
            nd.u0 = nd.u1 = nd.u2 = 0.0;           // u3 = NaN is the flag; the generic loop takes sin() of psi + spsi dz,
            nd.u3 = qnan();                        // the main loop needs the rotation form below (whole profile)
            trig = 1;
            steep |= (turn > kTrigAngle) ? 1 : 0;
        }
        // a NaN in |B| or psi (here or at the level above: then the slopes are NaN): the generic loop, whose sum skips
        // the blanked points one by one
        steep |= (b != b || p != p || nd.sb != nd.sb || spsi != spsi) ? 1 : 0;
        nodes[k] = nd;
        const double fn = sqrt(d) * kPlasma;       // :96
        pf2[k] = fn * fn;                          // :136 numerator
        gb[k] = kGyro * b;                         // :157 numerator
        pmax = fmax(pmax, fn * fn);
        bmax = fmax(bmax, fabs(b));
        neg |= (d < 0.0) ? 1 : 0;
        neg |= (b != b) ? 2 : 0;                   // NaN in |B| below the peak
        neg |= (p != p) ? 4 : 0;                   // ... in psi
        // some SAMPLED |B| is a number (np.nanmax over the regridded array, :201): grid point 0 sits on level 0, any
        // other grid point inside a segment - a number only when both its ends are
        neg |= ((k == 0 && b == b) || (k + 1 < K && b == b && bmag[k + 1] == bmag[k + 1])) ? 8 : 0;
    }
    bmax = wave_max(bmax);
    pmax = wave_max(pmax);
    neg = (__any(neg & 1) ? 1 : 0) | (__any(neg & 2) ? 2 : 0) | (__any(neg & 4) ? 4 : 0) | (__any(neg & 8) ? 8 : 0);
    ragged = __any(ragged) ? 1 : 0;
    // 5: some segment turns the field by more than kTrigAngle (sin() per point, generic loop), 1: some segment needs the
    // rotation form, 2: some segment keeps its cubic, 3: some keeps its quadratic, 0: all linear
    trig = __any(steep) ? 5 : (__any(trig) ? 1 : (__any(cubic) ? 2 : (__any(quadratic) ? 3 : 0)));
    if (lane == 0) {
        red[4 * W + wave] = bmax;
        red[5 * W + wave] = (double)neg;
        red[6 * W + wave] = (double)trig;
        red[7 * W + wave] = (double)ragged;
        red[8 * W + wave] = pmax;
    }
    __syncthreads();
    pmax = red[8 * W];
    bmax = red[4 * W];
    neg = (int)red[5 * W];
    trig = (int)red[6 * W];
    ragged = (int)red[7 * W];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        bmax = fmax(bmax, red[4 * W + w]);
        pmax = fmax(pmax, red[8 * W + w]);
        neg |= (int)red[5 * W + w];
        {
            const int o = (int)red[6 * W + w];          // order of need: sin() 5 > rotation 1 > cubic 2 > quadratic 3 > linear 0
            trig = (trig == 5 || o == 5) ? 5
                 : ((trig == 1 || o == 1) ? 1 : ((trig == 2 || o == 2) ? 2 : ((trig == 3 || o == 3) ? 3 : 0)));
        }
        ragged |= (int)red[7 * W + w];
    }
    bmax = uniform(bmax);
    red[10 * W + kKeepPf2Max] = pmax;
    red[10 * W + kKeepGbMax] = kGyro * bmax;
    neg = uniform(neg);
    trig = uniform(trig);
    ragged = uniform(ragged);
    if (neg & 1) info.bad = PRHF_STATUS_NEGDEN;    // library.py:93-94
    info.nan_b = (neg & 2) ? 1 : 0;
    info.nan_p = (neg & 4) ? 1 : 0;
    // library.py:201: nanmax|Y| < y_tol over the call's whole (F, N) array.  |Y| is largest
    // at the lowest frequency and the strongest field; the node maximum bounds the sampled
    // maximum from above and equals it unless |B| < ~4e-18 T ("Deviations").
    // (np.nanmax of nothing but NaN is NaN, which is not below the tolerance: a |B| column that leaves no sampled value
    //  a number - NaN at every level below the peak, or at one end of every segment and at level 0 - stays on the
    //  magnetised formulas and gives NaN everywhere, as in the reference)
    info.unmag = ((neg & 8) && (kGyro * bmax) / (fm * 1e6) < kUnmagTol) ? 1 : 0;
    info.poly_angle = trig == 5 ? 0 : (trig == 1 ? 4 : (trig == 2 ? 1 : (trig == 3 ? 2 : 3)));
    if (info.poly_angle == 4) {
        // Rotation form for EVERY segment of this profile: 2 cos^2(psi_j + r x) = 1 + cos(2 psi_j) cos(theta) -
        // sin(2 psi_j) sin(theta), theta = 2 r x: u0 = cos 2psi_j, u1 = 2 r [rad/km], u2 = sin 2psi_j (lean_step<POLY 4>).
        // u3 stays / becomes NaN: the generic loop (tails, fallbacks) keeps taking sin() of the interpolated angle.
        for (int k = tid; k < K; k += THREADS) {
            double sp, cp;
            prhf_cr::sincos_table(nodes[k].psi * kDegToRad, &sp, &cp);
            nodes[k].u0 = (cp - sp) * (cp + sp);
            nodes[k].u1 = 2.0 * (nodes[k].spsi * kDegToRad);
            nodes[k].u2 = 2.0 * (sp * cp);
            nodes[k].u3 = qnan();
        }
        __syncthreads();
    }
    // ---- phase 3: segment lookup: closed form when uniform, else a hint table --------------
    const double a0 = uniform(nodes[0].alt);
    const double span = uniform(nodes[K - 1].alt) - a0;
    info.a0 = a0;
    info.uniform = (!ragged && K > 1 && step0 > 0.0) ? 1 : 0;
    info.inv_step = info.uniform ? 1.0 / step0 : 0.0;
    info.inv_w = (span > 0.0) ? (double)kHintBuckets / span : 0.0;
    if (!info.uniform) {
        // hint[b] = last level with alt <= a0 + b*w
        const double w = span / (double)kHintBuckets;
        for (int b = tid; b < kHintBuckets; b += THREADS) {
            const double t = a0 + (double)b * w;
            int lo = 0, hi = K - 1;               // invariant: alt[lo] <= t (alt[0] = a0 <= t)
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (nodes[mid].alt <= t) lo = mid; else hi = mid - 1;
            }
            hint[b] = (unsigned short)lo;
        }
        __syncthreads();
    }
    return info;
}

// First guess of np.interp's segment for abscissa z (exact after fix_segment).
__device__ __forceinline__ int guess_segment(const unsigned short* __restrict__ hint, const BlockInfo& info,
                                             double z) {
    int j;
    if (info.uniform) {
        j = (int)((z - info.a0) * info.inv_step);
        j = j < 0 ? 0 : (j > info.K - 1 ? info.K - 1 : j);
    } else {
        int bucket = (int)((z - info.a0) * info.inv_w);
        bucket = bucket < 0 ? 0 : (bucket > kHintBuckets - 1 ? kHintBuckets - 1 : bucket);
        j = hint[bucket];
    }
    return j;
}

// mu' at abscissa offset dz = z - alt_j >= 0 inside the segment that starts at node nd.
template <int MODE, int TIER, bool UNMAG>
__device__ __forceinline__ double point_mup(const Node& nd, double dz, double f_hz, double f2, double cX,
                                            double cY2, bool poly_angle, double well_conditioned) {
    double mu, mup;
    // np.interp returns fp[j] itself for an abscissa that sits ON level j (numpy arr_interp: x == xp[j]) whatever the
    // slope: it only shows when the slope is NaN (a NaN at level j + 1) - grid point 0 always sits on level 0
    const double sden_ = dz == 0.0 ? 0.0 : nd.sden, sb_ = dz == 0.0 ? 0.0 : nd.sb, spsi_ = dz == 0.0 ? 0.0 : nd.spsi;
    if (TIER == 0) {
#pragma clang fp contract(off)
        const double den = sden_ * dz + nd.den;        // numpy arr_interp: slope*(x - xp[j]) + fp[j]
        if (!UNMAG) {
            // Where 1 - X is not small at any of the wave's points, the operation order does not matter
            // (both forms agree to 1e-12 there) and the reduced algebra - no divide, no sqrt - is used;
            // near X = 1, where the reference's rounding decides the answer, its order is kept.
            const double Xq = den * cX;
            if (__all(1.0 - Xq > well_conditioned)) {
#pragma clang fp contract(fast)
                const double b = sb_ * dz + nd.b;
                index_fast<MODE>(Xq, (b * b) * cY2, sin_sq_deg(spsi_ * dz + nd.psi), &mu, &mup);
                return mup;
            }
        }
        const double fn = sqrt(den) * kPlasma;         // :96
        const double X = (fn * fn) / f2;               // :136
        if (UNMAG) {
            index_unmagnetised(X, &mu, &mup);
        } else {
            const double b = sb_ * dz + nd.b;
            const double psi = spsi_ * dz + nd.psi;
            const double Y = (kGyro * b) / f_hz;       // :157
            index_faithful<MODE>(X, Y, psi, &mu, &mup);
        }
    } else {
#pragma clang fp contract(fast)
        const double den = sden_ * dz + nd.den;
        const double X = den * cX;                     // cX = cp^2 / f^2
        if (UNMAG) {
            index_unmagnetised(X, &mu, &mup);
        } else {
            const double b = sb_ * dz + nd.b;
            const double Y2 = (b * b) * cY2;           // cY2 = (g_p / f)^2
            double S2;
            if (poly_angle || nd.u3 == nd.u3) {        // per segment; poly_angle: true for the whole profile
                S2 = 1.0 - 0.5 * (nd.u0 + dz * (nd.u1 + dz * (nd.u2 + dz * nd.u3)));    // the nodes hold 2 cos^2
            } else {
                S2 = sin_sq_deg(spsi_ * dz + nd.psi);
            }
            index_fast<MODE>(X, Y2, S2, &mu, &mup);
        }
    }
    return mup;
}

// The group index of the main loop: mu' from the interpolated density, field strength and sin^2(psi), in the
// reduced form of index_fast_core specialised for the integration path below the reflection height, where
// D > 0 (X < 1 in O mode, X + Y < 1 in X mode, at every level and therefore between the levels: the three
// interpolants are linear) - no sign transfer, no validity compare - and written around a = 1 - X:
//   G = s beta - h,  D = a + G,  N = D - X a = G + a^2,  D - X = D - cX den,  (1 + X)/2 = k
// The angle enters as C2 = 2 cos^2 psi and the field as hY2 = Y^2 / 2: Y_L^2 = hY2 C2 and h = Y_T^2 / 2 =
// hY2 - Y_L^2 / 2 are then one product and one FMA (through sin^2 psi: Y_T^2, Y_L^2 = Y^2 - Y_T^2, h - three).
// The difference costs h an absolute error of 1e-16 Y^2 where psi is small - and there h itself no longer matters.
// 28 instructions from (den, Y^2/2, C2) - with Y^2/2 = (cY2/2) b^2 (two more) 30: 16 FMA, 11 MUL, 1 ADD,
// 2 v_rsq_f64 (the version through X needed 34).
template <int MODE, bool THIRD>
__device__ __forceinline__ double group_index_lean(double den, double hY2, double C2, double cX, double* a_out) {
#pragma clang fp contract(fast)
    constexpr double sgn = (MODE == PRHF_KMODE_O) ? 1.0 : -1.0;
    const double a = __builtin_fma(-cX, den, 1.0);             // 1 - X
    const double YL2 = hY2 * C2;                               // Y^2 cos^2 psi  (hY2 = Y^2 / 2, C2 = 2 cos^2 psi)
    const double h = __builtin_fma(-0.5, YL2, hY2);            // Y^2 sin^2 psi / 2
    const double t = a * YL2;
    const double ta = t * a;
    const double alpha = __builtin_fma(h, h, ta);
    const double rbeta = rsqrt_tier<THIRD>(alpha);
    const double G = __builtin_fma(sgn * alpha, rbeta, -h);    // s beta - h
    const double D = a + G;
    const double N = __builtin_fma(a, a, G);                   // D - X (1 - X)
    const double w = rsqrt_tier<THIRD>(N * D);                 // NaN when mu^2 < 0 (:233)
    const double DmX = __builtin_fma(-cX, den, D);
    const double v = __builtin_fma(-0.5, ta, t);               // t (1 + X) / 2 = t (1 - a / 2): t a is there already
    const double Sp1 = __builtin_fma(sgn * v, rbeta, 1.0);     // 1 + s t (1 + X) / (2 beta)
    const double Nw = N * w;
    const double q = __builtin_fma(-Nw, Nw, 1.0);              // X (1 - X) / D = 1 - mu^2
    const double U = __builtin_fma(Sp1, q, DmX);
    *a_out = a;
    return w * U;
}

// One grid point per lane on the lean path (uniform altitude grid or hint table, every segment on the
// sin^2 polynomial, span > 0).  g = (m_i, weight_i) - the weight is m_i+1 - m_i from the pair table, or what
// the caller put there (0 for a masked lane, 1e-6 / span for the last grid point, :415-416); returns
// acc + mu' * weight - the caller multiplies the sum by span once (:415).
// A NaN term (mu^2 < 0 by rounding, :233) makes the sum NaN: the caller notices and re-runs the pair through
// the generic loop, whose nansum skips such terms one by one (:288).  Validity needs no compare here: with
// D > 0, q = X(1-X)/D >= 0, so the mu > 1 cliff (:238) cannot trigger.
// CHECK: also report (in `viol`, a lane mask) the points whose 1 - X is not above `wc` - the default O-mode
// arithmetic only accepts wave-iterations where the reduced algebra is safe.
// POLY: degree of the 2 cos^2 psi polynomial every segment of the profile carries: 3 cubic, 2 economised
// quadratic (u3 = 0, not read), 1 economised linear (u2 = 0 too, not read).
// HINT: non-uniform altitude grid - kj is hint buckets per unit of m, the segment comes from the hint table
// (last level at or below the bucket's left edge) plus a walk up the levels inside the bucket.
// The monotone segment cursor of the HINT variant: the lane's current segment and the altitude of the level above it.
// A lane's grid points come in ascending order (lean_loop_body), so its abscissae only grow and the segment index only
// moves up: where no lane of the wave crosses a level - almost always, the stretched grid is much denser than the
// levels - the step costs one compare and a ballot and no LDS read at all (a bucket lookup and a walk per point -
// two dependent LDS round trips - made non-uniform altitude grids 1.5x slower than uniform ones).
struct SegCursor {
    int j;
    double above;       // nodes[j + 1].alt; level K is a +inf sentinel, so a walk stops by itself
};
// Where the main loop finds a staged profile's nodes: in LDS (G = false: a 32-bit LDS byte address, kept in a VGPR) or -
// a profile with more levels than LDS holds (vfo_tall_kernel) - in the workgroup's slab of global memory (G = true),
// read through a buffer resource with the byte offset in a VGPR, like the pair table: the same one vector instruction
// per node address either way.  The loop is the same; what differs is the latency it has to cover.
typedef double node_vec2 __attribute__((ext_vector_type(2)));
template <bool G> struct NodeSpace;
template <> struct NodeSpace<false> {
    unsigned base;
    __device__ __forceinline__ double f64(unsigned off) const {
        typedef __attribute__((address_space(3))) const double* LdsDouble;
        return *(LdsDouble)(uintptr_t)(base + off);
    }
    __device__ __forceinline__ node_vec2 v2(unsigned off) const {
        typedef __attribute__((address_space(3))) const node_vec2* LdsVec2;
        return *(LdsVec2)(uintptr_t)(base + off);
    }
};
template <> struct NodeSpace<true> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ double f64(unsigned off) const {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
        double d;
        __builtin_memcpy(&d, &v, sizeof d);
        return d;
    }
    __device__ __forceinline__ node_vec2 v2(unsigned off) const {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        node_vec2 d;
        __builtin_memcpy(&d, &v, sizeof d);
        return d;
    }
};
// the argument that names the nodes across the (not inlined) loop function: an LDS address or a pointer
template <bool G> struct NodeArg { typedef unsigned type; };
template <> struct NodeArg<true> { typedef const Node* type; };
template <bool G>
__device__ __forceinline__ double lds_alt(const NodeSpace<G>& ns, int j) {
    return ns.f64(__umul24((unsigned)j, (unsigned)sizeof(Node)));
}
// np.interp's segment of abscissa z = m span + a0 from the hint table (last level at or below the bucket's left edge)
// plus a walk up the levels inside the bucket: once per loop call and lane
template <bool G>
__device__ __forceinline__ SegCursor cursor_at(double m0, double span, double a0, double kj, const NodeSpace<G>& nodes_v, unsigned hint_v) {
#pragma clang fp contract(fast)
    typedef __attribute__((address_space(3))) const unsigned short* LdsU16;
    SegCursor c;
    // (a bucket < kHintBuckets: m <= 1 and kj < kHintBuckets, checked by the caller)
    c.j = *(LdsU16)(uintptr_t)(hint_v + 2u * (unsigned)(int)(m0 * kj));
    const double z = __builtin_fma(m0, span, a0);
    c.above = lds_alt(nodes_v, c.j + 1);
    while (__any(z >= c.above)) {
        if (z >= c.above) {
            ++c.j;
            c.above = lds_alt(nodes_v, c.j + 1);
        }
    }
    return c;
}

template <int MODE, bool CHECK, int POLY, bool HINT, bool G>
__device__ __forceinline__ double lean_step(double2 g, double span, double a0, double kj, double cX,
                                            double hcY2, double acc, double wc, unsigned long long& viol,
                                            const NodeSpace<G>& nodes_v, SegCursor& cur) {
#pragma clang fp contract(fast)
    const double m0 = g.x;
    typedef node_vec2 vec2;
    int j;
    if (HINT) {
        const double z = __builtin_fma(m0, span, a0);
        // np.interp's segment: alt[j] <= z < alt[j+1]; alt[cur.j] <= z holds from this lane's previous point
        while (__any(z >= cur.above)) {
            if (z >= cur.above) {
                ++cur.j;
                cur.above = lds_alt(nodes_v, cur.j + 1);
            }
        }
        j = cur.j;
    } else {
        // the pair table holds m clamped to [0, 1] (NaN -> 0), and kj = span / step <= K - 1 with
        // span <= alt[K-1] - alt[0]: (int)(m * kj) is inside [0, K-1] without a clamp here.  (A float
        // "magic number" add cannot replace the conversion: it rounds to nearest, and the interpolants are
        // anchored at the level BELOW the point; the bias of -1/2 that would fix that is not representable
        // next to 2^52.)
        j = (int)(m0 * kj);
    }
    const unsigned pn = __umul24((unsigned)j, (unsigned)sizeof(Node));
    double off = nodes_v.f64(pn + 8);
    const vec2 r_dd = nodes_v.v2(pn + 16), r_bb = nodes_v.v2(pn + 32), r_ua = nodes_v.v2(pn + 48);
    vec2 r_ub;
    if (POLY == 1) { r_ub.x = 0.0; r_ub.y = 0.0; }
    else if (POLY == 2 || POLY == 4) { r_ub.x = nodes_v.f64(pn + 64); r_ub.y = 0.0; }
    else r_ub = nodes_v.v2(pn + 64);
    double2 dd = make_double2(r_dd.x, r_dd.y);                     // den, sden
    double2 bb = make_double2(r_bb.x, r_bb.y);                     // b, sb
    double2 ua = make_double2(r_ua.x, r_ua.y);                     // u0, u1
    double2 ub = make_double2(r_ub.x, r_ub.y);                     // u2, u3
    double x = __builtin_fma(m0, span, off);                   // z - alt_j with z = m*span + a0 (:413)
    // The closed-form index can miss by one where z rounds onto a level (or, on a grid that is uniform only
    // to 1e-11 per step, within 1e-8 of a step of one): a linear piece is then evaluated that far outside
    // its segment.  Past the right end that is exact to rounding.  Past the LEFT end of a segment that
    // starts at a zero density it gives a density of -1e-8 x slope, i.e. X ~ -1e-9 at one point; this path
    // has no mu > 1 cliff test such a value could trip (the generic path, which has one, clamps), and the
    // effect on mu' there is of the same size - far below the reference's own noise.
    if (HINT) x = fmax(x, 0.0);                                 // the cursor guarantees alt[j] <= z: rounding only
    const double den = dd.y * x + dd.x;
    const double b = bb.y * x + bb.x;
    double C2;
    if (POLY == 4) {
        // rotation form: 2 cos^2(psi_j + r x) = 1 + cos 2psi_j cos t - sin 2psi_j sin t, t = 2 r x, |t| <= 2 kTrigAngle
        const double t = ua.y * x, t2 = t * t;
        const double ct = 1.0 + t2 * (-0.5 + t2 * (1.0 / 24.0 + t2 * (-1.0 / 720.0 + t2 * (1.0 / 40320.0))));
        const double st = t * (1.0 + t2 * (-1.0 / 6.0 + t2 * (1.0 / 120.0 + t2 * (-1.0 / 5040.0 + t2 * (1.0 / 362880.0)))));
        C2 = __builtin_fma(-ub.x, st, __builtin_fma(ua.x, ct, 1.0));
    } else {
        C2 = POLY == 1 ? ua.x + x * ua.y
                       : (POLY == 2 ? ua.x + x * (ua.y + x * ub.x) : ua.x + x * (ua.y + x * (ub.x + x * ub.y)));
    }
    double a;
    const double mup = group_index_lean<MODE, MODE == PRHF_KMODE_O && !CHECK>(den, hcY2 * (b * b), C2, cX, &a);
    if (CHECK) viol |= __ballot(!(a > wc));
    return __builtin_fma(mup, g.y, acc);
}

// The top segment of a pair in m: about half of a long grid's points lie between the last level below the
// reflection height and the reflection height itself (the stretched grid is dense there).  For those points the
// node is the same for every lane and every trip, so the three interpolants become polynomials in m with
// wave-uniform coefficients - no segment index, no LDS read, no abscissa, Y straight from m: 32 instructions
// per point instead of 37 (with the linear angle polynomial; 34 against 39 with the cubic).
struct TopSegment {
    double d0, d1;          // den  = d0 + d1 m
    double b0, b1;          // Y / sqrt(2) = g_p |B| / (sqrt(2) f) = b0 + b1 m
    double q0, q1, q2, q3;  // 2 cos^2 psi = q0 + m (q1 + m (q2 + m q3))
};
template <bool G>
__device__ __forceinline__ TopSegment top_segment(const NodeSpace<G>& nodes_v, int j, double span, double cY) {
#pragma clang fp contract(fast)
    const unsigned nd = __umul24((unsigned)j, (unsigned)sizeof(Node));
    // Node = {alt, off, den, sden, b, sb, u0, u1, u2, u3, psi, spsi}
    const double o = nodes_v.f64(nd + 8), s = span;            // x = s m + o
    const double den = nodes_v.f64(nd + 16), sden = nodes_v.f64(nd + 24), b = nodes_v.f64(nd + 32), sb = nodes_v.f64(nd + 40),
                 u0 = nodes_v.f64(nd + 48), u1 = nodes_v.f64(nd + 56), u2 = nodes_v.f64(nd + 64), u3 = nodes_v.f64(nd + 72);
    TopSegment t;
    t.d0 = den + sden * o;  t.d1 = sden * s;
    t.b0 = cY * (b + sb * o);  t.b1 = cY * (sb * s);
    t.q0 = u0 + o * (u1 + o * (u2 + o * u3));
    t.q1 = s * (u1 + o * (2.0 * u2 + 3.0 * o * u3));
    t.q2 = (s * s) * (u2 + 3.0 * o * u3);
    t.q3 = (s * s * s) * u3;
    return t;
}
template <int MODE, bool CHECK, int POLY>
__device__ __forceinline__ double lean_step_top(double2 g, const TopSegment& t, double cX, double hcY2,
                                                double acc, double wc, unsigned long long& viol) {
#pragma clang fp contract(fast)
    const double m0 = g.x;
    const double den = t.d1 * m0 + t.d0;
    const double Y = t.b1 * m0 + t.b0;
    const double C2 = POLY == 1 ? t.q0 + m0 * t.q1
                    : (POLY == 2 ? t.q0 + m0 * (t.q1 + m0 * t.q2) : t.q0 + m0 * (t.q1 + m0 * (t.q2 + m0 * t.q3)));
    double a;
    const double mup = group_index_lean<MODE, MODE == PRHF_KMODE_O && !CHECK>(den, Y * Y, C2, cX, &a);
    (void)hcY2;
    if (CHECK) viol |= __ballot(!(a > wc));
    return __builtin_fma(mup, g.y, acc);
}

// The main loop over grid points [first, end) of one pair: returns span * sum of mu' * weight (per lane).
// Whole wave-iterations run two per trip with scalar loop control and no lane masks; a last partial one
// carries its idle lanes along as copies of grid point 0 with weight 0.  `last_special` >= 0 (then
// end == last_special + 1): that grid point is the last of the grid, its thickness is 1e-6 km, not a grid
// step (:415-416) - it always sits in the partial iteration.
// Deliberately NOT inlined: inside the fused kernel ~100 wave-uniform values are live around this loop,
// and whenever the register allocator ran out of SGPRs it parked the buffer descriptor in VGPR lanes and
// paid 8 v_readlane per trip (seen three times while the surrounding code changed).  As a function the
// loop keeps its dozen scalars in SGPRs whatever the caller looks like; the call costs ~100 cycles per pair.
struct LeanResult {
    double acc;     // span * sum of mu' * weight, per lane
    int first;      // first grid point not consumed
};

// CHECK (default O-mode arithmetic): stop in front of the first point with 1 - X <= well_conditioned; the
// caller continues from there in the reference's operation order.
// TOP: with the top-segment phase (long grids on a uniform altitude grid); the short-grid callers use the
// variant without it, which needs 24 fewer vector registers around the call.
template <int MODE, bool CHECK, int POLY, bool HINT, bool TOP, bool G>
__device__ __forceinline__ LeanResult lean_loop_body(typename NodeArg<G>::type nodes_arg, unsigned hint_lds,
                                                     const double2* __restrict__ pairs, int first, int end,
                                                     int last_special, double span, double a0, double kj,
                                                     double cX, double cY2, double well_conditioned) {
#pragma clang fp contract(fast)
    // arguments arrive in VGPRs: back to SGPRs.  The node table travels as its 32-bit LDS address (a
    // generic pointer would turn every node read into a flat load).
    first = uniform(first); end = uniform(end); last_special = uniform(last_special);
    span = uniform(span); a0 = uniform(a0); kj = uniform(kj); cX = uniform(cX);
    const double hcY2 = uniform(0.5 * cY2);            // Y^2 / 2 = hcY2 |B|^2 (group_index_lean)
    const double wc = CHECK ? uniform(well_conditioned) : 0.0;
    pairs = reinterpret_cast<const double2*>(
        ((unsigned long long)(unsigned)uniform((int)((unsigned long long)pairs >> 32)) << 32) |
        (unsigned)uniform((int)(unsigned long long)pairs));
    const int lane = threadIdx.x & 63;
    double a0v = a0;                                   // VGPR copies: v_fma / v_mad take one SGPR operand
    unsigned hint_v = (unsigned)uniform((int)hint_lds);
    NodeSpace<G> nodes_v;
    if constexpr (G) {
        // a slab of global memory: its address back into scalar registers, then a buffer resource over it
        const unsigned long long addr = (unsigned long long)nodes_arg;
        const Node* base = reinterpret_cast<const Node*>(
            ((unsigned long long)(unsigned)uniform((int)(addr >> 32)) << 32) | (unsigned)uniform((int)addr));
        nodes_v.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<Node*>(base), 0, 0x7fffffff, 0x00020000);
        asm volatile("" : "+v"(a0v), "+v"(hint_v));
    } else {
        unsigned nodes_lds_v = (unsigned)uniform((int)nodes_arg);
        asm volatile("" : "+v"(a0v), "+v"(nodes_lds_v), "+v"(hint_v));
        nodes_v.base = nodes_lds_v;
    }
    // Pair-table loads go through a buffer descriptor: lane offset in a VGPR, grid position in an
    // SGPR, so the loop spends no vector instruction on addresses.  A prefetch may reach up to 127 entries
    // past `end`: the table is padded by PRHF_PAIR_PAD entries (launch_grid_pairs), and what is read there
    // is never used.
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(double2);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double2*>(pairs), 0, 0x7fffffff, 0x00020000);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto grid_at = [&](int i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, i * (int)sizeof(double2), 0);
        double2 g;
        __builtin_memcpy(&g, &v, sizeof g);
        return g;
    };
    // whole iterations: all 64 points below `end` and none of them the special last point
    const int whole_end = first + (((last_special >= 0 ? last_special : end) - first) & ~63);
    // Long loops: the stretched grid is dense near the reflection height - ~43 % of a 20 000-point grid lies in the
    // segment of the last point, another ~11 % and ~5 % in the two segments below it.  For the whole wave-iterations
    // inside one of these top kTopSegments segments the node is the same for every lane and every trip
    // (lean_step_top): no segment index, no LDS read.  The first grid point of segment j - the first i with
    // m_i >= m_star(j), the grid position of its left end: j / kj on a uniform altitude grid, else from the node's own
    // offset alt_0 - alt_j - is found in the (monotone) pair table by a 64-ary search: three rounds of one load and
    // one ballot for 20 000 points.  What lies between two such runs - the wave-iteration that straddles a level -
    // and everything below them takes the indexed steps.
    constexpr int kTopSegments = 3;
    static_assert(kTopSegments == 3, "the run loop below selects among three");
    int seg_begin[kTopSegments] = {0, 0, 0}, seg_end[kTopSegments] = {0, 0, 0}, seg_j[kTopSegments] = {0, 0, 0};   // [0]: the top segment
    int n_seg = 0;
    TopSegment top;
    __builtin_memset(&top, 0, sizeof top);
    if (TOP && end - first >= PRHF_TOP_MIN_POINTS) {
        const int i_last = (last_special >= 0 ? last_special : end - 1);
        const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 0, i_last * (int)sizeof(double2), 0);
        double2 gl;
        __builtin_memcpy(&gl, &vl, sizeof gl);
        const int j_top = HINT ? uniform(cursor_at(gl.x, span, a0v, kj, nodes_v, hint_v).j) : uniform((int)(gl.x * kj));
        int search_hi = i_last + 1;                    // m[search_hi - 1] >= m_star of the segment searched next
        int run_end = whole_end;                       // whole wave-iterations of that segment end here
        // (the two segments below the top one hold ~16 % of the points: worth their searches - three dependent loads
        //  each - on long grids only; measured: +5 % on 2000 points, -2.6 % on 20 000)
        const int max_seg = (end - first) >= PRHF_TOP3_MIN_POINTS ? kTopSegments : 1;
#pragma unroll
        for (int sidx = 0; sidx < kTopSegments; ++sidx) {
            const int j = j_top - sidx;
            if (j < 0 || n_seg != sidx || sidx >= max_seg) break;
            double m_star;
            if (HINT) {
                const double off_j = nodes_v.f64(__umul24((unsigned)j, (unsigned)sizeof(Node)) + 8u);
                m_star = uniform(-off_j / span);
            } else {
                m_star = uniform((double)j / kj);
            }
            int lo = first, hi = search_hi;            // the answer lies in [lo, hi)
            bool found = true;
            // A guess in closed form first: on the reference's stretched grid (library.py:314-320; the only one the Python
            // side makes) m_i = 1 - (e^(10 (1 - u_i)) - 1) / (e^10 - 1), u_i = i / (N - 1), so the first i with
            // m_i >= m_star is ceil((N - 1) (1 - ln(1 + (1 - m_star)(e^10 - 1)) / 10)) - one v_log_f32, good to 1e-3 of
            // an index here.  Four table entries around it are loaded at once and the answer is taken only if they
            // bracket it (m below m_star in the first, at or above it in a later one): any other grid - the C ABI takes
            // every non-decreasing multiplier - or a guess at the range's edge falls through to the search below, which
            // costs three dependent loads per segment.  The index is the search's own either way.
            {
                const double A = __builtin_fma(1.0 - m_star, 22025.465794806718, 1.0);
                const double ln_a = (double)__builtin_amdgcn_logf((float)A) * 0.6931471805599453;      // (v_log_f32: log2)
                const double n1 = (double)(CHECK ? i_last + 1 : i_last);                                // N - 1
                const int base = uniform((int)__builtin_fma(-ln_a, n1 * 0.1, n1)) - 1;                  // floor(i*) - 1
                if (base >= lo && base + 3 < hi) {
                    const u32x4 vg = __builtin_amdgcn_raw_buffer_load_b128(
                        rsrc, (unsigned)(base + (lane & 3)) * (unsigned)sizeof(double2), 0, 0);
                    double2 gg;
                    __builtin_memcpy(&gg, &vg, sizeof gg);
                    const unsigned hit4 = (unsigned)(__ballot(gg.x >= m_star) & 0xfull);
                    if (hit4 != 0u && (hit4 & 1u) == 0u) {
                        lo = base + __ffs((int)hit4) - 1;
                        hi = lo + 1;
                    }
                }
            }
            while (hi - lo > 1) {
                const int stride = (hi - lo + 63) >> 6;
                const int probe = min(lo + (lane + 1) * stride - 1, hi - 1);
                const u32x4 vp = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)probe * (unsigned)sizeof(double2), 0, 0);
                double2 gp;
                __builtin_memcpy(&gp, &vp, sizeof gp);
                const unsigned long long hit = __ballot(gp.x >= m_star);
                if (!hit) { found = false; break; }    // (a table that is not monotone: no run)
                const int L = __ffsll((long long)hit) - 1;
                const int nlo = lo + L * stride;
                hi = min(nlo + stride, hi);
                lo = nlo;
            }
            if (!found) break;
            const int aligned = first + ((lo - first + 63) & ~63);
            if (aligned + 128 > run_end) break;        // not worth a loop of its own (nor are the sparser ones below)
            seg_begin[sidx] = aligned;
            seg_end[sidx] = run_end;
            seg_j[sidx] = j;
            n_seg = sidx + 1;
            search_hi = lo + 1;
            run_end = first + ((lo - first) & ~63);
        }
    }
    const bool top_phase = n_seg > 0;
    // Y / sqrt(2) = g_p |B| / (sqrt(2) f) = cYs |B|; sqrt(x) = x rsqrt(x) to 0.6 ulp in 8 instructions (the library's: ~25)
    const double cYs = top_phase ? uniform(hcY2 * rsqrt_cubic(hcY2)) : 0.0;
    double accm = 0.0;                                 // sum of mu' * weight
    unsigned long long viol = 0;
    double2 g0 = grid_at(first);
    SegCursor cur;
    cur.j = 0;
    cur.above = 0.0;
    if (HINT) cur = cursor_at(first + lane < end ? g0.x : uniform(g0.x), span, a0v, kj, nodes_v, hint_v);
    // whole wave-iterations [first, stop) with the indexed steps, two per trip so that the prefetch registers swap
    // roles without moves
    auto run_indexed = [&](int stop) {
        for (; first + 128 <= stop; first += 128) {
            const double2 g1 = grid_at(first + 64);
            if (!CHECK) {
                accm = lean_step<MODE, false, POLY, HINT, G>(g0, span, a0v, kj, cX, hcY2, accm, wc, viol, nodes_v, cur);
                g0 = grid_at(first + 128);
                accm = lean_step<MODE, false, POLY, HINT, G>(g1, span, a0v, kj, cX, hcY2, accm, wc, viol, nodes_v, cur);
            } else {
                unsigned long long viol2 = 0;
                const double a1 = lean_step<MODE, true, POLY, HINT, G>(g0, span, a0v, kj, cX, hcY2, accm, wc, viol, nodes_v, cur);
                const double2 g2 = grid_at(first + 128);
                const double a2 = lean_step<MODE, true, POLY, HINT, G>(g1, span, a0v, kj, cX, hcY2, a1, wc, viol2, nodes_v, cur);
                if (viol) {                            // keep the points in front of the first one that fails
                    const int L = __ffsll((long long)viol) - 1;
                    if (lane < L) accm = a1;
                    first += L;
                    return;
                }
                if (viol2) {
                    const int L = __ffsll((long long)viol2) - 1;
                    accm = lane < L ? a2 : a1;
                    first += 64 + L;
                    viol = viol2;
                    return;
                }
                accm = a2;
                g0 = g2;
            }
        }
        if (first + 64 <= stop) {                      // odd whole wave-iteration left over
            const double a1 = lean_step<MODE, CHECK, POLY, HINT, G>(g0, span, a0v, kj, cX, hcY2, accm, wc, viol, nodes_v, cur);
            if (!(CHECK && viol)) {
                accm = a1;
                first += 64;
                g0 = grid_at(first);
            } else {
                const int L = __ffsll((long long)viol) - 1;
                if (lane < L) accm = a1;
                first += L;
            }
        }
    };
    // ... and with the node of one segment held in registers
    auto run_top = [&](int stop) {
        for (; first + 128 <= stop; first += 128) {
            const double2 g1 = grid_at(first + 64);
            if (!CHECK) {
                accm = lean_step_top<MODE, false, POLY>(g0, top, cX, hcY2, accm, wc, viol);
                g0 = grid_at(first + 128);
                accm = lean_step_top<MODE, false, POLY>(g1, top, cX, hcY2, accm, wc, viol);
            } else {
                unsigned long long viol2 = 0;
                const double a1 = lean_step_top<MODE, true, POLY>(g0, top, cX, hcY2, accm, wc, viol);
                const double2 g2 = grid_at(first + 128);
                const double a2 = lean_step_top<MODE, true, POLY>(g1, top, cX, hcY2, a1, wc, viol2);
                if (viol) {
                    const int L = __ffsll((long long)viol) - 1;
                    if (lane < L) accm = a1;
                    first += L;
                    return;
                }
                if (viol2) {
                    const int L = __ffsll((long long)viol2) - 1;
                    accm = lane < L ? a2 : a1;
                    first += 64 + L;
                    viol = viol2;
                    return;
                }
                accm = a2;
                g0 = g2;
            }
        }
        if (first + 64 <= stop) {
            const double a1 = lean_step_top<MODE, CHECK, POLY>(g0, top, cX, hcY2, accm, wc, viol);
            if (!(CHECK && viol)) {
                accm = a1;
                first += 64;
                g0 = grid_at(first);
            } else {
                const int L = __ffsll((long long)viol) - 1;
                if (lane < L) accm = a1;
                first += L;
            }
        }
    };
    if (TOP) {
        // (one copy of the two loops: the run's bounds are picked with scalar selects, not by unrolling)
#pragma unroll 1
        for (int sidx = n_seg - 1; sidx >= 0 && !(CHECK && viol); --sidx) {
            const int begin_s = sidx == 0 ? seg_begin[0] : (sidx == 1 ? seg_begin[1] : seg_begin[2]);
            const int end_s = sidx == 0 ? seg_end[0] : (sidx == 1 ? seg_end[1] : seg_end[2]);
            const int j_s = sidx == 0 ? seg_j[0] : (sidx == 1 ? seg_j[1] : seg_j[2]);
            run_indexed(begin_s);
            if (!(CHECK && viol)) {
                top = top_segment(nodes_v, j_s, span, cYs);
                run_top(end_s);
            }
        }
    }
    if (!top_phase && !(CHECK && viol)) run_indexed(whole_end);
    if (!(CHECK && viol) && first < end) {             // partial wave-iteration: at most 64 points left
        const int idx = first + lane;
        const bool is_last = idx == last_special;
        const bool live = idx < end;
        double2 g = g0;
        if (is_last) g.y = kBackoff / span;            // :415-416: the last thickness is 1e-6 km
        double a1;
        if (TOP && top_phase) {
            // an idle lane re-evaluates this iteration's first point (lane 0 is always live) with weight 0
            if (!live) g = make_double2(uniform(g0.x), 0.0);
            a1 = lean_step_top<MODE, CHECK, POLY>(g, top, cX, hcY2, accm, wc, viol);
        } else {
            // an idle lane re-evaluates a live point with weight 0: grid point 0 - or, under the cursor (whose lanes may
            // only move up), this iteration's first point, which lies above everything the lane has seen
            if (!live) g = make_double2(HINT ? uniform(g0.x) : 0.0, 0.0);
            a1 = lean_step<MODE, CHECK, POLY, HINT, G>(g, span, a0v, kj, cX, hcY2, accm, wc, viol, nodes_v, cur);
        }
        if (!(CHECK && viol)) {
            accm = a1;
            first = end;
        } else {
            // (an idle lane copies a live point of lower index, so the first failing lane is a live one)
            const int L = __ffsll((long long)viol) - 1;
            if (lane < L) accm = a1;
            first += L;
        }
    }
    LeanResult r;
    r.acc = accm * span;                               // :415: dh = (m_i+1 - m_i) * span
    r.first = first;
    return r;
}

template <int MODE, bool CHECK, int POLY, bool HINT, bool TOP, bool G>
__device__ __attribute__((noinline)) LeanResult lean_loop(typename NodeArg<G>::type nodes_arg, unsigned hint_lds,
                                                          const double2* __restrict__ pairs, int first, int end,
                                                          int last_special, double span, double a0, double kj,
                                                          double cX, double cY2, double well_conditioned) {
    return lean_loop_body<MODE, CHECK, POLY, HINT, TOP, G>(nodes_arg, hint_lds, pairs, first, end, last_special, span, a0,
                                                            kj, cX, cY2, well_conditioned);
}

// ---------------------------------------------------------------------------------------
// S7-S11 for grid points [i0, i1) of one pair; returns this wave's partial sum (all lanes).
// The multiplier loads of iteration n+1 are issued before the arithmetic of iteration n.
// ---------------------------------------------------------------------------------------
template <int MODE, int TIER, bool UNMAG, bool G>
__device__ __forceinline__ double integrate_chunk(const Node* __restrict__ nodes,
                                                  const unsigned short* __restrict__ hint,
                                                  const BlockInfo& info, const double* __restrict__ mult,
                                                  const double2* __restrict__ pairs, int n_points, int i0,
                                                  int i1, double f_hz, double f2, double cX, double cY2,
                                                  double h_refl, int lane, double well_conditioned) {
    const int K = info.K;
    const double a0 = info.a0;
    const double span = uniform(h_refl - a0);      // :413 (critical_height - aalt[0])
    const bool poly_angle = info.poly_angle != 0 && info.poly_angle != 4;      // every segment on a polynomial
    const bool lean_angle = info.poly_angle != 0;                              // ... or in rotation form: main loop
    const int last = n_points - 1;
    double acc = 0.0;
    int first = i0;                                // first grid point of the next wave-iteration
    if (!UNMAG && lean_angle && pairs != nullptr && (TIER == 1 || well_conditioned < 1.0)) {
        // Lean main loop of the common case (slowly turning field; uniform altitude grid, or any grid
        // through the hint table).  Fast tier: it takes every grid point of the range, the last one of the
        // grid (thickness 1e-6 km) included, so nothing is left for the generic loop below.  Default O-mode
        // arithmetic: it stops in front of the first point with 1 - X <= well_conditioned (and never takes the
        // last point of the grid); everything from there on goes through the generic loop in the reference's
        // operation order - usually a handful of points next to the reflection height, one wave-iteration.
#pragma clang fp contract(fast)
        first = uniform(first);
        const bool to_grid_end = i1 == n_points;
        const int lean_end = uniform((TIER == 1 || !to_grid_end) ? i1 : i1 - 1);
        const int last_special = (TIER == 1 && to_grid_end) ? last : -1;
        // index scale: (z - a0) / step = m * kj on a uniform grid, hint buckets per unit of m otherwise
        const bool by_hint = !info.uniform;
        // (the bucket scale is biased low by 1e-11: rounding must never select the bucket ABOVE the point,
        // whose hinted level could lie above it too - one bucket too low only lengthens the walk up)
        const double kj = uniform(by_hint ? span * info.inv_w * (1.0 - 1e-11) : span * info.inv_step);
        // span <= 0: left clamp, generic loop; the bound on kj keeps the closed-form index inside its table
        const bool in_table = by_hint ? (kj < (double)kHintBuckets && info.inv_w > 0.0) : (kj <= (double)(K - 1));
        if (lean_end > first && span > 0.0 && in_table) {
            // the loop is a function of its own (not inlined): it gets a fresh scalar-register budget,
            // see lean_loop
            typedef __attribute__((address_space(3))) const Node* LdsNodes;
            typedef __attribute__((address_space(3))) const unsigned short* LdsU16;
            typename NodeArg<G>::type nodes_lds;       // (G: the slab's address itself)
            if constexpr (G) {
                nodes_lds = nodes;
            } else {
                // (opaque to interprocedural constant propagation: told that every caller passes the dynamic-LDS base,
                //  the compiler drops the argument and has the loop function look the base up in a table in memory)
                unsigned at = (unsigned)(uintptr_t)(LdsNodes)nodes;
                asm volatile("" : "+v"(at));
                nodes_lds = at;
            }
            const unsigned hint_lds = (unsigned)(uintptr_t)(LdsU16)hint;
            const int poly = info.poly_angle == 4 ? 4 : 4 - info.poly_angle;    // degree: 3 cubic, 2 quadratic, 1 linear; 4: rotation form
            LeanResult r;
#define PRHF_LEAN(P, H, T) lean_loop<MODE, TIER == 0, P, H, T, G>(nodes_lds, hint_lds, pairs, first, lean_end, last_special, \
                                                                 span, a0, kj, cX, cY2, well_conditioned)
#define PRHF_LEAN_POLY(H, T) (poly == 1 ? PRHF_LEAN(1, H, T) : (poly == 2 ? PRHF_LEAN(2, H, T) : PRHF_LEAN(3, H, T)))
            if (poly == 4) r = by_hint ? PRHF_LEAN(4, true, false) : PRHF_LEAN(4, false, false);
            else if (by_hint) {
                if (lean_end - first >= PRHF_TOP_MIN_POINTS && i0 == 0 && to_grid_end) r = PRHF_LEAN_POLY(true, true);
                else r = PRHF_LEAN_POLY(true, false);
            }
            // (a chunk of a pair keeps to the indexed steps: the search for the top segment's first point costs
            // three dependent loads, which a latency-bound chunked launch cannot hide and every chunk would repeat)
            else if (lean_end - first >= PRHF_TOP_MIN_POINTS && i0 == 0 && to_grid_end) r = PRHF_LEAN_POLY(false, true);
            else r = PRHF_LEAN_POLY(false, false);
#undef PRHF_LEAN_POLY
#undef PRHF_LEAN
            // a NaN term (mu^2 < 0 by rounding, :233) or an infinite one (mu^2 == 0) poisons the lane sum: then
            // the whole range goes through the generic loop, whose nansum drops such terms one by one (:288)
            if (uniform((int)__any(!(__builtin_fabs(r.acc) <= 1.7976931348623157e308)))) {
                acc = 0.0;
                first = uniform(i0);
#ifdef PRHF_MARK_FALLBACK
                acc = 1e6;                          // diagnostics build (tools/count_fallbacks.py): such pairs show up 1e6 km too high
#endif
            } else {
                acc = r.acc;
                first = uniform(r.first);
            }
        }
    }
    // Generic loop over [first, i1).  Its wave-iterations are aligned to the END of the range - a short first
    // one (low lanes idle), full ones after it - so that the points next to the reflection height, the
    // expensive ones (reference order), share one full iteration instead of spilling into a nearly empty one.
    const int n_iter = uniform((i1 - first + 63) >> 6);
    int i = i1 - (n_iter << 6) + lane;             // below `first` only in the first iteration: idle lane
    auto grid = [&](int idx) { return mult[idx < first ? first : (idx < last ? idx : last)]; };
    double m0 = 0.0, m1 = 0.0;
    if (n_iter > 0) {
        m0 = grid(i);
        m1 = grid(i + 1);
    }
    for (int it = 0; it < n_iter; ++it) {
        // prefetch the next iteration's grid values; clamped, never conditional, so that the
        // compiler can count outstanding loads (s_waitcnt vmcnt(2)) instead of draining them
        const int inext = i + 64;
        const double n0 = grid(inext);
        const double n1 = grid(inext + 1);
        const bool active = i >= first;            // an idle lane carries the range's first point along
        double z, dh;
        if (TIER == 0) {
#pragma clang fp contract(off)
            z = m0 * span + a0;                    // :413
            dh = (i < last) ? (m1 * span + a0) - z : kBackoff;     // :415-416
        } else {
            z = __builtin_fma(m0, span, a0);
            dh = (i < last) ? (m1 - m0) * span : kBackoff;
        }
        // segment of np.interp: alt[j] <= z < alt[j+1]; the guess is almost always right
        int j = guess_segment(hint, info, z);
        Node nd = nodes[j];
        const double an = nodes[j + 1].alt;        // node K is a +inf sentinel
        if (__builtin_expect(__any((j > 0 && z < nd.alt) || z >= an), 0)) {
            while (j > 0 && z < nodes[j].alt) --j;
            while (j + 1 < K && z >= nodes[j + 1].alt) ++j;
            nd = nodes[j];
        }
        double dz;
        if (TIER == 0) {
            dz = z - nd.alt;
            if (dz < 0.0) dz = 0.0;                // z below the first level: left value
        } else {
            dz = fmax(__builtin_fma(m0, span, nd.off), 0.0);       // 0: z below the first level, left value
        }
        const double mup = point_mup<MODE, TIER, UNMAG>(nd, dz, f_hz, f2, cX, cY2, poly_angle, well_conditioned);
        const double term = mup * dh;              // :288
        if (active && term == term) acc = acc + term;              // nansum
        i = inext;
        m0 = n0;
        m1 = n1;
    }
    return wave_sum(acc);
}

// In-place inclusive running maximum of v[0..K) by the whole workgroup (O-mode level search).  Every thread owns
// ceil(K / THREADS) consecutive levels; wave scan of the thread maxima, wave totals through `red`.
template <int THREADS>
__device__ __forceinline__ void prefix_max_in_place(double* v, int K, double* red) {
    constexpr int W = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (K + THREADS - 1) / THREADS;
    const int base = tid * per;
    double mine = -__builtin_inf();
    for (int i = 0; i < per; ++i)
        if (base + i < K) mine = fmax(mine, v[base + i]);
    const double inc = wave_scan_max(mine);
    double before = wave_shift_up_1(inc, -__builtin_inf());     // maximum of the lower lanes' levels
    if (lane == 63) red[wave] = inc;               // (rows 0..3 of `red` were phase-1 scratch of stage_profile: free now)
    __syncthreads();
#pragma unroll
    for (int w = 0; w < W; ++w)
        if (w < wave) before = fmax(before, red[w]);
    for (int i = 0; i < per; ++i)
        if (base + i < K) {
            before = fmax(before, v[base + i]);
            v[base + i] = before;
        }
    __syncthreads();
}

// The frequencies of this profile that may reflect, in ascending index order, as a list in LDS; the others -
// those for which the bound of pair_reflects says "escapes for certain" - get their NaN here, with coalesced
// stores, and never become work items.  Whole workgroup; returns the list length (wave-uniform).
//
// O mode with `heights` given (room for one double per frequency): the list is exact.  The running maximum of
// f_N^2 is in LDS (prefix_max_in_place) and does not depend on the frequency, so each THREAD settles one
// frequency - a binary search for the first level above the cutoff and the three divisions of
// reflection_height, the same IEEE operations on the same values - instead of each WAVE settling one pair later
// on: ~2 vector instructions per pair instead of ~80, which is a sixth of a pair's cost on a 200-point grid.
// heights[i] belongs to cand[i]; every listed frequency reflects.
template <int THREADS>
__device__ __forceinline__ int list_candidates(const KArgs& a, const SegDev& sg, const double* keep,
                                               long long pair_base, unsigned short* cand, int* cand_count,
                                               const Node* nodes, const double* pf2, const double* gb, int K,
                                               double* heights) {
    constexpr int W = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = (int)a.n_freq;
    const double pmax = keep[kKeepPf2Max], gmax = keep[kKeepGbMax];
    int total = 0;
    for (int base = 0; base < F; base += THREADS) {
        const int f = base + tid;
        bool may = false;
        double h = 0.0;
        if (f < F) {
            const double* row = a.ftab + 8 * (long long)f;
            if (heights && sg.mode == PRHF_KMODE_X) {
#pragma clang fp contract(off)
                // X mode: X + Y is not monotone in the level, so every thread scans the levels of its frequency -
                // with the table's reciprocals, good to 4 ulp, which decides "above 1" and "the largest so far"
                // for certain outside a band of 1e-13 around 1 and 1e-14 around a tie; the exact quotients
                // (:136, :157, :389) are then needed at two levels only, the ones np.interp reads.  Inside the
                // bands the thread repeats the scan with exact divisions.  Eight levels per trip, loads in front: the
                // scan is a chain of LDS round trips in the prologue, and latency there is what it costs.
                const double f_hz = row[0], f2 = row[1], rf2 = row[4], rf = row[5];
                // The largest value so far is kept as (m1, first level kb of the trip or level that holds it, its
                // length nb); m2 is the largest value OUTSIDE that stretch: a tie between stretches shows as m1 ~ m2,
                // a tie inside the best stretch is settled when its levels are evaluated exactly.
                int ks = K, kb = -1, nb = 0;
                double m1 = -__builtin_inf(), m2 = -__builtin_inf();
                bool exact = false;
                // (a frequency that escapes for certain - the bound of pair_reflects - has nothing to scan)
                const int K_scan = (pmax * rf2 + gmax * rf < 1.0 - 1e-9) ? 0 : K;
                constexpr int U = 8;                    // levels per trip: sixteen LDS reads in flight, one wait
                int k = 0;
                for (; k + U <= K_scan; k += U) {
                    double c[U];
#pragma unroll
                    for (int i = 0; i < U; ++i) c[i] = __builtin_fma(pf2[k + i], rf2, gb[k + i] * rf);
                    const double g = fmax(fmax(fmax(c[0], c[1]), fmax(c[2], c[3])), fmax(fmax(c[4], c[5]), fmax(c[6], c[7])));
                    if (g > 1.0 - 1e-13) break;         // the crossing, or a level inside the band: level by level below
                    const double t = fmin(m1, g);
                    if (g > m1) { kb = k; nb = U; }
                    m1 = fmax(m1, g);
                    m2 = fmax(m2, t);
                }
                for (; k < K_scan; ++k) {               // the trip that stopped the loop above, or the last K % U levels
                    const double c = __builtin_fma(pf2[k], rf2, gb[k] * rf);
                    if (__builtin_fabs(c - 1.0) <= 1e-13) { exact = true; break; }
                    if (c > 1.0) { ks = k; break; }
                    if (c > m1) { m2 = m1; m1 = c; kb = k; nb = 1; }
                    else if (c > m2) m2 = c;
                }
                if (!exact && kb >= 0 && !(m1 - m2 > 1e-14 * __builtin_fabs(m1))) exact = true;
                double below = -__builtin_inf(), col_star = 0.0;
                if (K_scan == 0) {
                    ks = K;                             // below stays -inf: does not reflect
                } else if (exact) {
                    ks = K;
                    for (int kk = 0; kk < K; ++kk) {
                        const double col = pf2[kk] / f2 + gb[kk] / f_hz;
                        if (col > 1.0) { ks = kk; col_star = col; break; }
                        below = fmax(below, col);
                    }
                } else {
                    // exact quotients of the best stretch's levels that can hold the exact maximum
                    for (int i = 0; i < nb; ++i) {
                        const double c = __builtin_fma(pf2[kb + i], rf2, gb[kb + i] * rf);
                        if (c >= m1 - 4e-15 * __builtin_fabs(m1)) below = fmax(below, pf2[kb + i] / f2 + gb[kb + i] / f_hz);
                    }
                    if (ks < K) col_star = pf2[ks] / f2 + gb[ks] / f_hz;
                }
                if (ks == K) {
                    may = below >= 1.0;                 // == 1 exactly at the top level, else it never reaches the cutoff (:399)
                    h = nodes[K - 1].alt;
                } else if (ks == 0) {
                    may = true;
                    h = nodes[0].alt;                   // above the cutoff at the bottom already: left clamp
                } else {
                    may = true;
                    const double aj = nodes[ks - 1].alt;
                    if (below == 1.0) {
                        h = aj;
                    } else {
                        const double slope = (nodes[ks].alt - aj) / (col_star - below);
                        h = slope * (1.0 - below) + aj;
                    }
                }
                h = h - kBackoff;                       // :407
            } else if (heights) {
#pragma clang fp contract(off)
                // first level whose quotient exceeds 1: fl(p / f2) > 1 <=> p - f2 > f2 2^-53 (see reflection_height)
                const double f2 = row[1];
                const double ulp_half = f2 * 0x1p-53;
                int lo = 0, hi = K;                     // answer in [lo, hi]; hi == K: none
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((pf2[mid] - f2) > ulp_half) hi = mid; else lo = mid + 1;
                }
                const int kstar = lo;
                const double below = (kstar > 0 ? pf2[kstar - 1] : -__builtin_inf()) / f2;     // :136
                if (kstar == K) {
                    may = below >= 1.0;                 // == 1 exactly at the top level, else it never reaches the cutoff (:399)
                    h = nodes[K - 1].alt;
                } else if (kstar == 0) {
                    may = true;
                    h = nodes[0].alt;                   // above the cutoff at the bottom already: left clamp
                } else {
                    may = true;
                    const double aj = nodes[kstar - 1].alt;
                    if (below == 1.0) {
                        h = aj;
                    } else {
                        const double col_star = pf2[kstar] / f2;
                        const double slope = (nodes[kstar].alt - aj) / (col_star - below);
                        h = slope * (1.0 - below) + aj;
                    }
                }
                h = h - kBackoff;                       // :407
            } else {
                double ub = pmax * row[4];
                if (sg.mode == PRHF_KMODE_X) ub = ub + gmax * row[5];
                may = !(ub < 1.0 - 1e-9);
            }
            if (!may) a.out[sg.out_off + pair_base + f] = qnan();      // never reaches the cutoff (:399)
        }
        const unsigned long long mask = __ballot(may);
        if (lane == 0) cand_count[wave] = __popcll(mask);
        __syncthreads();
        int at = total, chunk = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int n = cand_count[w];
            if (w < wave) at += n;
            chunk += n;
        }
        // (X mode keeps its heights where f_N^2 lived: every thread has finished reading that by now - one round,
        //  n_freq <= THREADS - and the barriers above lie between)
        if (may) {
            const int pos = at + __popcll(mask & ((1ull << lane) - 1ull));
            cand[pos] = (unsigned short)f;
            if (heights) heights[pos] = h;
        }
        __syncthreads();
        total += chunk;
    }
    return uniform(total);
}

// Per-frequency scalars of a pair.  Long launches read them from a table built once per launch
// (freq_table_kernel: F x 8 doubles, one 64-byte row per frequency, fetched with the wave-uniform index);
// short ones compute them in place.  f_hz, f2 are the reference's own values (:491, f**2); cX, cY2 only
// feed the reduced algebra; inv_f2, inv_f only feed the sufficient "escapes for certain" test.
struct PairFreq {
    double f_hz, f2, cX, cY2, inv_f2, inv_f;
};
__device__ __forceinline__ PairFreq pair_freq(const KArgs& a, int f) {
#pragma clang fp contract(off)
    PairFreq p;
    if (a.ftab) {
        const double* row = a.ftab + 8 * (long long)f;
        p.f_hz = uniform(row[0]); p.f2 = uniform(row[1]); p.cX = uniform(row[2]); p.cY2 = uniform(row[3]);
        p.inv_f2 = uniform(row[4]); p.inv_f = uniform(row[5]);
    } else {
        const double fm = a.freq[f];                           // (not positive and finite: NaN, see freq_table_kernel)
        p.f_hz = uniform((fm > 0.0 && fm < __builtin_inf()) ? fm * 1e6 : qnan());     // :491
        p.f2 = uniform(p.f_hz * p.f_hz);                       // f**2
        p.cX = uniform((kPlasma * kPlasma) / p.f2);
        const double cY = kGyro / p.f_hz;
        p.cY2 = uniform(cY * cY);
        p.inv_f2 = uniform(1.0 / p.f2);
        p.inv_f = uniform(1.0 / p.f_hz);
    }
    return p;
}

// S3-S6 of one pair: does the frequency reflect, and where.  First a sufficient test for "escapes": division
// and addition are monotone, so the reference's X (O mode) or X + Y (X mode) at every level is <= the same
// expression of the two per-profile maxima; when that bound - evaluated with reciprocals, so good to a few
// ulp - stays 1e-9 below 1, the running maximum cannot reach 1 (:399) and the level scan is skipped (it is
// most of the per-pair cost when n_points is small).  Otherwise the scan decides, exactly.
// (K == 1 keeps the full path: np.interp's one-node quirk.)
template <int MODE>
__device__ __forceinline__ bool pair_reflects(const Node* nodes, const double* pf2, const double* gb,
                                              const BlockInfo& info, const double* keep, const PairFreq& pf,
                                              int lane, int list_pos, double* h_out) {
    if (info.heights) {                            // settled when the candidate list was made (list_candidates)
        *h_out = uniform(info.heights[list_pos]);
        return true;
    }
    if (info.K > 1 && info.n_cand < 0) {           // (a candidate list has applied this bound already)
        double ub = keep[kKeepPf2Max] * pf.inv_f2;
        if (MODE == PRHF_KMODE_X) ub = ub + keep[kKeepGbMax] * pf.inv_f;
        if (uniform((int)(ub < 1.0 - 1e-9))) return false;
    }
    double h = 0.0;
    const bool r = uniform((int)reflection_height<MODE>(nodes, pf2, gb, info.K, pf.f_hz, pf.f2, lane, &h)) != 0;
    *h_out = uniform(h);
    return r;
}

// A one-level bottomside: np.interp with a single node returns that node even for the NaN abscissae of an
// escaping frequency (numpy arr_interp, lenxp == 1), so the reference's sum keeps the last term
// mu'(level 0) * 1e-6 (:415-416, :288).
template <int MODE, int TIER>
__device__ __forceinline__ double one_level_term(const Node* nodes, const BlockInfo& info, const PairFreq& pf,
                                                 double well_conditioned) {
    const bool poly = info.poly_angle != 0 && info.poly_angle != 4;
    const double mup = info.unmag
        ? point_mup<MODE, TIER, true>(nodes[0], 0.0, pf.f_hz, pf.f2, pf.cX, pf.cY2, poly, well_conditioned)
        : point_mup<MODE, TIER, false>(nodes[0], 0.0, pf.f_hz, pf.f2, pf.cX, pf.cY2, poly, well_conditioned);
    const double term = mup * kBackoff;
    return (term == term) ? term : 0.0;
}

// A grid that collapses onto the bottom level.  When the cutoff is exceeded at level 0 already (in X mode: every
// frequency below the gyrofrequency there, X + Y > 1 with no plasma at all) the reference's reflection height is
// alt[0] - 1e-6 (:407 on the left clamp), its grid z_i = m_i span + alt[0] with span = -1e-6 lies at or below
// alt[0], np.interp clamps every point to level 0, and the sum is mu'(level 0) times the telescoped thicknesses
//   S = (z_last - z_0) + 1e-6          (:413-416; the differences of neighbours are exact, Sterbenz),
// a few 1e-15 km - or NaN when mu'(level 0) is.  No loop over the grid is needed for that: in the config-4 sweep
// (0.5 - 16 MHz, f_H up to 1.7 MHz) 5 % of the pairs are of this kind, and through the generic loop they took
// 15 % of the kernel's time; in O mode they are the frequencies below the plasma frequency of the bottom level.
// mu'(level 0) is evaluated in the tier's own arithmetic.  Returns false (generic loop) where S is so close to zero
// that the reference's own "sum == 0 -> NaN" test (:290) hangs on the order of its additions (alt[0] == 0).
template <int MODE, int TIER>
__device__ __forceinline__ bool collapsed_grid_sum(const Node* nodes, const BlockInfo& info, const PairFreq& pf,
                                                   const double* __restrict__ mult, int n_points, double h_refl,
                                                   double well_conditioned, double* sum_out) {
#pragma clang fp contract(off)
    const double span = h_refl - info.a0;
    const double z_first = mult[0] * span + info.a0;
    const double z_last = mult[n_points - 1] * span + info.a0;
    const double S = (z_last - z_first) + kBackoff;
    if (!(__builtin_fabs(S) > 1e-21)) return false;
    const bool poly = info.poly_angle != 0 && info.poly_angle != 4;
    const double mup = info.unmag
        ? point_mup<MODE, TIER, true>(nodes[0], 0.0, pf.f_hz, pf.f2, pf.cX, pf.cY2, poly, well_conditioned)
        : point_mup<MODE, TIER, false>(nodes[0], 0.0, pf.f_hz, pf.f2, pf.cX, pf.cY2, poly, well_conditioned);
    const double term = mup * S;
    *sum_out = (term == term) ? term : 0.0;        // every term NaN: nansum gives 0 (:288)
    return true;
}

template <int MODE, int TIER, int THREADS, bool G>
__device__ __forceinline__ void run_items(const KArgs& a, const SegDev& sg, const Node* nodes,
                                          const double* pf2, const double* gb, const unsigned short* hint,
                                          const unsigned short* cand, const BlockInfo& info, long long prof_local,
                                          int block_in_prof, int blocks_per_prof, int* item_next, double* red) {
    constexpr int W = THREADS / 64;
    const int lane = threadIdx.x & 63;
    const double* keep = kept_scalars<THREADS>(red);
    const double wc = uniform(sg.well_conditioned);
    const int F = uniform((int)a.n_freq);
    const int C = uniform(sg.chunks);
    // items: (frequency, chunk), or - unchunked long launches - the entries of the candidate list
    const int T = info.n_cand >= 0 ? info.n_cand : F * C;      // < 2^31: n_freq <= 2^20, chunks <= n_points / 256
    const double* mult = a.mult + sg.mult_off;
    const double2* pairs = (a.pairs && sg.lean) ? reinterpret_cast<const double2*>(a.pairs) + sg.mult_off : nullptr;
    const long long pair_base = prof_local * F;
    const int first_item = block_in_prof * W, round_items = blocks_per_prof * W;
    // Few pairs on a long grid (one profile, the reference's own call; SegDev::slots > 0): a pair is cut into C <= S
    // chunks that all live in THIS workgroup - wave w takes item (block_in_prof * W + w): frequency item / S, chunk
    // item % S, S = 2, 4 or 8 slots per pair - and the workgroup adds the chunk sums itself, in chunk order, through
    // LDS: no scratch in global memory, no second kernel (the launch of vfo_finalize_kernel behind a 21 us kernel
    // cost 7 us), same value whichever wave finishes first.
    // Otherwise items are handed out first come, first served.  The SIMD arbiter favours its older waves: with a
    // fixed share per wave, waves 0-3 of a workgroup were done at 70 % of its life (tools/wave_trace.py)
    // and their slots sat empty for the rest.  Which wave computes a pair does not change its value.
    // Everything that steers this loop is wave-uniform and kept in SGPRs (uniform()), so that the
    // compiler emits scalar branches and not exec-masked loops around the wave-level operations inside.
    // Every lane takes part in the atomic (the compiler folds the 64 increments into one LDS add of 64
    // and hands lane 0 the old value): no lane-divergent branch in this loop's control flow.
    // (One loop, one copy of the item's code for both kinds: a second call site costs the kernel 8 - 13 VGPRs.)
    static_assert(11 * W + 8 <= PRHF_RED_DOUBLES, "reduction scratch too small");
    const int S = uniform(sg.slots), wave = threadIdx.x >> 6;
    const bool local = S > 0;
    double* part = red + 10 * W + 8;                   // (behind the kept scalars; rows 0 .. 9 may still be read by a
    if (local && lane == 0) part[wave] = 0.0;          //  wave that left stage_profile early)
    auto next_item = [&]() {
        const int u = uniform(atomicAdd(item_next, 1)) >> 6;
        // this block's items: rounds of W, interleaved with the profile's other blocks
        return uniform((u / W) * round_items + first_item + (u % W));
    };
    int t = local ? uniform(block_in_prof * W + wave) : next_item();
    const int t_end = local ? uniform(min(F * S, t + 1)) : T;      // (block-local chunks: this wave's one item)
    for (; t < t_end; t = local ? t_end : next_item()) {
        int f = t, c = 0;
        if (local) {
            f = t / S;
            c = t % S;
            if (c >= C) break;                     // (fewer chunks than slots: an idle wave)
        } else if (info.n_cand >= 0) {
            f = uniform((int)cand[t]);
        } else if (C > 1) {                        // (unchunked launches spare the integer division)
            f = t % F;
            c = t / F;
        }
        double result = qnan();
        bool reflects = false;
        if (!info.bad) {
            const PairFreq pf = pair_freq(a, f);
            double h = 0.0;
            reflects = pair_reflects<MODE>(nodes, pf2, gb, info, keep, pf, lane, t, &h);
            double collapsed = 0.0;
            if (reflects && info.K > 1 && h < info.a0 &&
                uniform((int)collapsed_grid_sum<MODE, TIER>(nodes, info, pf, mult, sg.n_points, h, sg.well_conditioned,
                                                            &collapsed))) {
                result = (c == C - 1) ? uniform(collapsed) : 0.0;
            } else if (reflects) {
                const int i0 = c * sg.chunk_len;
                const int i1 = min(sg.n_points, i0 + sg.chunk_len);
                if (info.unmag)
                    result = integrate_chunk<MODE, TIER, true, G>(nodes, hint, info, mult, pairs, sg.n_points, i0, i1,
                                                               pf.f_hz, pf.f2, pf.cX, pf.cY2, h, lane, wc);
                else
                    result = integrate_chunk<MODE, TIER, false, G>(nodes, hint, info, mult, pairs, sg.n_points, i0,
                                                                i1, pf.f_hz, pf.f2, pf.cX, pf.cY2, h, lane, wc);
            } else if (info.K == 1) {
                result = (c == C - 1) ? one_level_term<MODE, TIER>(nodes, info, pf, sg.well_conditioned) : 0.0;
                reflects = true;
            }
        }
        if (lane == 0) {
            if (local) {
                part[wave] = reflects ? result : qnan();
            } else if (C == 1) {
                // :290-292: exact zero means every term was NaN -> NaN; then add min(alt)
                const double vh = (reflects && result != 0.0) ? result + keep[kKeepAltMin] : qnan();
                a.out[sg.out_off + pair_base + f] = vh;
            } else {
                a.partial[sg.partial_off + (pair_base + f) * C + c] = reflects ? result : qnan();
            }
        }
    }
    if (local) {
        __syncthreads();
        const int item = block_in_prof * W + wave;
        const int f = item / S;
        if (f < F && item % S == 0 && lane == 0) {
            double sum = 0.0;
            for (int cc = 0; cc < C; ++cc) sum = sum + part[wave + cc];
            // :290-292: exact zero means every term was NaN -> NaN; then add min(alt) (vfo_finalize_kernel's rule)
            a.out[sg.out_off + pair_base + f] = (sum == sum && sum != 0.0) ? sum + keep[kKeepAltMin] : qnan();
        }
    }
}

}  // namespace

// Returns the wall clock at the end of staging in -DPRHF_TRACE builds (0 otherwise).
template <int TIER, int THREADS, bool G>
__device__ __forceinline__ unsigned long long run_block(const KArgs& a, const SegDev& sg, Node* nodes, double* pf2, double* gb,
                                          unsigned short* hint, unsigned short* cand, int* cand_count, double* red,
                                          long long prof_local, int block_in_prof, int blocks_per_prof,
                                          int* item_next) {
    const long long p = sg.prof_begin + prof_local;
    BlockInfo info = stage_profile<TIER, THREADS>(
        a.den + p * a.prof_stride, a.bmag + p * a.field_stride, a.bpsi + p * a.field_stride,
        a.alt + p * a.alt_stride, a.freq, (int)a.n_freq, (int)a.n_alt, nodes, pf2, gb, hint, red, (int)a.lds_levels);
    PRHF_MARK(1);
    if (sg.mode == PRHF_KMODE_X && info.nan_b && !info.bad) info.bad = kNanRow;     // (see stage_profile)
    if (sg.mode == PRHF_KMODE_O && !info.bad) prefix_max_in_place<THREADS>(pf2, info.K, red);
    PRHF_MARK(2);
    info.n_cand = -1;
    info.heights = nullptr;
    if (a.ftab && !a.no_candidates && sg.chunks == 1 && a.n_freq <= PRHF_MAX_CAND && !info.bad && info.K > 1)
    {
        // Reflection heights settled while the list is made live where the level search kept its input: O mode
        // never reads g_p |B| per level again; X mode (one round of frequencies only) reads neither array again
        double* heights = nullptr;
        if (a.n_freq <= a.lds_levels) {
            if (sg.mode == PRHF_KMODE_O) heights = gb;
            else if (a.n_freq <= THREADS && sg.thread_scan) heights = pf2;
        }
        info.n_cand = list_candidates<THREADS>(a, sg, kept_scalars<THREADS>(red), prof_local * a.n_freq, cand,
                                               cand_count, nodes, pf2, gb, info.K, heights);
        info.heights = heights;
    }
#ifdef PRHF_TRACE
    const unsigned long long t_staged = wall_clock64();
#else
    const unsigned long long t_staged = 0;
#endif
    if (threadIdx.x == 0 && block_in_prof == 0) {
        if (info.bad) post_status(a.status, (unsigned)info.bad);
        if (sg.chunks > 1 && sg.slots == 0) a.altmin[sg.altmin_off + prof_local] = kept_scalars<THREADS>(red)[kKeepAltMin];
    }
    if (sg.mode == PRHF_KMODE_O)
        run_items<PRHF_KMODE_O, TIER, THREADS, G>(a, sg, nodes, pf2, gb, hint, cand, info, prof_local, block_in_prof,
                                               blocks_per_prof, item_next, red);
    else
        run_items<PRHF_KMODE_X, TIER, THREADS, G>(a, sg, nodes, pf2, gb, hint, cand, info, prof_local, block_in_prof,
                                               blocks_per_prof, item_next, red);
    return t_staged;
}

// TIER_SEL 0 / 1: every slice in that tier; 2: each slice in its own tier (mixed launches).
// TALL: profiles of more levels than LDS holds (the reference has no limit, library.py:371-375).  The staged
// profile - nodes, f_N^2, g_p |B| - then lives in a slab of global memory that belongs to this workgroup
// (KArgs::tall, one slab per resident workgroup: the launch is persistent whenever it has more blocks than slots),
// and only the hint table, the reduction scratch and the counters stay in LDS.  Every device function takes the
// staged arrays as generic pointers; the main loop reads its nodes through NodeSpace<G> - LDS words in the LDS
// kernels, the slab behind a buffer resource here (context option tall_lean = 1, the default: 1.1 - 1.4 x the time of
// the LDS path on long grids) - and with tall_lean = 0 the host switches it off for such launches (SegDev::lean = 0,
// no candidate list), so that a tall profile runs through the generic loop like any other profile that leaves the
// main loop (about three times slower).  Same values either way (fixture G13, tests/test_gpu_random.py).
template <int TIER_SEL, int THREADS, bool TALL>
__device__ __forceinline__ void vfo_kernel_body(const KArgs& a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_alt = (int)a.lds_levels;               // (room of the staged arrays: a.n_alt unless the host knows better)
    Node* nodes;
    double* pf2;
    unsigned short* hint;
    if (TALL) {
        unsigned char* slab = a.tall + (size_t)blockIdx.x * a.tall_stride;
        nodes = reinterpret_cast<Node*>(slab);
        pf2 = reinterpret_cast<double*>(slab + (size_t)(n_alt + 1) * sizeof(Node));
        hint = reinterpret_cast<unsigned short*>(smem);
    } else {
        nodes = reinterpret_cast<Node*>(smem);                          // n_alt + 1 nodes
        pf2 = reinterpret_cast<double*>(smem + (size_t)(n_alt + 1) * sizeof(Node));
        hint = reinterpret_cast<unsigned short*>(pf2 + 2 * (size_t)n_alt);
    }
    double* gb = pf2 + n_alt;
    unsigned short* cand = hint + kHintBuckets;
    double* red = reinterpret_cast<double*>(cand + PRHF_MAX_CAND);

    // Long launches are persistent: as many workgroups as the device keeps resident, each pulling the
    // next block of work from a queue when it finishes one.  (Left to the hardware dispatcher, the
    // very uneven workgroup lifetimes - 0.7 to 5.7 ms at n_points = 20000 - leave ~12 % of the
    // workgroup slots empty: measured with tools/wave_trace.py.)  Every wave leaves the loop through
    // the same uniform test, so the grid always drains.
    __shared__ long long next_bid;
    __shared__ int item_next;
    __shared__ int cand_count[PRHF_BLOCK_THREADS / 64 + 1];
    // Follow-up of a short-grid launch: the blocks to evaluate are listed (block_list[1 .. block_list[0]])
    const unsigned* list = a.block_list;
    const long long n_blocks = list ? (long long)list[0] : a.n_blocks;
    // (the last kernel behind a short-grid launch whose blocks were drawn in cost order leaves the class counters at
    //  zero for the next call's short_order_kernel)
    if (a.zero_after && blockIdx.x == 0 && threadIdx.x < PRHF_ORDER_CLASSES) a.zero_after[threadIdx.x] = 0u;
    long long ticket = blockIdx.x;
    if (ticket >= n_blocks) return;
    long long bid = list ? (long long)list[1 + ticket] : ticket;
    for (;;) {
        if (threadIdx.x == 0) item_next = 0;   // ordered before its first use by the barriers of stage_profile
#ifdef PRHF_TRACE
        const unsigned long long t_start = wall_clock64();
#endif
        int s = 0;
        while (s + 1 < a.n_segs && bid >= a.seg[s + 1].block_begin) ++s;
        const SegDev& sg = a.seg[s];
        long long lb = bid - sg.block_begin;
        const long long head_blocks = sg.tail_prof * sg.blocks_per_prof;
        int bpp = sg.blocks_per_prof;
        long long prof0 = 0;
        if (lb >= head_blocks) {               // the slice's tail: more, shorter workgroups per profile
            lb -= head_blocks;
            bpp = sg.tail_bpp;
            prof0 = sg.tail_prof;
        }
        const long long prof_local = prof0 + lb / bpp;
        const int block_in_prof = (int)(lb % bpp);
        // Mixed launches: a slice's workgroups run at a wave priority that falls with their length.  The SIMD arbiter
        // prefers its older waves, and in a persistent launch the workgroup that arrived first on a CU stays the older
        // one for ever: its neighbour's waves get the issue slots it leaves.  With equal work per block that only shifts
        // time between the two; in a mixed list (longest blocks first) the neighbour's long first block was still
        // running when everything else was done - config 5: the 20000-point blocks of the CUs' second workgroups took
        // 4 - 9 ms against 1 - 4 ms for the first ones, and the launch ended on them with 2 - 250 of 512 slots busy
        // for its last sixth (tools/wave_trace5.py).  A long block now outranks the shorter ones that the other
        // workgroup pulls after its own.
        // (Homogeneous launches: raising a block's priority when its second half begins evened out the lives - the
        // longest fell from 12 to 9.6 ms at 20000 points - and left the launch time where it was.)
        int prio = sg.prio;
        if (a.n_segs == 1 && a.queue != nullptr) {
            // one slice: the blocks of the last three resident rounds rank below everything pulled before them, round
            // by round, so that the launch ends on its newest (and, in the last round, quartered) blocks and not on
            // an old one that its neighbour kept waiting
            const long long left = n_blocks - ticket, round = gridDim.x;
            prio = left > 3 * round ? 3 : (left > 2 * round ? 2 : (left > round ? 1 : 0));
        }
        if (a.n_segs > 1 || a.queue != nullptr) {
            if (prio >= 3) __builtin_amdgcn_s_setprio(3);
            else if (prio == 2) __builtin_amdgcn_s_setprio(2);
            else if (prio == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }

        const unsigned long long t_staged = (TIER_SEL == 0 || (TIER_SEL == 2 && sg.tier == 0))
            ? run_block<0, THREADS, TALL>(a, sg, nodes, pf2, gb, hint, cand, cand_count, red, prof_local, block_in_prof, bpp,
                                    &item_next)
            : run_block<1, THREADS, TALL>(a, sg, nodes, pf2, gb, hint, cand, cand_count, red, prof_local, block_in_prof, bpp,
                                    &item_next);
        (void)t_staged;
#ifdef PRHF_TRACE
        if (a.trace && (threadIdx.x & 63) == 0) {
            unsigned long long* t = a.trace + (bid * (THREADS / 64) + (threadIdx.x >> 6)) * 6;
            t[0] = t_start;
            t[1] = wall_clock64();
            t[2] = t_staged;
            t[3] = trace_marks()[0];           // argmax known; [4] nodes staged; [5] running maximum done
            t[4] = trace_marks()[1];
            t[5] = trace_marks()[2];
        }
#endif
        if (a.queue == nullptr) break;
        __syncthreads();                       // every wave is done with the staged profile (and with next_bid)
        if (threadIdx.x == 0) next_bid = (long long)gridDim.x + atomicAdd(a.queue, 1u);
        __syncthreads();
        ticket = uniform((int)next_bid);
        if (ticket >= n_blocks) break;
        bid = list ? (long long)list[1 + ticket] : ticket;
    }
}

template <int TIER_SEL, int THREADS>
__global__ __launch_bounds__(THREADS, PRHF_MIN_WAVES_PER_SIMD) void vfo_kernel(const KArgs a) {
    vfo_kernel_body<TIER_SEL, THREADS, false>(a);
}
// (one instantiation: each slice in its own tier)
template <int THREADS>
__global__ __launch_bounds__(THREADS, PRHF_MIN_WAVES_PER_SIMD) void vfo_tall_kernel(const KArgs a) {
    vfo_kernel_body<2, THREADS, true>(a);
}

// Pair table of the fast tier's main loop: (m_i, m_i+1 - m_i) side by side, one 16-byte load per
// grid point.  Runs over the whole (possibly concatenated) multiplier array; the difference that
// straddles two grids belongs to a last grid point, which the main loop never touches.
// m is clamped to the unit interval the stretched grid lives on (library.py:361-364) - a no-op for
// every grid the reference can produce - so that the main loop's LDS index needs no clamp of its
// own whatever the caller passed (fmax/fmin also turn a NaN into 0).
__global__ void grid_pairs_kernel(const double* __restrict__ mult, long long n, double2* __restrict__ pairs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + PRHF_PAIR_PAD) return;
    if (i >= n) {                                  // padding: the main loop's prefetch may read it, nothing uses it
        pairs[i] = make_double2(0.0, 0.0);
        return;
    }
    const double m = mult[i];
    pairs[i] = make_double2(fmin(fmax(m, 0.0), 1.0), (i + 1 < n ? mult[i + 1] : m) - m);
}

// pairs must hold n + PRHF_PAIR_PAD entries
hipError_t launch_grid_pairs(const double* mult, long long n, double* pairs, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(grid_pairs_kernel, dim3((unsigned)((n + PRHF_PAIR_PAD + 255) / 256)), dim3(256), 0, stream, mult,
                       n, reinterpret_cast<double2*>(pairs));
    return hipGetLastError();
}

// Per-frequency scalars of a launch (PairFreq): row f = {f_hz, f2, cX, cY2, 1/f2, 1/f_hz, 0, 0}.
__global__ __launch_bounds__(256) void freq_table_kernel(const double* __restrict__ freq_mhz, long long n_freq,
                                                         double* __restrict__ tab, const ZeroWords zero) {
#pragma clang fp contract(off)
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_freq) {
        // A frequency that is not a positive finite number is carried as NaN: every comparison of the level search then
        // fails and the column comes out NaN (the reference: NaN for 0 and NaN, something meaningless for f < 0).
        const double fm = freq_mhz[f];
        const double f_hz = (fm > 0.0 && fm < __builtin_inf()) ? fm * 1e6 : qnan();       // :491
        const double f2 = f_hz * f_hz;                             // f**2
        const double cY = kGyro / f_hz;
        double* row = tab + 8 * f;
        row[0] = f_hz; row[1] = f2; row[2] = (kPlasma * kPlasma) / f2; row[3] = cY * cY;
        row[4] = 1.0 / f2; row[5] = 1.0 / f_hz; row[6] = 0.0; row[7] = 0.0;
    }
    if (blockIdx.x != 0) return;
    // Workgroup 0 alone: the control words that the launches behind this one count in - block queues, the heads of the
    // short-grid kernels' lists, the classes of the block order - are zeroed here instead of by one memset each (a
    // short-grid launch had six of them in front of it: 25 us of a 530 us launch) ...
    for (int k = 0; k < zero.n; ++k)
        for (int i = threadIdx.x; i < zero.words[k]; i += blockDim.x) zero.p[k][i] = 0u;
    // ... and row n_freq of the table: min |freq_mhz| over the launch (the isotropic test of library.py:201 needs the
    // lowest frequency; the short-grid kernels read it here instead of scanning the frequencies once per profile) and,
    // in its second word, the largest finite |freq_mhz| (short_order_kernel's scale).
    __shared__ double part[4];
    __shared__ double part_max[4];
    double fm = __builtin_inf(), fx = 0.0;
    for (long long i = threadIdx.x; i < n_freq; i += blockDim.x) {
        const double v = fabs(freq_mhz[i]);
        fm = fmin(fm, v);
        if (v < __builtin_inf()) fx = fmax(fx, v);             // (the largest finite one: the block-ordering proxy's scale)
    }
    fm = wave_min(fm);
    fx = wave_max(fx);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = fm; part_max[threadIdx.x >> 6] = fx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* row = tab + 8 * n_freq;
        row[0] = fmin(fmin(part[0], part[1]), fmin(part[2], part[3]));
        row[1] = fmax(fmax(part_max[0], part_max[1]), fmax(part_max[2], part_max[3]));
        for (int k = 2; k < 8; ++k) row[k] = 0.0;
    }
}

// One wavefront per profile: first-occurrence argmax of the density column, a NaN ranking above every number
// (stage_profile's phase 1, library.py:371); the launch's maximum goes to *max_peak.
__global__ __launch_bounds__(256) void peak_levels_kernel(const double* __restrict__ den, long long n_prof, long long n_alt,
                                                          long long prof_stride, unsigned* __restrict__ max_peak) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= n_prof) return;
    const double* d = den + p * prof_stride;
    double bv = -__builtin_inf();
    int bi = 0x7fffffff;
    for (int i = lane; i < (int)n_alt; i += 64) {
        const double v = d[i];
        const double key = (v != v) ? __builtin_inf() : v;
        if (key > bv) { bv = key; bi = i; }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0 && bi != 0x7fffffff) atomicMax(max_peak, (unsigned)bi);
}

hipError_t launch_peak_levels(const double* den, long long n_prof, long long n_alt, long long prof_stride,
                              unsigned* max_peak, hipStream_t stream) {
    if (n_prof <= 0) return hipSuccess;
    hipLaunchKernelGGL(peak_levels_kernel, dim3((unsigned)((n_prof + 3) / 4)), dim3(256), 0, stream, den, n_prof, n_alt,
                       prof_stride, max_peak);
    return hipGetLastError();
}

hipError_t launch_freq_table(const double* freq_mhz, long long n_freq, double* tab, const ZeroWords& zero, hipStream_t stream) {
    if (n_freq <= 0) return hipSuccess;
    hipLaunchKernelGGL(freq_table_kernel, dim3((unsigned)((n_freq + 255) / 256)), dim3(256), 0, stream, freq_mhz, n_freq,
                       tab, zero);
    return hipGetLastError();
}

// Chunked pairs: add the chunk sums in a fixed order, then the reference's 0 -> NaN and + min(alt).
__global__ void vfo_finalize_kernel(const KArgs a, int s) {
    const SegDev& sg = a.seg[s];
    const long long n_pairs = (sg.prof_end - sg.prof_begin) * a.n_freq;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs) return;
    const double* part = a.partial + sg.partial_off + t * sg.chunks;
    double sum = 0.0;
    for (int c = 0; c < sg.chunks; ++c) sum = sum + part[c];
    const double amin = a.altmin[sg.altmin_off + t / a.n_freq];
    a.out[sg.out_off + t] = (sum == sum && sum != 0.0) ? sum + amin : qnan();
}

// Standalone Appleton-Hartree indices on flat arrays (find_mu_mup, library.py:161-256).
// The reference switches to the isotropic formulas when nanmax|Y| < 1e-12 over the WHOLE array (:201).  That is
// decided here in the same pass: the magnetised formulas are evaluated while max|Y| is collected (`track`: two
// words that start at 0 - [0] max |Y| ignoring NaN, as a bit pattern, which orders like the double for non-negative
// values; [1] != 0 when any element was not NaN), and only if the array turns out to be isotropic - it almost
// never is - the launcher runs the pass again with `unmag` set.  (A separate pre-pass read Y a second time: 48
// instead of 40 bytes moved per element of an HBM-bound op.)
template <int TIER>
__global__ void mu_mup_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                              const double* __restrict__ psi, long long n, int mode, int unmag,
                              double* __restrict__ mu_out, double* __restrict__ mup_out,
                              unsigned long long* track) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double ymax = 0.0;
    int seen = 0;
    auto one = [&](double x, double y, double p, double* mu, double* mup) {
        const double ay = fabs(y);
        ymax = fmax(ymax, ay);
        seen |= (ay == ay) ? 1 : 0;
        if (unmag) {
            index_unmagnetised(x, mu, mup);
        } else if (TIER == 0) {
            if (mode == PRHF_KMODE_O) index_faithful<PRHF_KMODE_O>(x, y, p, mu, mup);
            else index_faithful<PRHF_KMODE_X>(x, y, p, mu, mup);
        } else {
            const double s2 = sin_sq_deg(p);
            if (mode == PRHF_KMODE_O) index_fast<PRHF_KMODE_O>(x, y * y, s2, mu, mup);
            else index_fast<PRHF_KMODE_X>(x, y * y, s2, mu, mup);
        }
    };
    // two elements per thread and pass where the five arrays allow 16-byte accesses (wider loads in flight, half
    // the address arithmetic); results are written once and never read back here: streaming stores
    const bool wide = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(psi) |
                        reinterpret_cast<uintptr_t>(mu_out) | reinterpret_cast<uintptr_t>(mup_out)) & 15) == 0;
    const long long n2 = wide ? n >> 1 : 0;
    typedef double vec2 __attribute__((ext_vector_type(2)));
    auto pair_of = [&](long long i, vec2 x, vec2 y, vec2 ps) {
        double mu0, mup0, mu1, mup1;
        one(x.x, y.x, ps.x, &mu0, &mup0);
        one(x.y, y.y, ps.y, &mu1, &mup1);
        const vec2 m = {mu0, mu1}, mp = {mup0, mup1};
        __builtin_nontemporal_store(m, reinterpret_cast<vec2*>(mu_out) + i);
        __builtin_nontemporal_store(mp, reinterpret_cast<vec2*>(mup_out) + i);
    };
    // (plain loads, streaming stores: non-temporal loads measured no better; the loads of two trips are issued
    //  together - 96 bytes per lane in flight - before the arithmetic of the first)
    // (every workgroup streams through a contiguous piece of the arrays - not a grid-wide stride, which has 4096
    //  workgroups x 5 arrays open as many DRAM pages at once)
    const long long piece = (n2 + gridDim.x - 1) / gridDim.x;
    const long long p_end = (blockIdx.x + 1) * piece < n2 ? (blockIdx.x + 1) * piece : n2;
    const long long step = blockDim.x;
    long long i = blockIdx.x * piece + threadIdx.x;
    for (; i + step < p_end; i += 2 * step) {
        const long long k = i + step;
        const vec2 x0 = reinterpret_cast<const vec2*>(X)[i], y0 = reinterpret_cast<const vec2*>(Y)[i];
        const vec2 p0 = reinterpret_cast<const vec2*>(psi)[i];
        const vec2 x1 = reinterpret_cast<const vec2*>(X)[k], y1 = reinterpret_cast<const vec2*>(Y)[k];
        const vec2 p1 = reinterpret_cast<const vec2*>(psi)[k];
        pair_of(i, x0, y0, p0);
        pair_of(k, x1, y1, p1);
    }
    for (; i < p_end; i += step) {
        const vec2 x = reinterpret_cast<const vec2*>(X)[i], y = reinterpret_cast<const vec2*>(Y)[i];
        const vec2 ps = reinterpret_cast<const vec2*>(psi)[i];
        pair_of(i, x, y, ps);
    }
    for (long long i = 2 * n2 + tid; i < n; i += stride) {
        double mu, mup;
        one(X[i], Y[i], psi[i], &mu, &mup);
        mu_out[i] = mu;
        mup_out[i] = mup;
    }
    if (track) {
        // one pair of atomics per workgroup: they all hit the same two words, ~12 ns each, one after the other
        __shared__ double wg_max[4];
        __shared__ int wg_seen[4];
        ymax = wave_max(ymax);
        seen = __any(seen) ? 1 : 0;
        if ((threadIdx.x & 63) == 0) { wg_max[threadIdx.x >> 6] = ymax; wg_seen[threadIdx.x >> 6] = seen; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int waves = (int)(blockDim.x >> 6);
            for (int w = 1; w < waves; ++w) { ymax = fmax(ymax, wg_max[w]); seen |= wg_seen[w]; }
            atomicMax(track, (unsigned long long)__double_as_longlong(ymax));
            if (seen) atomicOr(track + 1, 1ull);
        }
    }
}

hipError_t launch_vfo(const KArgs& a, long long n_blocks, int tier, size_t lds_bytes, hipStream_t stream) {
    constexpr int THREADS = PRHF_BLOCK_THREADS;
    if (n_blocks <= 0 || a.n_blocks <= 0) return hipSuccess;
    if (tier == 0)
        hipLaunchKernelGGL((vfo_kernel<0, THREADS>), dim3((unsigned)n_blocks), dim3(THREADS), lds_bytes, stream, a);
    else if (tier == 1)
        hipLaunchKernelGGL((vfo_kernel<1, THREADS>), dim3((unsigned)n_blocks), dim3(THREADS), lds_bytes, stream, a);
    else if (tier == 2)
        hipLaunchKernelGGL((vfo_kernel<2, THREADS>), dim3((unsigned)n_blocks), dim3(THREADS), lds_bytes, stream, a);
    else
        hipLaunchKernelGGL((vfo_tall_kernel<THREADS>), dim3((unsigned)n_blocks), dim3(THREADS), lds_bytes, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (int s = 0; s < a.n_segs; ++s) {
        if (a.seg[s].chunks <= 1 || a.seg[s].slots > 0) continue;     // (block-local chunks are added up by their workgroup)
        const long long n_pairs = (a.seg[s].prof_end - a.seg[s].prof_begin) * a.n_freq;
        if (n_pairs <= 0) continue;
        const unsigned blocks = (unsigned)((n_pairs + 255) / 256);
        hipLaunchKernelGGL(vfo_finalize_kernel, dim3(blocks), dim3(256), 0, stream, a, s);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_mu_mup(const double* X, const double* Y, const double* psi, long long n, int mode, int tier,
                         unsigned long long* absmax_scratch, unsigned long long* absmax_host,
                         double* mu, double* mup, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipError_t e = hipMemsetAsync(absmax_scratch, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    for (int unmag = 0; unmag < 2; ++unmag) {
        unsigned long long* track = unmag ? nullptr : absmax_scratch;
        if (tier == 0)
            hipLaunchKernelGGL(mu_mup_kernel<0>, dim3(blocks), dim3(256), 0, stream, X, Y, psi, n, mode, unmag, mu, mup, track);
        else
            hipLaunchKernelGGL(mu_mup_kernel<1>, dim3(blocks), dim3(256), 0, stream, X, Y, psi, n, mode, unmag, mu, mup, track);
        e = hipGetLastError();
        if (e != hipSuccess || unmag) return e;
        e = hipMemcpyAsync(absmax_host, absmax_scratch, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        double ymax;
        __builtin_memcpy(&ymax, absmax_host, sizeof ymax);
        // all-NaN Y: np.nanmax gives NaN and the comparison is false -> magnetised formulas (already written)
        if (!(absmax_host[1] != 0 && ymax < kUnmagTol)) return hipSuccess;
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------
// Standalone find_vh (library.py:259-293): one wavefront per row of (n_rows, n_cols) arrays,
// lanes stride the columns (coalesced), NaN terms skipped, 0 -> NaN, + alt_min.
// ---------------------------------------------------------------------------------------
template <int TIER>
__global__ void find_vh_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                               const double* __restrict__ psi, const double* __restrict__ dh, long long n_rows,
                               long long n_cols, double alt_min, int mode, int unmag, double* __restrict__ vh,
                               unsigned long long* track) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const long long base = row * n_cols;
    double acc = 0.0, ymax = 0.0;
    int seen = 0;
    auto one = [&](double x, double y, double p, double thickness) {
        double mu, mup;
        const double ay = fabs(y);
        ymax = fmax(ymax, ay);                                 // the isotropic test of :201, same pass (mu_mup_kernel)
        seen |= (ay == ay) ? 1 : 0;
        if (unmag) {
            index_unmagnetised(x, &mu, &mup);
        } else if (TIER == 0) {
            if (mode == PRHF_KMODE_O) index_faithful<PRHF_KMODE_O>(x, y, p, &mu, &mup);
            else index_faithful<PRHF_KMODE_X>(x, y, p, &mu, &mup);
        } else {
            const double s2 = sin_sq_deg(p);
            if (mode == PRHF_KMODE_O) index_fast<PRHF_KMODE_O>(x, y * y, s2, &mu, &mup);
            else index_fast<PRHF_KMODE_X>(x, y * y, s2, &mu, &mup);
        }
        const double term = mup * thickness;                   // :288
        if (term == term) acc = acc + term;
    };
    // two columns per lane and trip where the row starts on a 16-byte boundary in all four arrays
    const bool wide = (n_cols & 1) == 0 &&
        ((reinterpret_cast<uintptr_t>(X + base) | reinterpret_cast<uintptr_t>(Y + base) |
          reinterpret_cast<uintptr_t>(psi + base) | reinterpret_cast<uintptr_t>(dh + base)) & 15) == 0;
    if (wide) {
        typedef double vec2 __attribute__((ext_vector_type(2)));
        const vec2* X2 = reinterpret_cast<const vec2*>(X + base);
        const vec2* Y2 = reinterpret_cast<const vec2*>(Y + base);
        const vec2* P2 = reinterpret_cast<const vec2*>(psi + base);
        const vec2* D2 = reinterpret_cast<const vec2*>(dh + base);
        for (long long i = lane; i < (n_cols >> 1); i += 64) {
            const vec2 x = X2[i], y = Y2[i], p = P2[i], d = D2[i];
            one(x.x, y.x, p.x, d.x);
            one(x.y, y.y, p.y, d.y);
        }
    } else {
        for (long long i = lane; i < n_cols; i += 64) one(X[base + i], Y[base + i], psi[base + i], dh[base + i]);
    }
    acc = wave_sum(acc);
    if (lane == 0) vh[row] = (acc != 0.0) ? acc + alt_min : qnan();     // :290-292
    if (track) {
        ymax = wave_max(ymax);
        seen = __any(seen) ? 1 : 0;
        if (lane == 0) {
            atomicMax(track, (unsigned long long)__double_as_longlong(ymax));
            if (seen) atomicOr(track + 1, 1ull);
        }
    }
}

hipError_t launch_find_vh(const double* X, const double* Y, const double* psi, const double* dh, long long n_rows,
                          long long n_cols, double alt_min, int mode, int tier,
                          unsigned long long* absmax_scratch, unsigned long long* absmax_host, double* vh,
                          hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(absmax_scratch, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((n_rows + 3) / 4);
    for (int unmag = 0; unmag < 2; ++unmag) {
        unsigned long long* track = unmag ? nullptr : absmax_scratch;
        if (tier == 0)
            hipLaunchKernelGGL(find_vh_kernel<0>, dim3(blocks), dim3(256), 0, stream, X, Y, psi, dh, n_rows, n_cols,
                               alt_min, mode, unmag, vh, track);
        else
            hipLaunchKernelGGL(find_vh_kernel<1>, dim3(blocks), dim3(256), 0, stream, X, Y, psi, dh, n_rows, n_cols,
                               alt_min, mode, unmag, vh, track);
        e = hipGetLastError();
        if (e != hipSuccess || unmag) return e;
        e = hipMemcpyAsync(absmax_host, absmax_scratch, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        double ymax;
        __builtin_memcpy(&ymax, absmax_host, sizeof ymax);
        if (!(absmax_host[1] != 0 && ymax < kUnmagTol)) return hipSuccess;      // magnetised: done (:201)
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------
// Standalone regrid_to_nonuniform_grid (library.py:324-438) for ONE profile: one workgroup per
// frequency stages the profile, wave 0 finds the reflection height, all waves write the
// (n_points) rows of the seven float64 outputs and the int64 index row.  IEEE arithmetic in the
// reference's order: the outputs are bit-identical to NumPy's.
// ---------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void regrid_kernel(const RegridArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_alt = (int)a.n_alt;
    Node* nodes = reinterpret_cast<Node*>(smem);
    double* pf2 = reinterpret_cast<double*>(smem + (size_t)(n_alt + 1) * sizeof(Node));
    double* gb = pf2 + n_alt;
    unsigned short* hint = reinterpret_cast<unsigned short*>(gb + n_alt);
    double* red = reinterpret_cast<double*>(hint + kHintBuckets + PRHF_MAX_CAND);
    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const double one_mhz = 1.0;     // stage_profile only needs a frequency column for the isotropic test
    const BlockInfo info = stage_profile<0, THREADS>(a.den, a.bmag, a.bpsi, a.alt, &one_mhz, 0, n_alt, nodes,
                                                     pf2, gb, hint, red, n_alt + 1);
    // (the standalone regrid keeps refusing NaN profiles: the operator's NaN rules - stage_profile - are about its sums)
    if (threadIdx.x == 0 && (info.bad || info.nan_b || info.nan_p))
        post_status(a.status, (unsigned)((info.bad & ~kNanRow) | ((info.bad & kNanRow) || info.nan_b || info.nan_p
                                                                   ? PRHF_STATUS_NANINPUT : 0)));
    if (a.mode == PRHF_KMODE_O && !info.bad) prefix_max_in_place<THREADS>(pf2, info.K, red);
    const double f_hz = a.freq_hz[f];
    double h = qnan();
    int reflects = 0;
    if (!info.bad) {
        double hh;
        const double f2 = f_hz * f_hz;
        const bool ok = (a.mode == PRHF_KMODE_O)
            ? reflection_height<PRHF_KMODE_O>(nodes, pf2, gb, info.K, f_hz, f2, lane, &hh)
            : reflection_height<PRHF_KMODE_X>(nodes, pf2, gb, info.K, f_hz, f2, lane, &hh);
        if (ok) { h = hh; reflects = 1; }
    }
    const int K = info.K;
    const double a0 = info.bad ? 0.0 : info.a0;
    const double span = h - a0;
    const long long row = (long long)f * a.n_points;
    for (int i = threadIdx.x; i < a.n_points; i += THREADS) {
        const double z = a.mult[i] * span + a0;                              // :413 (NaN when escaping)
        double dh = kBackoff;                                                // :415-416
        if (i + 1 < a.n_points) dh = (a.mult[i + 1] * span + a0) - z;
        double d = z, b = z, p = z;                                          // np.interp(NaN) = NaN
        if (!info.bad && (reflects || K == 1)) {
            int j = 0;
            if (reflects) {
                j = guess_segment(hint, info, z);
                while (j > 0 && z < nodes[j].alt) --j;
                while (j + 1 < K && z >= nodes[j + 1].alt) ++j;
            }
            const Node nd = nodes[j];
            double dz = reflects ? z - nd.alt : 0.0;                         // K == 1: the single node
            if (dz < 0.0) dz = 0.0;
            d = nd.sden * dz + nd.den;                                       // :424-426
            b = nd.sb * dz + nd.b;
            p = nd.spsi * dz + nd.psi;
        }
        // (written once, never read back here: streaming stores)
        __builtin_nontemporal_store(f_hz, a.out_freq + row + i);
        __builtin_nontemporal_store(d, a.out_den + row + i);
        __builtin_nontemporal_store(b, a.out_bmag + row + i);
        __builtin_nontemporal_store(p, a.out_bpsi + row + i);
        __builtin_nontemporal_store(dh, a.out_dist + row + i);
        __builtin_nontemporal_store(z, a.out_alt + row + i);
        __builtin_nontemporal_store(h, a.out_crit + row + i);
        __builtin_nontemporal_store((long long)i, a.out_ind + row + i);
    }
}

hipError_t launch_vfo_tall(const KArgs& a, long long grid_blocks, hipStream_t stream) {
    if (!a.tall || a.tall_stride < tall_slab_bytes(a.n_alt)) return hipErrorInvalidValue;
    return launch_vfo(a, grid_blocks, 3, lds_bytes_tall(), stream);
}

hipError_t launch_regrid(const RegridArgs& a, size_t lds_bytes, hipStream_t stream) {
    if (a.n_freq <= 0 || a.n_points <= 0) return hipSuccess;
    hipLaunchKernelGGL(regrid_kernel<512>, dim3((unsigned)a.n_freq), dim3(512), lds_bytes, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Residual rows of the fitting driver (residual_VH, library.py:660-669): per candidate profile,
// modeled NaNs are replaced by max(nanmean|vh_model|, 100) and residual = vh_obs - vh_model;
// cost = sum of squares (what lmfit's brute-force grid search minimises, library.py:794-798).
// One wavefront per candidate row.
// ---------------------------------------------------------------------------------------
__global__ void residual_kernel(const double* __restrict__ vh_model, const double* __restrict__ vh_obs,
                                long long n_prof, int n_freq, double* __restrict__ residual,
                                double* __restrict__ cost) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_prof) return;
    const double* v = vh_model + row * n_freq;
    double sum = 0.0, cnt = 0.0;
    for (int f = lane; f < n_freq; f += 64) {
        const double a = fabs(v[f]);
        if (a == a) { sum += a; cnt += 1.0; }
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    // np.maximum(np.nanmean(|v|), 100): an all-NaN row has NaN mean and np.maximum propagates it (:664)
    const double mean = (cnt > 0.0) ? sum / cnt : qnan();
    const double fill = (mean == mean) ? fmax(mean, 100.0) : qnan();
    double c = 0.0;
    for (int f = lane; f < n_freq; f += 64) {
        double m = v[f];
        if (!(m == m)) m = fill;
        const double r = vh_obs[f] - m;                        // :668
        if (residual) residual[row * n_freq + f] = r;
        c += r * r;
    }
    c = wave_sum(c);
    if (lane == 0 && cost) cost[row] = c;
}

hipError_t launch_residual(const double* vh_model, const double* vh_obs, long long n_prof, int n_freq,
                           double* residual, double* cost, hipStream_t stream) {
    if (n_prof <= 0) return hipSuccess;
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((n_prof + 3) / 4)), dim3(256), 0, stream, vh_model, vh_obs,
                       n_prof, n_freq, residual, cost);
    return hipGetLastError();
}

#include "prhf_short.inc"

static const void* short_kernel_for(int threads, int lanes) {
    if (threads == PRHF_COMPACT_THREADS)
        return lanes == 8 ? reinterpret_cast<const void*>(&vfo_short_kernel<PRHF_COMPACT_THREADS, 8>)
                          : reinterpret_cast<const void*>(&vfo_short_kernel<PRHF_COMPACT_THREADS, 16>);
    if (threads == PRHF_SHORT_THREADS)
        return lanes == 8 ? reinterpret_cast<const void*>(&vfo_short_kernel<PRHF_SHORT_THREADS, 8>)
                          : reinterpret_cast<const void*>(&vfo_short_kernel<PRHF_SHORT_THREADS, 16>);
    return nullptr;
}

// lanes: 16 or 8 lanes per pair (vfo_short_kernel's LP)
hipError_t launch_vfo_short(const KArgs& a, long long grid_blocks, size_t lds_bytes, int threads, int lanes, hipStream_t stream) {
    if (grid_blocks <= 0 || a.n_blocks <= 0) return hipSuccess;
    const void* kernel = short_kernel_for(threads, lanes);
    if (!kernel || (lanes != 8 && lanes != 16)) return hipErrorInvalidValue;
    void* params[] = {const_cast<KArgs*>(&a)};
    return hipLaunchKernel(kernel, dim3((unsigned)grid_blocks), dim3((unsigned)threads), params, lds_bytes, stream);
}

hipError_t launch_short_order(const KArgs& a, unsigned* order, const double* freq_mhz, double* tab, const ZeroWords& zero,
                              hipStream_t stream) {
    if (a.n_blocks <= 0) return hipErrorInvalidValue;        // (the table must be made: the caller takes launch_freq_table then)
    hipLaunchKernelGGL(short_order_kernel, dim3((unsigned)((a.n_blocks + 63) / 64)), dim3(1024), 0, stream, a, order, freq_mhz,
                       tab, zero);
    return hipGetLastError();
}

hipError_t launch_vfo_shortx(const KArgs& a, long long grid_blocks, size_t lds_bytes, int threads, hipStream_t stream) {
    if (grid_blocks <= 0 || a.n_blocks <= 0) return hipSuccess;
    if (threads == PRHF_COMPACT_THREADS)
        hipLaunchKernelGGL((vfo_shortx_kernel<PRHF_COMPACT_THREADS>), dim3((unsigned)grid_blocks), dim3(PRHF_COMPACT_THREADS),
                           lds_bytes, stream, a);
    else if (threads == PRHF_SHORT_THREADS)
        hipLaunchKernelGGL((vfo_shortx_kernel<PRHF_SHORT_THREADS>), dim3((unsigned)grid_blocks), dim3(PRHF_SHORT_THREADS),
                           lds_bytes, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

#include "prhf_snell.inc"

// Resident workgroups per CU the runtime predicts for the fused kernel (diagnostics).
hipError_t query_occupancy(int tier, size_t lds_bytes, int* blocks_per_cu) {
    constexpr int THREADS = PRHF_BLOCK_THREADS;
    if (tier == 0)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, vfo_kernel<0, THREADS>, THREADS, lds_bytes);
    if (tier == 1)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, vfo_kernel<1, THREADS>, THREADS, lds_bytes);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, vfo_kernel<2, THREADS>, THREADS, lds_bytes);
}

hipError_t configure_kernels(size_t max_lds_bytes) {
    constexpr int THREADS = PRHF_BLOCK_THREADS;
    const void* kernels[] = {reinterpret_cast<const void*>(&vfo_kernel<0, THREADS>),
                             reinterpret_cast<const void*>(&vfo_kernel<1, THREADS>),
                             reinterpret_cast<const void*>(&vfo_kernel<2, THREADS>),
                             short_kernel_for(PRHF_SHORT_THREADS, 16), short_kernel_for(PRHF_SHORT_THREADS, 8),
                             short_kernel_for(PRHF_COMPACT_THREADS, 16), short_kernel_for(PRHF_COMPACT_THREADS, 8),
                             reinterpret_cast<const void*>(&vfo_shortx_kernel<PRHF_SHORT_THREADS>),
                             reinterpret_cast<const void*>(&vfo_shortx_kernel<PRHF_COMPACT_THREADS>),
                             reinterpret_cast<const void*>(&regrid_kernel<512>)};
    for (const void* k : kernels) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace prhf

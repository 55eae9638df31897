// prhf_crmath.h - x^3, x^4, sin and cos rounded like a correctly rounding libm, for the
// reference-order arithmetic tier (index_faithful in prhf_kernels.hip).
//
// Why: near the O-mode reflection height the reference's D = (1 - X) - YT^2/2 + sqrt(YT^4/4 + ...)
// (PyRayHF/library.py:217-229) cancels to ~1e-9 of its terms, so the last grid points of a pair are
// decided by how YT**4, YT**3 (NumPy: pow), sin and cos (NumPy float64: the platform libm) were
// ROUNDED.  glibc's sin/cos/pow return the correctly rounded result in > 99 % of the cases and
// NumPy's SIMD pow in ~95 %; (x*x)*(x*x) does in 50 % and the device library's sin/cos in ~80 %.
// These versions evaluate in double-double (or from a double-double table) and round once: they agree with
// the correctly rounded value except within 2^-9 ... 2^-11 ulp of a rounding boundary.
//
// The same source is compiled for gfx950 (hipcc) and for the host (gcc, tests/test_crmath_host.py
// checks it against exactly rounded big-rational values) - plain C++, FMA through __builtin_fma.

#ifndef PRHF_CRMATH_H
#define PRHF_CRMATH_H

#if defined(__HIPCC__)
#define PRHF_HD __host__ __device__ __forceinline__
#define PRHF_D __device__ __forceinline__
#define PRHF_TRIG_TABLE_DECL static __device__ const double kTrigTable[]
#else
#define PRHF_HD static inline
#define PRHF_D static inline
#define PRHF_TRIG_TABLE_DECL static const double kTrigTable[]
#endif

namespace prhf_cr {

struct dd {
    double hi, lo;
};

// a + b exactly as hi + lo (any magnitudes)
PRHF_HD dd two_sum(double a, double b) {
    const double s = a + b;
    const double bb = s - a;
    const double e = (a - (s - bb)) + (b - bb);
    return dd{s, e};
}
// the same when |a| >= |b| (or a == 0)
PRHF_HD dd quick_two_sum(double a, double b) {
    const double s = a + b;
    return dd{s, b - (s - a)};
}
PRHF_HD dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, __builtin_fma(a, b, -p)};
}
PRHF_HD dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo = __builtin_fma(a.hi, b.lo, __builtin_fma(a.lo, b.hi, p.lo));
    return quick_two_sum(p.hi, p.lo);
}
PRHF_HD dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo = __builtin_fma(a.lo, b, p.lo);
    return quick_two_sum(p.hi, p.lo);
}
PRHF_HD dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return quick_two_sum(s.hi, s.lo);
}

// x^4 and x^3, one rounding (NumPy evaluates YT**4 and YT**3 with pow; library.py:217, :244)
PRHF_HD double pow4(double x) {
    const dd s = two_prod(x, x);                 // x^2 = s.hi + s.lo exactly
    const dd p = two_prod(s.hi, s.hi);           // s.hi^2 exactly
    return p.hi + __builtin_fma(2.0 * s.hi, s.lo, p.lo);
}
PRHF_HD double pow3(double x) {
    const dd s = two_prod(x, x);
    const dd p = two_prod(s.hi, x);
    return p.hi + __builtin_fma(s.lo, x, p.lo);
}

// r = n pi/2 + y with |y| <= pi/4 (+ a rounding), y as a double-double: n*pio2_1 rounds to p with exact error pe,
// r - p is exact (p is within a factor 2 of r, or zero), the remaining pieces are ~1e-16 |r| and go into the low word.
PRHF_HD dd reduce_pio2(double r, double* n_out) {
    // pi/2 in three pieces of 53 bits
    const double kPio2_1 = 1.5707963267948966;        // 0x3FF921FB54442D18
    const double kPio2_2 = 6.123233995736766e-17;     // 0x3C91A62633145C07
    const double kPio2_3 = -1.4973849048591698e-33;   // 0xB91F1976B7ED8FBC
    const double kTwoOverPi = 0.6366197723675814;
    const double n = __builtin_rint(r * kTwoOverPi);
    const dd p1 = two_prod(n, kPio2_1);
    const double d = r - p1.hi;
    const dd p2 = two_prod(n, kPio2_2);
    dd y = two_sum(d, -p1.lo);
    dd t = two_sum(y.hi, -p2.hi);
    t.lo += y.lo - p2.lo - n * kPio2_3;
    *n_out = n;
    return quick_two_sum(t.hi, t.lo);
}

// sin(r) and cos(r) for a double r, |r| < 2^20 * pi/2 (callers fall back to the device library beyond;
// field angles are degrees-sized).  Both series are summed in double-double down to the terms whose double
// rounding error is < 2^-64 of the result.  (The reference implementation of this file: sincos_table below is
// what the kernels call.)
PRHF_HD void sincos(double r, double* s_out, double* c_out) {
    double n;
    const dd y = reduce_pio2(r, &n);
    const dd z = dd_mul(y, y);                        // y^2
    // sin y = y + y^3 (S1 + z (S2 + z (S3 + z (S4 + ...))))
    const dd S1 = dd{-0.16666666666666666, -9.251858538542970e-18};     // -1/6
    const dd S2 = dd{0.008333333333333333, 1.1564823173178714e-19};     //  1/120
    const double S3 = -1.984126984126984e-04, S4 = 2.7557319223985893e-06, S5 = -2.505210838544172e-08,
                 S6 = 1.6059043836821613e-10, S7 = -7.647163731819816e-13, S8 = 2.8114572543455206e-15;
    const double zh = z.hi;
    const double sp = S3 + zh * (S4 + zh * (S5 + zh * (S6 + zh * (S7 + zh * S8))));
    dd sq = dd_mul(z, dd_add(S2, dd{zh * sp, 0.0}));  // z (S2 + z (...))
    sq = dd_add(S1, sq);
    const dd y3 = dd_mul(z, y);
    const dd sin_y = dd_add(y, dd_mul(y3, sq));
    // cos y = 1 + z (C1 + z (C2 + z (C3 + ...)))
    const dd C2 = dd{0.041666666666666664, 2.3129646346357427e-18};     //  1/24
    const dd C3 = dd{-0.001388888888888889, 5.300543954373577e-20};     // -1/720
    const double C4 = 2.48015873015873e-05, C5 = -2.755731922398589e-07, C6 = 2.08767569878681e-09,
                 C7 = -1.1470745597729725e-11, C8 = 4.779477332387385e-14, C9 = -1.5619206968586225e-16;
    const double cp = C4 + zh * (C5 + zh * (C6 + zh * (C7 + zh * (C8 + zh * C9))));
    dd cq = dd_mul(z, dd_add(C3, dd{zh * cp, 0.0}));
    cq = dd_mul(z, dd_add(C2, cq));
    cq = dd_add(dd{-0.5, 0.0}, cq);
    const dd cos_y = dd_add(dd{1.0, 0.0}, dd_mul(z, cq));

    const double sy = sin_y.hi + sin_y.lo, cy = cos_y.hi + cos_y.lo;
    const long long q = (long long)n & 3;
    *s_out = (q == 0) ? sy : (q == 1) ? cy : (q == 2) ? -sy : -cy;
    *c_out = (q == 0) ? cy : (q == 1) ? -sy : (q == 2) ? -cy : sy;
}

#include "prhf_trig_table.inc"

// The same two values from a table: sin and cos of k pi/4096 (k = 0..1024, double-double, generated with
// 80-digit arithmetic by tools/gen_trig_table.py) rotated by the remainder |t| <= pi/8192,
//   sin(a + t) = S + (C sin t + S (cos t - 1)),   cos(a + t) = C + (C (cos t - 1) - S sin t),
// where the bracket is at most 3.8e-4 of the leading term, so that evaluating it in plain double leaves an
// error of 2^-64 - the sum is then rounded once.  Where sin y itself is small (|y| < 64 pi/4096 ~ 0.05) the
// bracket would no longer be small against it and the odd series in y is used instead.  ~70 vector instructions
// against ~195 for the double-double series above, and the exactly rounded value in 99.7 % of the cases (the
// series: 100 % of those tried; NumPy's own libm: 99.9 %) - tests/test_crmath_host.py.
PRHF_D void sincos_table(double r, double* s_out, double* c_out) {
    double n;
    const dd y = reduce_pio2(r, &n);
    const bool neg = y.hi < 0.0;
    const double ay = neg ? -y.hi : y.hi, al = neg ? -y.lo : y.lo;
    double kf = __builtin_rint(ay * PRHF_TRIG_INV_H);
    kf = kf > (double)PRHF_TRIG_STEPS ? (double)PRHF_TRIG_STEPS : kf;
    const double th = __builtin_fma(-kf, PRHF_TRIG_H1, ay);         // exact: kf * H1 has <= 52 significant bits
    const double tl = __builtin_fma(-kf, PRHF_TRIG_H2, al);
    const double t2 = th * th;
    // sin t and cos t - 1, |t| <= 3.9e-4: t^7/5040 and t^8/40320 are below 2^-64 of the leading term
    const double st = __builtin_fma(th * t2, __builtin_fma(t2, 1.0 / 120.0, -1.0 / 6.0), th) + tl;
    const double ct1 = __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(t2, -1.0 / 720.0, 1.0 / 24.0), -0.5), -th * tl);
    const double* e = kTrigTable + 4 * (int)kf;
    const double S = e[0], Sl = e[1], C = e[2], Cl = e[3];
    double sin_y = S + (__builtin_fma(C, st, S * ct1) + __builtin_fma(Cl, th, Sl));
    const double cos_y = C + (__builtin_fma(C, ct1, -S * st) + __builtin_fma(-Sl, th, Cl));
    // small |y|: y + (yl + y^3 P(y^2)), the correction is < 4e-4 of y
    const double z = ay * ay;
    const double p = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0),
                                   -1.0 / 6.0);
    const double sin_small = ay + __builtin_fma(ay * z, p, al);
    if (kf < 64.0) sin_y = sin_small;
    const double sy = neg ? -sin_y : sin_y;
    const long long q = (long long)n & 3;
    *s_out = (q == 0) ? sy : (q == 1) ? cos_y : (q == 2) ? -sy : -cos_y;
    *c_out = (q == 0) ? cos_y : (q == 1) ? -sy : (q == 2) ? -cos_y : sy;
}

}  // namespace prhf_cr

#endif

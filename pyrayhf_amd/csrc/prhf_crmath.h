// prhf_crmath.h - x^3, x^4, sin and cos rounded like a correctly rounding libm, for the
// reference-order arithmetic tier (index_faithful in prhf_kernels.hip).
//
// Why: near the O-mode reflection height the reference's D = (1 - X) - YT^2/2 + sqrt(YT^4/4 + ...)
// (PyRayHF/library.py:217-229) cancels to ~1e-9 of its terms, so the last grid points of a pair are
// decided by how YT**4, YT**3 (NumPy: pow), sin and cos (NumPy float64: the platform libm) were
// ROUNDED.  glibc's sin/cos/pow return the correctly rounded result in > 99 % of the cases and
// NumPy's SIMD pow in ~95 %; (x*x)*(x*x) does in 50 % and the device library's sin/cos in ~80 %.
// These versions evaluate in double-double and round once: they agree with the correctly rounded
// value except within ~2^-9 ulp of a rounding boundary.
//
// The same source is compiled for gfx950 (hipcc) and for the host (gcc, tests/test_crmath_host.py
// checks it against exactly rounded big-rational values) - plain C++, FMA through __builtin_fma.

#ifndef PRHF_CRMATH_H
#define PRHF_CRMATH_H

#if defined(__HIPCC__)
#define PRHF_HD __host__ __device__ __forceinline__
#else
#define PRHF_HD static inline
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

// sin(r) and cos(r) for a double r, |r| < 2^20 * pi/2 (callers fall back to the device library beyond;
// field angles are degrees-sized).  r = n pi/2 + y, |y| <= pi/4 + tiny, y kept in double-double;
// both series are summed in double-double down to the terms whose double rounding error is < 2^-64
// of the result.
PRHF_HD void sincos(double r, double* s_out, double* c_out) {
    // pi/2 in three pieces of 53 bits
    const double kPio2_1 = 1.5707963267948966;        // 0x3FF921FB54442D18
    const double kPio2_2 = 6.123233995736766e-17;     // 0x3C91A62633145C07
    const double kPio2_3 = -1.4973849048591698e-33;   // 0xB91F1976B7ED8FBC
    const double kTwoOverPi = 0.6366197723675814;
    const double n = __builtin_rint(r * kTwoOverPi);
    // y = r - n pi/2: n*pio2_1 rounds to p with exact error pe, r - p is exact (p is within a factor 2 of r,
    // or zero), the remaining pieces are ~1e-16 |r| and go into the low word
    const dd p1 = two_prod(n, kPio2_1);
    const double d = r - p1.hi;
    const dd p2 = two_prod(n, kPio2_2);
    dd y = two_sum(d, -p1.lo);
    dd t = two_sum(y.hi, -p2.hi);
    t.lo += y.lo - p2.lo - n * kPio2_3;
    y = quick_two_sum(t.hi, t.lo);

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

}  // namespace prhf_cr

#endif

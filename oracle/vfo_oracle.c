/*
 * vfo_oracle.c - plain-C restatement of the vertical-ionogram forward operator.
 * TEST INFRASTRUCTURE ONLY: used by tests/ (cross-check of the NumPy oracle and of the HIP path)
 * and by bench.py's cpu_baseline leg (a fused multi-core CPU baseline).  The product package
 * never loads it.
 *
 * One (profile, frequency) pair at a time, scalar loops, libm, float64, the reference's
 * operation order (reference PyRayHF/library.py:40-509; line numbers below are that file).
 * It differs from the reference's NumPy path only where NumPy's SIMD pow is not correctly
 * rounded (~5 % of YT**4 / YT**3, by one ulp; sin/cos are the same libm calls) and, by an ulp of
 * the sum, in the summation order (NumPy sums pairwise; so does pairwise_sum below, in the same
 * blocks of 128).  NOT bit-identical to the reference: tests/test_oracle_c.py holds it to the golden
 * vectors of the reference at 1e-12 (X mode) and the noise-aware rule (O mode).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off, no fast-math)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PLASMA_CONST 8.97866275          /* library.py:61 */
#define GYRO_CONST 2.799249247e10        /* library.py:64 */
#define BACKOFF_KM 1e-6                  /* library.py:378 */
#define DEG2RAD 0.017453292519943295     /* numpy deg2rad: x * (pi/180) */
#define UNMAG_TOL 1e-12                  /* library.py:163 */

/* numpy's binary_search_with_guess result for a key inside [xp[0], xp[n-1]]: last index i with
 * xp[i] <= key (numpy/_core/src/multiarray/compiled_base.c). */
static int64_t last_le(const double* xp, int64_t n, double key) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) / 2;
        if (xp[mid] <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* np.interp(x, xp, fp) for one abscissa, with precomputed slopes as numpy does when
 * len(xp) <= len(x) (same formula either way). */
static double interp1(double x, const double* xp, const double* fp, const double* slope, int64_t n) {
    if (n == 1) return fp[0];                        /* numpy: lenxp == 1, also for NaN x */
    if (isnan(x)) return x;
    if (x > xp[n - 1]) return fp[n - 1];
    if (x < xp[0]) return fp[0];
    int64_t j = last_le(xp, n, x);
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    return slope[j] * (x - xp[j]) + fp[j];
}

/* YT**4 and YT**3 (library.py:217, :244) are pow() calls in NumPy: one rounding.  Double-double products
 * rounded once reproduce the correctly rounded value (checked against exact rationals, tests/test_oracle_c.py);
 * glibc's pow does in 99.9 % of the cases, NumPy's SIMD pow in ~95 %, (x*x)*(x*x) in 50 %. */
static double pow4_once(double x) {
    const double h = x * x, l = fma(x, x, -h);
    const double p = h * h, e = fma(h, h, -p);
    return p + fma(2.0 * h, l, e);
}
static double pow3_once(double x) {
    const double h = x * x, l = fma(x, x, -h);
    const double p = h * x, e = fma(h, x, -p);
    return p + fma(l, x, e);
}

/* library.py:161-256; mode: 0 = O, 1 = X.  Returns mu' (NaN where the reference has NaN). */
static double group_index(double X, double Y, double psi_deg, int mode, int unmag) {
    if (unmag) {                                     /* :201-207 */
        double m2 = 1.0 - X;
        if (!(m2 > 0.0)) return NAN;
        return 1.0 / sqrt(m2);
    }
    const double sgn = mode == 0 ? 1.0 : -1.0;
    const double r = psi_deg * DEG2RAD;
    const double s = sin(r), c = cos(r);
    const double YT = Y * s, YL = Y * c;             /* :210-211 */
    const double Xm1 = 1.0 - X;                      /* :214 */
    const double alpha = 0.25 * pow4_once(YT) + (YL * YL) * (Xm1 * Xm1);   /* :217 */
    const double beta = sqrt(alpha);                 /* :218 */
    const double D = (Xm1 - 0.5 * (YT * YT)) + sgn * beta;                /* :229 */
    double rad = 1.0 - X * Xm1 / D;                  /* :232 */
    if (rad < 0.0) rad = NAN;                        /* :233 */
    double mu = sqrt(rad);
    if (mu > 1.0) mu = NAN;                          /* :238 */
    const double dbdX = (-(YL * YL)) * Xm1 / beta;   /* :241 */
    const double dDdX = -1.0 + sgn * dbdX;           /* :242 */
    const double dadY = pow3_once(YT) * s + ((2.0 * YL) * (Xm1 * Xm1)) * c;   /* :244-245 */
    const double dbdY = 0.5 * dadY / beta;           /* :246 */
    const double dDdY = (-YT) * s + sgn * dbdY;      /* :247 */
    const double dmudY = (X * Xm1 * dDdY) / (2.0 * mu * (D * D));            /* :250 */
    const double dmudX = (1.0 / (2.0 * mu * D)) * (2.0 * X - 1.0 + X * Xm1 / D * dDdX);   /* :251 */
    return mu - (2.0 * X * dmudX + Y * dmudY);       /* :254 */
}

/* np.nansum along a contiguous row: NaN -> 0, then NumPy's pairwise summation
 * (blocks of <= 128 with 8 accumulators). */
static double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

/* One profile, all frequencies (library.py:459-509).  work: n_points doubles of scratch.
 * Returns 0, or -2 negative density below the peak, -3 peak at level 0. */
static int one_profile(const double* freq_mhz, int64_t n_freq, const double* den, const double* bmag,
                       const double* bpsi, const double* alt, int64_t n_alt, const double* mult,
                       int64_t n_points, int mode, double* vh, double* work, double* slopes) {
    int64_t K = 0;                                   /* :371 first-occurrence argmax */
    double alt_min = alt[0];
    for (int64_t i = 0; i < n_alt; ++i) {
        /* np.argmax: a NaN outranks every number, the first one wins */
        if (den[K] == den[K] && (den[i] != den[i] || den[i] > den[K])) K = i;
        if (alt[i] < alt_min) alt_min = alt[i];
    }
    if (K == 0) return -3;
    double bmax = 0.0, fmin = INFINITY;
    for (int64_t k = 0; k < K; ++k) {
        if (den[k] < 0.0) return -2;                 /* :93-94 */
        if (fabs(bmag[k]) > bmax) bmax = fabs(bmag[k]);
    }
    for (int64_t f = 0; f < n_freq; ++f) if (freq_mhz[f] < fmin) fmin = freq_mhz[f];
    /* :201 over the whole (F, N) array; see DESIGN.md "Deviations" (1): decided from the nodes */
    const int unmag = (GYRO_CONST * bmax) / (fmin * 1e6) < UNMAG_TOL;
    double* sden = slopes;
    double* sb = slopes + n_alt;
    double* sp = slopes + 2 * n_alt;
    for (int64_t k = 0; k + 1 < K; ++k) {
        const double da = alt[k + 1] - alt[k];
        sden[k] = (den[k + 1] - den[k]) / da;
        sb[k] = (bmag[k + 1] - bmag[k]) / da;
        sp[k] = (bpsi[k + 1] - bpsi[k]) / da;
    }
    for (int64_t f = 0; f < n_freq; ++f) {
        const double fhz = freq_mhz[f] * 1e6;        /* :491 */
        const double f2 = fhz * fhz;
        /* :380-407 running maximum and np.interp(1.0, running_max, alt) */
        double run = -INFINITY, below = -INFINITY, above = 0.0;
        int64_t kstar = K;
        for (int64_t k = 0; k < K; ++k) {
            const double fn = sqrt(den[k]) * PLASMA_CONST;
            double col = (fn * fn) / f2;
            if (mode == 1) col = col + (GYRO_CONST * bmag[k]) / fhz;
            if (col > 1.0) { kstar = k; above = col; break; }
            if (col > run) run = col;
        }
        below = run;
        double h;
        if (kstar == K) {
            if (!(below >= 1.0)) {
                if (K == 1) {                        /* np.interp with one node ignores NaN abscissae */
                    const double fn = sqrt(den[0]) * PLASMA_CONST;
                    const double term = group_index((fn * fn) / f2, (GYRO_CONST * bmag[0]) / fhz, bpsi[0], mode, unmag)
                                        * BACKOFF_KM;
                    vh[f] = (isnan(term) || term == 0.0) ? NAN : term + alt_min;
                } else {
                    vh[f] = NAN;
                }
                continue;
            }
            h = alt[K - 1];
        } else if (kstar == 0) {
            h = alt[0];
        } else {
            const int64_t j = kstar - 1;
            h = (below == 1.0) ? alt[j] : (alt[j + 1] - alt[j]) / (above - below) * (1.0 - below) + alt[j];
        }
        h -= BACKOFF_KM;                             /* :407 */
        const double span = h - alt[0];
        double z = mult[0] * span + alt[0];          /* :413 */
        for (int64_t i = 0; i < n_points; ++i) {
            const double znext = (i + 1 < n_points) ? mult[i + 1] * span + alt[0] : 0.0;
            const double dh = (i + 1 < n_points) ? znext - z : BACKOFF_KM;      /* :415-416 */
            const double d = interp1(z, alt, den, sden, K);                      /* :424-426 */
            const double b = interp1(z, alt, bmag, sb, K);
            const double p = interp1(z, alt, bpsi, sp, K);
            const double fn = sqrt(d) * PLASMA_CONST;                            /* :96 */
            const double term = group_index((fn * fn) / f2, (GYRO_CONST * b) / fhz, p, mode, unmag) * dh;   /* :288 */
            work[i] = isnan(term) ? 0.0 : term;
            z = znext;
        }
        const double total = pairwise_sum(work, n_points);
        vh[f] = (total == 0.0) ? NAN : total + alt_min;                          /* :290-292 */
    }
    return 0;
}

/* (P, N_alt) profiles x (F) frequencies -> (P, F).  alt_stride 0 = shared altitude column.
 * n_threads <= 0: all cores.  Returns the first non-zero per-profile status. */
int vfo_oracle_batch(const double* freq_mhz, int64_t n_freq, const double* den, const double* bmag,
                     const double* bpsi, const double* alt, int64_t n_prof, int64_t n_alt,
                     int64_t alt_stride, const double* mult, int64_t n_points, int mode, double* vh,
                     int n_threads) {
    int status = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        double* work = (double*)malloc(sizeof(double) * (size_t)(n_points + 3 * n_alt));
#pragma omp for schedule(dynamic, 1)
        for (int64_t p = 0; p < n_prof; ++p) {
            int rc = one_profile(freq_mhz, n_freq, den + p * n_alt, bmag + p * n_alt, bpsi + p * n_alt,
                                 alt + p * alt_stride, n_alt, mult, n_points, mode, vh + p * n_freq, work,
                                 work + n_points);
            if (rc != 0) {
#pragma omp critical
                if (status == 0) status = rc;
                for (int64_t f = 0; f < n_freq; ++f) vh[p * n_freq + f] = NAN;
            }
        }
        free(work);
    }
    return status;
}

int vfo_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

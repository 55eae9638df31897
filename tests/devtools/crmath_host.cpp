// Host build of pyrayhf_amd/csrc/prhf_crmath.h for tests/test_crmath_host.py (g++ -O2 -mfma -ffp-contract=off).
#include "prhf_crmath.h"

extern "C" {
void cr_sincos(const double* r, long n, double* s, double* c) {
    for (long i = 0; i < n; ++i) prhf_cr::sincos(r[i], &s[i], &c[i]);
}
void cr_sincos_table(const double* r, long n, double* s, double* c) {
    for (long i = 0; i < n; ++i) prhf_cr::sincos_table(r[i], &s[i], &c[i]);
}
void cr_pow34(const double* x, long n, double* p3, double* p4) {
    for (long i = 0; i < n; ++i) {
        p3[i] = prhf_cr::pow3(x[i]);
        p4[i] = prhf_cr::pow4(x[i]);
    }
}
}

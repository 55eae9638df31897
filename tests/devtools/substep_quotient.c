/* The tracers' sub-step nodes t = j / N (reference library.py:1656-1657) are formed on the GPU from the correctly
 * rounded reciprocal of N and one fma correction (prhf_snell.inc substep_node): this checks that form against the IEEE
 * division for every 0 <= j <= N <= limit.  Test infrastructure (tests/test_host_substep_quotient.py). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
    const int limit = argc > 1 ? atoi(argv[1]) : 8192;
    long bad = 0, total = 0;
    for (int N = 1; N <= limit; ++N) {
        const double dN = (double)N, rN = 1.0 / dN;
        for (int j = 0; j <= N; ++j) {
            const double a = (double)j;
            const double q0 = a * rN;
            const double q = fma(fma(-q0, dN, a), rN, q0);
            if (q != a / dN) ++bad;
            ++total;
        }
    }
    printf("%ld %ld\n", total, bad);
    return bad != 0;
}

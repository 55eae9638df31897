// Host-side code of the build under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizers_host.py; CPU
// only - GPU sanitizers are not available on the pool).  One executable, three parts:
//   1. the launch planner of libprhf.so (pyrayhf_amd/csrc/prhf_plan.h: plan_slice, validate_work_list,
//      first_decreasing_grid_entry) over a sweep of launch shapes and option settings, with the invariants the kernels
//      rely on checked on every plan;
//   2. the double-double sin / cos / pow of the reference-order tier (prhf_crmath.h via crmath_host.cpp);
//   3. the plain-C oracle (oracle/vfo_oracle.c, TEST INFRASTRUCTURE) on a small seeded batch with its edge cases.
// Prints "sanitize_host: ok" and exits 0; a sanitizer report or a broken invariant ends it with a non-zero code.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "prhf_plan.h"

extern "C" {
void cr_sincos(const double* r, long n, double* s, double* c);
void cr_sincos_table(const double* r, long n, double* s, double* c);
void cr_pow34(const double* x, long n, double* p3, double* p4);
int vfo_oracle_batch(const double* freq_mhz, int64_t n_freq, const double* den, const double* bmag, const double* bpsi,
                     const double* alt, int64_t n_prof, int64_t n_alt, int64_t alt_stride, const double* mult,
                     int64_t n_points, int mode, double* vh, int n_threads);
}

static int failures = 0;
#define CHECK(cond, ...)                                          \
    do {                                                          \
        if (!(cond)) {                                            \
            std::fprintf(stderr, "CHECK failed: %s: ", #cond);    \
            std::fprintf(stderr, __VA_ARGS__);                    \
            std::fprintf(stderr, "\n");                           \
            ++failures;                                           \
        }                                                         \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static double uniform(double lo, double hi) { return lo + (hi - lo) * ((next_u64() >> 11) * (1.0 / 9007199254740992.0)); }

static void planner() {
    const long long slots_of[] = {256, 512, 1024};
    const int points_of[] = {1, 2, 50, 64, 65, 200, 256, 257, 1024, 2000, 8192, 20000, 100000};
    const long long profs_of[] = {0, 1, 2, 63, 64, 625, 2500, 12500, 100000};
    const long long freqs_of[] = {1, 3, 174, 256, 512, 1024};
    for (int variant = 0; variant < 4; ++variant) {
        Knobs kn;
        if (variant == 1) { kn.local_chunks = 0; kn.split_few_profiles = 0; }
        if (variant == 2) { kn.tail_bpp = 1; kn.target_waves = 64; }
        if (variant == 3) { kn.tail_rounds = 3.5; kn.tail_bpp = 64; kn.split_min_points = 1; kn.target_waves = 1e9; }
        for (long long slots : slots_of) for (int n : points_of) for (long long P : profs_of) for (long long F : freqs_of) {
            prhf::SegDev s;
            std::memset(&s, 0, sizeof s);
            s.prof_begin = 7; s.prof_end = 7 + P; s.n_points = n;
            plan_slice(s, F, slots, kn);
            CHECK(s.chunks >= 1 && s.chunk_len >= 64 && s.chunk_len % 64 == 0, "n=%d P=%lld F=%lld: chunks %d x %d", n, P, F, s.chunks, s.chunk_len);
            CHECK((long long)s.chunks * s.chunk_len >= n, "chunks do not cover the grid: %d x %d < %d", s.chunks, s.chunk_len, n);
            CHECK((long long)(s.chunks - 1) * s.chunk_len < n, "an empty chunk: %d x %d for %d", s.chunks, s.chunk_len, n);
            CHECK(s.blocks_per_prof >= 1 && s.tail_bpp >= 1, "blocks per profile %d / %d", s.blocks_per_prof, s.tail_bpp);
            CHECK(s.tail_prof >= 0 && s.tail_prof <= P, "tail_prof %lld of %lld", s.tail_prof, P);
            CHECK(s.slots == 0 || ((s.slots & (s.slots - 1)) == 0 && s.slots <= kWavesPerBlock && s.chunks <= s.slots),
                  "block-local chunks: %d chunks in %d slots", s.chunks, s.slots);
            if (s.slots > 0) CHECK((long long)s.blocks_per_prof * kWavesPerBlock >= F * s.slots, "slots do not fit the workgroups");
            // a workgroup has at least one item per wave, or is the only one of its profile; the tail is cut into at most
            // tail_bpp (<= 64) workgroups per profile
            const long long items = F * (s.slots > 0 ? s.slots : s.chunks);
            CHECK(s.blocks_per_prof == 1 || (long long)(s.blocks_per_prof - 1) * kWavesPerBlock < items,
                  "%d workgroups per profile for %lld items", s.blocks_per_prof, items);
            CHECK(s.tail_bpp <= 64 || s.tail_bpp == s.blocks_per_prof, "tail_bpp %d", s.tail_bpp);
        }
    }
    // work lists: every rule of validate_work_list, and the grid check
    char why[160];
    prhf_segment ok[3] = {{0, 10, PRHF_MODE_O, 200, 0, 0}, {10, 30, PRHF_MODE_X, 2000, 200, 10 * 174}, {30, 30, PRHF_MODE_X, 50, 2200, 30 * 174}};
    CHECK(validate_work_list(ok, 3, 30, 174, 2250, why, sizeof why) == PRHF_OK, "%s", why);
    struct Bad { prhf_segment seg; const char* what; } bad[] = {
        {{-1, 10, PRHF_MODE_O, 200, 0, 0}, "negative begin"}, {{5, 4, PRHF_MODE_O, 200, 0, 0}, "end < begin"},
        {{0, 31, PRHF_MODE_O, 200, 0, 0}, "end > n_prof"}, {{0, 10, 7, 200, 0, 0}, "mode"}, {{0, 10, PRHF_MODE_O, 0, 0, 0}, "n_points"},
        {{0, 10, PRHF_MODE_O, 200, -1, 0}, "grid offset"}, {{0, 10, PRHF_MODE_O, 200, 2100, 0}, "grid end"},
        {{0, 10, PRHF_MODE_O, 200, 0, 5}, "output offset not on a row"}, {{0, 10, PRHF_MODE_O, 200, 0, -174}, "negative output offset"}};
    for (const Bad& b : bad) CHECK(validate_work_list(&b.seg, 1, 30, 174, 2250, why, sizeof why) == PRHF_EINVAL && why[0], "%s accepted", b.what);
    prhf_segment overlap[2] = {{0, 10, PRHF_MODE_O, 200, 0, 0}, {10, 20, PRHF_MODE_X, 200, 0, 5 * 174}};
    CHECK(validate_work_list(overlap, 2, 30, 174, 2250, why, sizeof why) == PRHF_EINVAL, "overlapping rows accepted");
    std::vector<double> grid(2250);
    for (size_t i = 0; i < grid.size(); ++i) grid[i] = (double)(i % 250) / 250.0;      // decreases at 250, 500, ...: between slices only
    prhf_segment tiles[2] = {{0, 1, PRHF_MODE_O, 250, 0, 0}, {1, 2, PRHF_MODE_O, 250, 250, 174}};
    CHECK(first_decreasing_grid_entry(grid.data(), 2250, tiles, 2) == -1, "a step between two slices is not a decrease");
    grid[100] = -1.0;
    CHECK(first_decreasing_grid_entry(grid.data(), 2250, tiles, 2) == 100, "the decrease at 100 was missed");
    prhf_segment outside = {0, 1, PRHF_MODE_O, 250, 2100, 0};
    CHECK(first_decreasing_grid_entry(grid.data(), 2250, &outside, 1) == -1, "a range outside the array must not be read");
}

static void crmath() {
    const long n = 4096;
    std::vector<double> r(n), s(n), c(n), s2(n), c2(n), p3(n), p4(n);
    for (long i = 0; i < n; ++i) r[i] = (i < n / 2) ? uniform(0.0, 1.5707963267948966) : uniform(-40.0, 40.0);
    r[0] = 0.0; r[1] = 1.5707963267948966; r[2] = -0.0; r[3] = 1e-300; r[4] = 0.7853981633974483;
    cr_sincos(r.data(), n, s.data(), c.data());
    cr_sincos_table(r.data(), n, s2.data(), c2.data());
    cr_pow34(r.data(), n, p3.data(), p4.data());
    for (long i = 0; i < n; ++i) {
        CHECK(std::fabs(s[i] - std::sin(r[i])) <= 2.3e-16 * std::fabs(std::sin(r[i])) + 1e-300, "sin(%.17g)", r[i]);
        CHECK(std::fabs(c2[i] - std::cos(r[i])) <= 2.3e-16 * std::fabs(std::cos(r[i])) + 1e-300, "cos(%.17g)", r[i]);
        CHECK(std::fabs(s2[i] - s[i]) <= 2.3e-16 * std::fabs(s[i]) + 1e-300, "table sin(%.17g)", r[i]);
        CHECK(std::fabs(p4[i] - r[i] * r[i] * r[i] * r[i]) <= 4e-16 * p4[i] + 1e-300, "pow4(%.17g)", r[i]);
        (void)p3; (void)c;
    }
}

static void oracle() {
    const int64_t n_alt = 620, P = 6, F = 40;
    std::vector<double> alt(n_alt), den(P * n_alt), bmag(P * n_alt), bpsi(P * n_alt), freq(F);
    for (int64_t k = 0; k < n_alt; ++k) alt[k] = 80.0 + (double)k;
    for (int64_t p = 0; p < P; ++p) {
        const double nm = std::pow(10.0, uniform(11.3, 12.5)), hm = uniform(220.0, 420.0), hs = uniform(35.0, 70.0);
        const double ne = std::pow(10.0, uniform(10.3, 11.3)), he = uniform(6.0, 12.0), b0 = uniform(2.2e-5, 6.0e-5), psi0 = uniform(0.0, 89.0);
        for (int64_t k = 0; k < n_alt; ++k) {
            const double z = (alt[k] - hm) / hs, ze = (alt[k] - 110.0) / he;
            den[p * n_alt + k] = nm * std::exp(0.5 * (1.0 - z - std::exp(-z))) + ne * std::exp(0.5 * (1.0 - ze - std::exp(-ze)));
            bmag[p * n_alt + k] = (p == 4 ? 0.0 : b0) * std::pow((6371.0 + 80.0) / (6371.0 + alt[k]), 3.0);   // profile 4: unmagnetised
            bpsi[p * n_alt + k] = psi0 + 0.001 * (alt[k] - 80.0);
        }
    }
    for (int64_t k = 2; k < n_alt; ++k) den[5 * n_alt + k] = den[5 * n_alt + 1] * 0.5;     // profile 5: peak at level 1
    den[5 * n_alt] = den[5 * n_alt + 1] * 0.25;
    for (int64_t f = 0; f < F; ++f) freq[f] = 0.3 + 0.45 * (double)f;                     // some below the gyrofrequency, some escape
    const int grids[] = {2, 3, 50, 200, 2000};
    for (int mode = 0; mode < 2; ++mode)
        for (int n : grids) {
            std::vector<double> mult(n), vh(P * F, -1.0);
            for (int i = 0; i < n; ++i) {                                                  // smooth_nonuniform_grid(0, 1, n, 10)
                const double u = (double)i / (double)(n - 1);
                mult[i] = 1.0 - (std::exp(10.0 * (1.0 - u)) - 1.0) / (std::exp(10.0) - 1.0);
            }
            const int rc = vfo_oracle_batch(freq.data(), F, den.data(), bmag.data(), bpsi.data(), alt.data(), P, n_alt, 0,
                                            mult.data(), n, mode, vh.data(), 2);
            CHECK(rc == 0, "oracle status %d (mode %d, n %d)", rc, mode, n);
            int finite = 0;
            for (double v : vh) {
                CHECK(v != -1.0, "an output the oracle never wrote");
                if (v == v) { ++finite; CHECK(v >= 80.0 - 1e-9 && v < 5000.0, "virtual height %g", v); }
            }
            CHECK(finite > 0 && finite < (int)vh.size(), "mode %d n %d: %d finite of %zu", mode, n, finite, vh.size());
        }
    // error paths: a negative density, a peak at level 0
    std::vector<double> mult(50), vh(F);
    for (int i = 0; i < 50; ++i) mult[i] = (double)i / 49.0;
    std::vector<double> neg(den.begin(), den.begin() + n_alt);
    neg[3] = -1.0;
    CHECK(vfo_oracle_batch(freq.data(), F, neg.data(), bmag.data(), bpsi.data(), alt.data(), 1, n_alt, 0, mult.data(), 50, 0, vh.data(), 1) != 0,
          "a negative density went through");
    std::vector<double> falling(n_alt);
    for (int64_t k = 0; k < n_alt; ++k) falling[k] = 1e12 / (1.0 + (double)k);
    CHECK(vfo_oracle_batch(freq.data(), F, falling.data(), bmag.data(), bpsi.data(), alt.data(), 1, n_alt, 0, mult.data(), 50, 1, vh.data(), 1) != 0,
          "a peak at level 0 went through");
}

int main() {
    planner();
    crmath();
    oracle();
    if (failures) {
        std::fprintf(stderr, "sanitize_host: %d checks failed\n", failures);
        return 1;
    }
    std::printf("sanitize_host: ok\n");
    return 0;
}

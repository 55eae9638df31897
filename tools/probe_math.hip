// probe_math.hip - measure on gfx950 (a) accuracy of v_rcp_f64 / v_rsq_f64 and of refinement
// schemes, (b) issue cost of FP64 VALU instructions.  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/probe_math.hip -o /tmp/probe_math && /tmp/probe_math
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_eval(const double* x, double* out, int n, int which) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], y;
    switch (which) {
    case 0: y = __builtin_amdgcn_rcp(v); break;
    case 1: y = __builtin_amdgcn_rsq(v); break;
    case 2: { y = __builtin_amdgcn_rcp(v); double e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y); } break;      // 1 NR
    case 3: { y = __builtin_amdgcn_rcp(v); double e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
              e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y); } break;                                             // 2 NR
    case 4: { y = __builtin_amdgcn_rcp(v); double e = __builtin_fma(-v, y, 1.0); double p = __builtin_fma(e, e, e);
              y = __builtin_fma(y, p, y); } break;                                                                            // cubic
    case 5: { y = __builtin_amdgcn_rsq(v); double h = 0.5 * y; double e = __builtin_fma(-(v * y), h, 0.5);
              y = __builtin_fma(y, e, y); } break;                                                                            // 1 NR
    case 6: { y = __builtin_amdgcn_rsq(v); double h = 0.5 * y; double e = __builtin_fma(-(v * y), h, 0.5);
              y = __builtin_fma(y, e, y); h = 0.5 * y; e = __builtin_fma(-(v * y), h, 0.5); y = __builtin_fma(y, e, y); } break;  // 2 NR
    case 7: { y = __builtin_amdgcn_rsq(v); double t = v * y; double e = __builtin_fma(-t, y, 1.0);   // e = 1 - v y^2
              double p = __builtin_fma(0.375, e, 0.5); double ye = y * e; y = __builtin_fma(ye, p, y); } break;              // cubic
    case 8: y = 1.0 / v; break;
    case 9: y = 1.0 / sqrt(v); break;
    case 10: { float f = __builtin_amdgcn_rsqf((float)v); y = (double)f; double t = v * y; double e = __builtin_fma(-t, y, 1.0);
               double p = __builtin_fma(0.375, e, 0.5); double ye = y * e; y = __builtin_fma(ye, p, y);
               t = v * y; e = __builtin_fma(-t, y, 1.0); p = __builtin_fma(0.375, e, 0.5); ye = y * e; y = __builtin_fma(ye, p, y); } break; // f32 seed + 2 cubic
    default: y = 0;
    }
    out[i] = y;
}

template <int OP>
__global__ void k_rate(double* out, int iters) {
    double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#define STEP(v) \
        if (OP == 0) v = __builtin_fma(v, b, c); \
        else if (OP == 1) v = v * b; \
        else if (OP == 2) v = v + c; \
        else if (OP == 3) v = __builtin_amdgcn_rcp(v); \
        else if (OP == 4) v = __builtin_amdgcn_rsq(v); \
        else if (OP == 5) v = __builtin_amdgcn_sqrt(v); \
        else if (OP == 6) { float f = (float)v; f = __builtin_amdgcn_rsqf(f); v = (double)f; } \
        else if (OP == 7) v = fmax(v, c); \
        else if (OP == 8) v = __builtin_amdgcn_ldexp(v, 1);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
double rate(double* d_out, int waves_per_simd) {
    const int iters = 20000, blocks = 256 * 4, threads = 64 * waves_per_simd;   // 1024 blocks: one per SIMD-ish
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: waves_per_simd * iters * 8 ; time -> ns per wave-instruction per SIMD
    return ms * 1e6 / ((double)waves_per_simd * iters * 8.0);
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), y(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> mant(1.0, 4.0), ex(-40, 10);
    for (int i = 0; i < n; ++i) x[i] = mant(rng) * std::pow(2.0, std::floor(ex(rng)));
    double *dx, *dy;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dy, n * 8));
    CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    const char* names[] = {"v_rcp_f64", "v_rsq_f64", "rcp+1NR", "rcp+2NR", "rcp+cubic", "rsq+1NR", "rsq+2NR", "rsq+cubic",
                           "1.0/x (IEEE)", "1/sqrt(x) (IEEE)", "rsq_f32 seed + 2 cubic"};
    for (int w = 0; w < 11; ++w) {
        hipLaunchKernelGGL(k_eval, dim3(n / 256), dim3(256), 0, 0, dx, dy, n, w);
        CHECK(hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost));
        double worst = 0, sum = 0;
        const bool is_rcp = (w == 0 || w == 2 || w == 3 || w == 4 || w == 8);
        for (int i = 0; i < n; ++i) {
            long double ref = is_rcp ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
            double rel = (double)fabsl(((long double)y[i] - ref) / ref);
            worst = rel > worst ? rel : worst; sum += rel;
        }
        printf("%-24s max rel err %.3e (2^%.1f)  mean %.3e\n", names[w], worst, std::log2(worst > 0 ? worst : 1e-300), sum / n);
    }
    const char* ops[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "cvt+v_rsq_f32+cvt", "v_max_f64", "v_ldexp_f64"};
    for (int wps = 1; wps <= 4; wps *= 2) {
        printf("-- %d wave(s) per SIMD: ns per wave-instruction per SIMD (x clock GHz = cycles)\n", wps);
        double r[9];
        r[0] = rate<0>(dy, wps); r[1] = rate<1>(dy, wps); r[2] = rate<2>(dy, wps); r[3] = rate<3>(dy, wps); r[4] = rate<4>(dy, wps);
        r[5] = rate<5>(dy, wps); r[6] = rate<6>(dy, wps); r[7] = rate<7>(dy, wps); r[8] = rate<8>(dy, wps);
        for (int o = 0; o < 9; ++o) printf("   %-20s %.3f ns  (%.1f cycles @2.4GHz)\n", ops[o], r[o], r[o] * 2.4);
    }
    return 0;
}
